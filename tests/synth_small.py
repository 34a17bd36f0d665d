"""Array-level synthetic cases for smoke() and the GPU parity tests: a random genome with planted gaps, FR pairs
with substitution errors, truth-derived 32-byte alignment records (clips at gap edges, unmapped inside gaps,
MAPQ mix, chimeric mates, abnormal inserts).  numpy only; the same arrays feed the HIP path and the oracle."""
import numpy as np

from oracle.c_oracle import ALNREC, GAP

_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def small_case(seed=1, n_pairs=4000, L=150, n_scaffolds=3, scaffold_len=30000, gaps_per_scaffold=3, gap_len=800,
               insert=300, sd=30, flank=300, err=0.005, n_frac=0.0):
    rng = np.random.RandomState(seed)
    genome = [rng.randint(0, 4, scaffold_len).astype(np.uint8) for _ in range(n_scaffolds)]
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    gaps = []
    for s in range(n_scaffolds):
        for j in range(gaps_per_scaffold):
            st = (j + 1) * scaffold_len // (gaps_per_scaffold + 1) + rng.randint(-50, 50)
            gaps.append((s, st, st + gap_len, j + 1))
    garr = np.array(gaps, dtype=GAP)
    truth = [lut[g].tobytes() for g in genome]
    flanks = []
    for (s, st, en, _) in gaps:
        t = truth[s]
        flanks.append((t[max(0, st - flank):st - 5].decode(), t[en + 5:en + flank].decode()))
    reads, recs = [], np.zeros(2 * n_pairs, dtype=ALNREC)

    def align(s, pos):
        lo, hi = pos, pos + L
        for (gs, gst, gen, _) in gaps:
            if gs != s or gen <= lo or gst >= hi:
                continue
            left, right = max(0, gst - lo), max(0, hi - gen)
            if left >= right:
                return (left >= 20, lo + 1, 2 if left < L else 0) if left >= 20 else (False, 0, 0)
            return (True, gen + 1, 1) if right >= 20 else (False, 0, 0)
        return True, lo + 1, 0

    for p in range(n_pairs):
        s = rng.randint(n_scaffolds)
        ins = max(L + 1, int(round(rng.normal(insert, sd))))
        kind = rng.randint(100)
        if kind < 3:
            ins *= 3
        elif kind < 6:
            ins = L + 10 + rng.randint(40)
        a = rng.randint(0, scaffold_len - ins)
        b = a + ins - L
        s2 = s
        if 6 <= kind < 9:
            s2 = rng.randint(n_scaffolds)
            b = rng.randint(0, scaffold_len - L)
        seqs = []
        for (sc, pos, rev) in ((s, a, False), (s2, b, True)):
            t = bytearray(truth[sc][pos:pos + L])
            for j in np.nonzero(rng.random_sample(L) < err)[0]:
                t[j] = b"ACGT"[(b"ACGT".index(t[j]) + 1 + rng.randint(3)) % 4]
            if n_frac and rng.random_sample() < n_frac:
                t[rng.randint(L)] = ord("N")
            t = bytes(t)
            seqs.append(t.translate(_COMP)[::-1] if rev else t)
        flip = rng.randint(2)
        al = [align(s, a), align(s2, b)]
        mq = [60 if r < 85 else 0 if r < 92 else 29 + r % 3 for r in rng.randint(0, 100, 2)]
        ends = [(s, a), (s2, b)]
        for i in (0, 1):
            j = 1 - i
            m, pos, clip = al[i]
            mm, mpos, _ = al[j]
            flag = 1 | (0x40 if (i == 0) != bool(flip) else 0x80) | (0x10 if i == 1 else 0x20)
            if not m:
                flag |= 4
            if not mm:
                flag |= 8
            ref = ends[i][0] if m else (ends[j][0] if mm else 0xFFFFFFFF)
            rpos = pos if m else (mpos if mm else 0)
            mref = ends[j][0] if mm else (ref if m else 0xFFFFFFFF)
            mp = mpos if mm else rpos
            tlen = 0
            if m and mm and ends[0][0] == ends[1][0]:
                span = max(a, b) + L - min(a, b)
                tlen = span if ends[i][1] <= ends[j][1] else -span
            mate_no = 0 if (flag & 0x40) else 1
            recs[2 * p + i] = (rpos, mp, tlen, ref, mref, flag, mq[i] if m else 0, clip if m else 0, 2 * p + mate_no)
        first, second = (seqs[0], seqs[1]) if not flip else (seqs[1], seqs[0])
        reads.append(first)
        reads.append(second)
    order = np.lexsort((recs["pos"], recs["ref"]))
    return {"gaps": garr, "n_scaffolds": n_scaffolds, "flanks": flanks, "reads_blob": b"".join(reads), "L": L,
            "recs": recs[order].copy(), "n_reads": 2 * n_pairs}
