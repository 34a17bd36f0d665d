"""Text-level synthetic inputs for the golden fixtures (draft FASTA + .fai, SAM text, FASTQ pairs).

This repo's own generator (SplitMix64, no Python `random`), shaped after SURVEY.md §8d: a true
genome with planted N-run gaps in the draft, FR read pairs with substitution errors, and a SAM
derived from truth (soft/hard clips at gap edges, unmapped reads inside gaps, a MAPQ mix that
straddles the reference's thresholds 0 / 30 / 60, chimeric mates on other scaffolds, abnormal
inserts).  Only fields 0-8 of each SAM line carry information (the reference reads nothing else,
collect_reads_for_gaps.py:76-91); SEQ/QUAL are `*`.
"""

MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)

    def below(self, n):
        return self.next() % n

    def chance(self, num, den):
        return self.next() % den < num

    def gauss_int(self, mean, sd):
        # Irwin-Hall(12) integer approximation of N(mean, sd)
        t = sum(self.next() % 4096 for _ in range(12)) - 12 * 2047.5
        return int(round(mean + sd * t / (4096.0 * 1.0)))


COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def wrap(seq, w=60):
    return "\n".join(seq[i:i + w] for i in range(0, len(seq), w)) + "\n"


CASES = {
    # name: scaffolds [(name, length, [(gap_start, gap_len), ...])], libs [(IS, sd, n_pairs)]
    "twolib": {
        "scaffolds": [
            ("scf0", 30000, [(4000, 600), (4780, 150), (15000, 2000), (26000, 50)]),  # 2nd gap 180 bp after 1st; last < min_gap
            ("scf1", 8000, []),
            ("scf2", 20000, [(150, 400), (9000, 1000)]),  # first gap closer than flank_length to the scaffold start
        ],
        "libs": [(300, 30, 1700), (5000, 500, 500)],
        "min_gap": 100, "flank": 300, "L": 150,
    },
    # IS exactly at the 750 mode switch (collect_reads_for_gaps.py:275) and one below; |TLEN| crafted at dist1/dist2 +- 1;
    # non-default min_gap_size / flank_length
    "bounds": {
        "scaffolds": [("s1", 15000, [(2500, 60), (2600, 49), (7000, 900)]), ("s2", 10000, [(5000, 300)])],
        "libs": [(750, 20, 900), (749, 20, 700)],
        "min_gap": 50, "flank": 150, "L": 120,
        "craft_inserts": [808, 809, 810, 811, 688, 689, 690, 691, 121, 2000],
    },
    # not a golden case: error-free reads over short gaps, used by the end-to-end closure test (tests/test_gpu_pipeline.py)
    "closable": {
        "scaffolds": [("c0", 12000, [(3000, 120), (6000, 140), (9000, 110)]), ("c1", 9000, [(4000, 130)])],
        "libs": [(300, 30, 2800)],
        "min_gap": 100, "flank": 300, "L": 150, "no_err": True,
    },
    # BASELINE.json configs[0] / SURVEY.md §8d "C1": 100 kb, one 2-kb gap at [49 000, 51 000), 10 000 pairs of 2x150 (30x), IS 300/30
    "c1": {
        "scaffolds": [("chr1", 100000, [(49000, 2000)])],
        "libs": [(300, 30, 10000)],
        "min_gap": 100, "flank": 300, "L": 150,
    },
    "edge": {
        "scaffolds": [
            ("ctgA", 6000, [(3, 120), (2000, 100), (2250, 99), (2500, 300), (5890, 110)]),  # start<5; ==min; <min; trailing N-run
            ("ctgB.1", 5000, [(1000, 500), (1900, 400), (2650, 350)]),  # three gaps chained inside one window
        ],
        "libs": [(260, 20, 900)],   # IS < 750 -> short-IS branch; dist1=200, dist2=320
        "min_gap": 100, "flank": 300, "L": 100,
    },
}


def make_case(name, seed):
    spec = CASES[name]
    rng = SplitMix64(seed)
    L = spec["L"]
    true_seqs, draft_seqs = {}, {}
    for sname, slen, gaps in spec["scaffolds"]:
        t = "".join("ACGT"[rng.below(4)] for _ in range(slen))
        d = list(t)
        for gs, gl in gaps:
            for i in range(gs, min(slen, gs + gl)):
                d[i] = "N"
        # a few lower-case (soft-masked) bases right after one gap: the reference's scan only stops at UPPER-case ACGT
        if name == "edge" and sname == "ctgA":
            for i in range(2100, 2104):
                d[i] = d[i].lower() if d[i] != "N" else "n"
        true_seqs[sname] = t
        draft_seqs[sname] = "".join(d)
    draft_fa, fai, off = "", "", 0
    for sname, slen, _ in spec["scaffolds"]:
        hdr = ">" + sname + " synthetic\n"
        body = wrap(draft_seqs[sname])
        fai += "%s\t%d\t%d\t60\t61\n" % (sname, slen, off + len(hdr))
        draft_fa += hdr + body
        off += len(hdr) + len(body)
    names = [s[0] for s in spec["scaffolds"]]
    lens = {s[0]: s[1] for s in spec["scaffolds"]}
    gapmap = {s[0]: [(gs, min(s[1], gs + gl)) for gs, gl in s[2]] for s in spec["scaffolds"]}

    def align(sname, s):
        """truth -> (mapped, pos1, cigar).  Aligned part = longest stretch outside gaps, >= 20 bp."""
        e = s + L
        segs, cur = [], s
        for gs, ge in gapmap[sname]:
            if ge <= s or gs >= e:
                continue
            if gs > cur:
                segs.append((cur, gs))
            cur = max(cur, ge)
        if cur < e:
            segs.append((cur, e))
        if not segs:
            return False, 0, "*"
        a, b = max(segs, key=lambda x: (x[1] - x[0], -x[0]))
        if b - a < 20:
            return False, 0, "*"
        lc, rc = a - s, e - b
        clipch = "H" if rng.chance(1, 12) else "S"
        cig = ("%d%s" % (lc, clipch) if lc else "") + "%dM" % (b - a) + ("%d%s" % (rc, clipch) if rc else "")
        return True, a + 1, cig

    libs = []
    for li, (IS, sd, npairs) in enumerate(spec["libs"]):
        recs, fq1, fq2 = [], [], []
        for p in range(npairs):
            qn = "r%d_%d" % (li, p)
            sname = names[rng.below(len(names))]
            ins = max(L + 1, rng.gauss_int(IS, sd))
            kind = rng.below(100)
            crafted = spec.get("craft_inserts")
            if kind < 3:
                ins = ins * 3          # too long  (>= dist2)
            elif kind < 6:
                ins = L + 10 + rng.below(40)   # too short (<= dist1 in the short-IS branch)
            if crafted and p < 40 * len(crafted):
                ins, kind = crafted[p % len(crafted)], 50      # an ordinary same-scaffold pair with an exact insert
            ins = min(ins, lens[sname] - 1)
            s1 = rng.below(lens[sname] - ins)
            s2 = s1 + ins - L
            m2name = sname
            if 6 <= kind < 10:      # chimeric: mate on another scaffold (or far away on the same one)
                m2name = names[rng.below(len(names))]
                s2 = rng.below(lens[m2name] - L)
            seqs = []
            for (sn, s, rev) in ((sname, s1, False), (m2name, s2, True)):
                t = list(true_seqs[sn][s:s + L])
                for _ in range(2):
                    if rng.chance(3, 8):
                        j = rng.below(L)
                        sub = "ACGT"[("ACGT".index(t[j]) + 1 + rng.below(3)) % 4]
                        if not spec.get("no_err"):
                            t[j] = sub
                t = "".join(t)
                seqs.append(revcomp(t) if rev else t)
            first_is_fwd = not rng.chance(1, 2)   # which mate number the forward read gets
            al = [align(sname, s1), align(m2name, s2)]
            mq = []
            for _ in range(2):
                r = rng.below(100)
                mq.append(60 if r < 82 else 0 if r < 90 else (29, 30, 31)[rng.below(3)] if r < 95 else 1 + rng.below(59))
            ends = [(sname, s1, False), (m2name, s2, True)]
            for i in (0, 1):
                j = 1 - i
                mapped, pos, cig = al[i]
                mmapped, mpos, _ = al[j]
                flag = 0x1
                flag |= 0x40 if ((i == 0) == first_is_fwd) else 0x80
                if ends[i][2]:
                    flag |= 0x10
                if ends[j][2]:
                    flag |= 0x20
                if not mapped:
                    flag |= 0x4
                if not mmapped:
                    flag |= 0x8
                rname, rpos = ends[i][0], pos
                if not mapped:
                    if mmapped:
                        rname, rpos = ends[j][0], mpos
                    else:
                        rname, rpos = "*", 0
                if mmapped:
                    rnext, pnext = ends[j][0], mpos
                elif mapped:
                    rnext, pnext = rname, rpos
                else:
                    rnext, pnext = "*", 0
                tlen = 0
                if mapped and mmapped and ends[i][0] == ends[j][0]:
                    lo = min(ends[0][1], ends[1][1])
                    hi = max(ends[0][1], ends[1][1]) + L
                    tlen = (hi - lo) if ends[i][1] <= ends[j][1] else -(hi - lo)
                    if kind >= 10:
                        flag |= 0x2
                rn = "=" if (rnext == rname and rname != "*") else rnext
                q = mq[i] if mapped else 0
                recs.append((names.index(rname) if rname != "*" else len(names), rpos, len(recs),
                             "\t".join([qn, str(flag), rname, str(rpos), str(q), cig if mapped else "*",
                                        rn, str(pnext), str(tlen), "*", "*"]) + "\n"))
            a, b = (0, 1) if first_is_fwd else (1, 0)
            extra = " lib%d" % li if p % 3 == 0 else ""
            fq1.append("@%s/1%s\n%s\n+\n%s\n" % (qn, extra, seqs[a], "I" * L))
            fq2.append("@%s/2%s\n%s\n+\n%s\n" % (qn, extra, seqs[b], "I" * L))
        recs.sort(key=lambda r: (r[0], r[1], r[2]))
        libs.append({"is": IS, "sd": sd, "sam": "".join(r[3] for r in recs),
                     "fq1": "".join(fq1), "fq2": "".join(fq2)})
    return {"name": name, "seed": seed, "draft_fa": draft_fa, "fai": fai, "libs": libs,
            "min_gap": spec["min_gap"], "flank": spec["flank"], "true_seqs": true_seqs}
