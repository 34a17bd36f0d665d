"""csrc/textio.hip (host loops behind the C ABI) against the per-record Python formatting it replaces: bam_io.BamCols' SAM columns +
collect_both_unmapped_reads' FASTQ form, and device_collect's re-written FASTQ record.  No GPU needed: the functions take no context."""
import random
import struct

import numpy as np
import pytest

import __graft_entry__ as G

G.build()
from gappadder_amd import _lib as B                      # noqa: E402
from gappadder_amd import bam_io, textio                 # noqa: E402
from gappadder_amd.device_collect import DeviceCollector  # noqa: E402


def _bam_record(rng, names, qname, flag, with_seq=True, star_qual=False, cigar=()):
    ref = rng.choice([-1] + list(range(len(names))))
    mref = rng.choice([-1, ref] + list(range(len(names))))
    l_seq = rng.choice([0, 1, 2, 7, 100, 151]) if with_seq else 0
    seq = bytes(rng.randrange(256) for _ in range((l_seq + 1) // 2))
    qual = bytes([0xFF] * l_seq) if star_qual else bytes(rng.randrange(0, 60) for _ in range(l_seq))
    cig = b"".join(struct.pack("<I", (n << 4) | op) for n, op in cigar)
    body = struct.pack("<iiBBHHHiiii", ref, rng.randrange(-1, 10 ** 9), len(qname) + 1, rng.randrange(256), 4680, len(cigar), flag, l_seq, mref,
                       rng.randrange(-1, 10 ** 9), rng.randrange(-10 ** 6, 10 ** 6)) + qname + b"\0" + cig + seq + qual
    body += b"XTA" + b"U"            # an optional field behind the mandatory ones
    return struct.pack("<i", len(body)) + body


def test_bam_records_text_equals_the_per_record_columns():
    rng = random.Random(11)
    names = ["scf1", "chr_two", "x"]
    recs = []
    for i in range(400):
        q = ("read%d/%d" % (rng.randrange(10 ** 6), i)).encode()
        recs.append(_bam_record(rng, names, q, rng.choice([77, 141, 4, 8, 12, 128, 129, 0, 65535]), with_seq=i % 9 != 0, star_qual=i % 5 == 0,
                                cigar=[(rng.randrange(1, 300), rng.randrange(9)) for _ in range(rng.choice([0, 0, 1, 3]))]))
    stream = b"".join(recs)
    rb = np.cumsum([0] + [len(r) for r in recs[:-1]]).astype(np.uint64)
    cols = bam_io.BamCols(stream, rb, names)
    want_sam = "".join("\t".join(cols[i] + list(cols.seq_qual(i))) + "\n" for i in range(len(recs)))
    want_fq = ""
    for line in want_sam.splitlines():
        f = line.split()
        want_fq += "@" + f[0] + ("_2\n" if int(f[1]) > 128 else "_1\n") + f[9] + "\n+\n" + f[10] + "\n"
    # a gathered subset in another order, as gf_bam_fetch hands it over
    sel = [5, 3, 399, 0, 77]
    sam, fq = textio.bam_records_text(np.frombuffer(stream, dtype=np.uint8), rb, names)
    assert sam.decode("latin-1") == want_sam and fq.decode("latin-1") == want_fq
    blob = b"".join(recs[i] for i in sel)
    sam2, _ = textio.bam_records_text(np.frombuffer(blob, dtype=np.uint8), np.cumsum([0] + [len(recs[i]) for i in sel[:-1]]), names)
    assert sam2.decode("latin-1") == "".join(want_sam.splitlines(True)[i] for i in sel)
    assert textio.bam_records_text(np.zeros(0, dtype=np.uint8), [], names) == (b"", b"")


def test_bam_records_text_refuses_bytes_that_are_no_record():
    rng = random.Random(2)
    rec = _bam_record(rng, ["a"], b"q", 77)
    for bad in (rec[:40], rec[:-1], struct.pack("<i", 8) + rec[4:]):
        with pytest.raises(B.GapFillError) as e:
            textio.bam_records_text(np.frombuffer(bad, dtype=np.uint8), [0], ["a"])
        assert e.value.code == B.GF_E_FORMAT
    two = struct.pack("<iii", 40, 5, 0) + rec[12:]        # refID 5 of a one-name header
    with pytest.raises(B.GapFillError):
        textio.bam_records_text(np.frombuffer(two, dtype=np.uint8), [0], ["a"])


def test_fastq_records_text_equals_the_per_record_rewrite(tmp_path):
    rng = random.Random(5)
    files = []
    for m in range(2):
        recs = []
        for i in range(300):
            head = rng.choice(["@r%d/%d" % (i, m + 1), "@r%d extra words" % i, "@r%d/%d more/x" % (i, m + 1), "@", "@/x", "  @lead%d" % i, "@r%d\t tab" % i])
            seq = "".join(rng.choice("ACGTN") for _ in range(rng.randrange(0, 120))) + rng.choice(["", " ", "\r", " \t"])
            qual = "".join(chr(rng.randrange(33, 74)) for _ in range(len(seq.strip()))) + rng.choice(["", "\r"])
            recs.append("%s\n%s\n+%s\n%s\n" % (head, seq, rng.choice(["", "r%d" % i]), qual))
        recs.append("@cut\nACG")          # a record cut short at the end of the file
        files.append("".join(recs).encode())
    bounds = []
    for f in files:
        # record boundaries: every fourth line start
        starts = [0]
        nl = [i for i, c in enumerate(f) if c == 10]
        for j in range(3, len(nl), 4):
            starts.append(nl[j] + 1)
        if starts[-1] >= len(f):
            starts.pop()
        bounds.append((starts, starts[1:] + [len(f)]))
    which, begin, end = [], [], []
    for _ in range(800):
        m = rng.randrange(2)
        i = rng.randrange(len(bounds[m][0]))
        which.append(m); begin.append(bounds[m][0][i]); end.append(bounds[m][1][i])
    which.append(0); begin.append(10); end.append(10)      # an empty slice
    suffix = (b"_1", b"_2")
    want = [DeviceCollector._record(files[m], b, e, suffix[m]) for m, b, e in zip(which, begin, end)]
    out, out_end, ids, ids_end = textio.fastq_records_text(files, begin, end, which, suffix, want_ids=True)
    out, ids = out.tobytes(), ids.tobytes()
    assert out == b"".join(t for _, t in want) and ids == b"".join(r for r, _ in want)
    assert list(out_end) == list(np.cumsum([len(t) for _, t in want])) and list(ids_end) == list(np.cumsum([len(r) for r, _ in want]))
    # the same records read by position from open files instead of images in memory
    paths = []
    for m, f in enumerate(files):
        paths.append(str(tmp_path / ("m%d.fq" % m)))
        open(paths[-1], "wb").write(f)
    with open(paths[0], "rb") as f0, open(paths[1], "rb") as f1:
        o2, e2, i2, ie2 = textio.fastq_records_text([f0, f1], begin, end, which, suffix, want_ids=True)
        assert o2.tobytes() == out and i2.tobytes() == ids and list(e2) == list(out_end) and list(ie2) == list(ids_end)
        with pytest.raises(B.GapFillError):
            textio.fastq_records_text([f0, f1], [0], [len(files[0]) + 1], [0], suffix)
    out2, end2 = textio.fastq_records_text(files, begin[:3], end[:3], which[:3], suffix)
    assert out2.tobytes() == b"".join(t for _, t in want[:3]) and len(end2) == 3
    out3, end3 = textio.fastq_records_text(files, [], [], [], suffix)
    assert len(out3) == 0 and len(end3) == 0
    with pytest.raises(B.GapFillError):
        textio.fastq_records_text(files, [0], [len(files[0]) + 1], [0], suffix)


def test_bridging_reads_in_one_library_call_equal_the_python_definition():
    """gf_bridging_reads (host code of the library, all gaps of a round in one call) against gappadder_amd/assemble_gaps.py::bridging_reads — the
    definition of the rescue round's alignment stand-in (assemble_gaps.py:166-217): reads that bridge two contigs, reads inside one contig with and
    without sequencing errors, strangers, reverse-complemented reads, a seed repeated more than eight times in a contig, N in reads and contigs
    (the definition then runs its text look-up instead of the 2-bit sort), lower case, reads shorter than the seed, gaps with one contig."""
    import random
    from gappadder_amd import assemble_gaps as AG
    rng = random.Random(5)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    rc = lambda x: x[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))
    items = []
    for g in range(60):
        genome = rnd(rng.randint(600, 2500))
        if g % 6 == 0:
            genome = genome[:200] + "ACGTTGCA" * 20 + genome[200:]                    # a low-complexity stretch: windows that repeat > 8 times
        cuts = sorted(rng.sample(range(100, len(genome) - 100), rng.randint(1, 4)))
        contigs, a = [], 0
        for c in cuts + [len(genome)]:
            piece = genome[a:c - rng.randint(0, 40)]                                 # holes between the contigs
            if len(piece) > 40:
                contigs.append(("c%d" % len(contigs), rc(piece) if rng.random() < 0.5 else piece))
            a = c
        if g % 9 == 0:
            contigs = contigs[:1]                                                     # one contig: nothing to bridge
        if g % 7 == 0 and contigs:
            n, s_ = contigs[0]
            contigs[0] = (n, s_[:30] + "N" + s_[31:])
        reads = {}
        for r in range(rng.randint(5, 60)):
            kind = rng.random()
            at = rng.randint(0, len(genome) - 151)
            s_ = genome[at:at + rng.choice((150, 150, 100, 25))]
            if kind < 0.25:
                p = rng.randint(0, len(s_) - 1)
                s_ = s_[:p] + rng.choice("ACGT") + s_[p + 1:]
            elif kind < 0.35:
                s_ = rnd(len(s_))
            elif kind < 0.45:
                s_ = s_[:len(s_) // 2] + rnd(len(s_) - len(s_) // 2)                  # chimeric: clipped wherever it lands
            if rng.random() < 0.5:
                s_ = rc(s_)
            if rng.random() < 0.1:
                s_ = s_.lower()
            if rng.random() < 0.05 and len(s_) > 40:
                s_ = s_[:40] + "N" + s_[41:]
            reads["r%d" % r] = s_
        items.append(([(n, s_.upper()) for n, s_ in contigs], reads))
    got = AG.bridging_reads_batch(items)
    want = [AG.bridging_reads(c, r) for c, r in items]
    assert got == want
    assert sum(len(w) for w in want) > 40 and any(not w for w in want) and sum(1 for w in want if w) > 20
    for seed, budget_items in ((20, items[:10]), (32, items[10:20])):               # other seed lengths
        assert AG.bridging_reads_batch(budget_items, seed_len=seed) == [AG.bridging_reads(c, r, seed) for c, r in budget_items]
    assert AG.bridging_reads_batch([]) == [] and AG.bridging_reads_batch([([], {})]) == [[]]
