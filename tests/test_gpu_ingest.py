"""GPU ingest (gf_fastq_pack): FASTQ text parsed and packed on the device vs the host restatement (the line
machine of fastq_io + gf_pack_reads' layout, which tests/test_oracle_golden.py pins to KmerUtils), bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gf():
    from gappadder_amd.hip_api import GapFill
    g = GapFill(0)
    yield g
    g.close()


def _expect(text, L):
    from gappadder_amd.hip_api import GapFill
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines = lines[:-1]
    n = len(lines) // 4
    seqs = [lines[4 * i + 1].rstrip(b"\r")[:L] for i in range(n)]
    blob = b"".join(s.ljust(L, b"N") for s in seqs)
    packed, nm = GapFill.pack_reads(blob, L, with_mask=True)
    hdr, off = [], 0
    for i, l in enumerate(lines[:4 * n]):
        if i % 4 == 0:
            hdr.append(off)
        off += len(l) + 1
    return packed, nm, np.array(hdr, dtype=np.uint64), n


def _fastq(rng, n, L, var=False, with_n=True):
    lut = np.frombuffer(b"ACGTacgtNn.", np.uint8) if with_n else np.frombuffer(b"ACGT", np.uint8)
    out = []
    for i in range(n):
        ln = rng.randint(1, L + 1) if var else L
        p = np.full(ln, 0, np.int64)
        p[:] = rng.randint(0, 4, ln)
        if with_n:
            m = rng.rand(ln) < 0.03
            p[m] = rng.randint(4, len(lut), m.sum())
        s = lut[p].tobytes()
        out.append(b"@r%d/%d some comment\n" % (i * 7919 % 100003, 1 + i % 2) + s + b"\n+\n" + b"I" * ln + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("n,L,var", [(1, 150, False), (5000, 150, False), (3000, 100, True), (70000, 151, False), (257, 33, True)])
def test_fastq_pack_matches_host(gf, n, L, var):
    rng = np.random.RandomState(n + L)
    text = _fastq(rng, n, L, var)
    packed, nm, hdr, st = gf.fastq_pack(text, L)
    ep, em, eh, en = _expect(text, L)
    assert st == 0 and len(packed) == en == n
    assert np.array_equal(packed, ep) and np.array_equal(nm, em) and np.array_equal(hdr, eh)
    # ids cut from the host's text at the returned offsets (run_multi_threads_discordant.py:212-214)
    i = n // 2
    assert text[int(hdr[i]):].split(None, 1)[0].split(b"/")[0][1:] == b"r%d" % (i * 7919 % 100003)


def test_fastq_pack_edge_cases(gf):
    rng = np.random.RandomState(3)
    L = 50
    text = _fastq(rng, 40, L)
    # last line without its newline
    packed, nm, hdr, st = gf.fastq_pack(text[:-1], L)          # (status bit 8 says so: a whole file is complete, a piece of one may be cut there)
    ep, em, eh, en = _expect(text, L)
    assert st == 8 and np.array_equal(packed, ep) and np.array_equal(nm, em) and np.array_equal(hdr, eh)
    # CRLF line ends
    crlf = text.replace(b"\n", b"\r\n")
    packed, nm, hdr, st = gf.fastq_pack(crlf, L)
    assert st == 0 and np.array_equal(packed, ep) and np.array_equal(nm, em)
    # a trailing partial record is ignored and flagged
    packed, nm, hdr, st = gf.fastq_pack(text + b"@tail\nACGT\n", L)
    assert st == 2 and np.array_equal(packed, ep)
    # sequence lines longer than read_len are truncated and flagged
    packed, nm, hdr, st = gf.fastq_pack(text, 20)
    ep20, em20, _, _ = _expect(text, 20)
    assert st == 1 and np.array_equal(packed, ep20) and np.array_equal(nm, em20)
    # empty input
    packed, nm, hdr, st = gf.fastq_pack(b"", L)
    assert len(packed) == 0 and st == 0


def test_fastq_pack_feeds_the_screen(gf):
    """Reads ingested on the GPU give the same screen hits as reads packed on the host."""
    import synth_small as S
    c = S.small_case(seed=4, n_pairs=3000, L=100, insert=250)
    L = c["L"]
    blob = c["reads_blob"]
    n = len(blob) // L
    text = b"".join(b"@q%d\n" % i + blob[i * L:(i + 1) * L] + b"\n+\n" + b"#" * L + b"\n" for i in range(n))
    packed, nm, hdr, st = gf.fastq_pack(text, L)
    from gappadder_amd.hip_api import GapFill
    p2, nm2 = GapFill.pack_reads(blob, L, with_mask=True)
    assert st == 0 and np.array_equal(packed, p2) and np.array_equal(nm, nm2)
    gf.set_gaps(c["gaps"], c["n_scaffolds"], c["flanks"])
    a = gf.screen_reads(packed, L, 31, 1, n_mask=nm)
    b = gf.screen_reads(p2, L, 31, 1, n_mask=nm2)
    assert len(a) > 20 and np.array_equal(a, b)


def test_sam_pack_matches_host_decoder_on_golden_sam(gf):
    """gf_sam_pack (SAM text parsed on the GPU) == sam_io.decode (the host restatement of collect_reads_for_gaps.py:76-91) on
    every SAM fixture, plus header lines, short lines, CRLF-free odd spacing, unknown reference names and '=' / named RNEXT."""
    from golden_util import CASES, Case
    from gappadder_amd import sam_io
    total = 0
    for name in CASES:
        case = Case(name)
        sidx = {n: i for i, n in enumerate(case.fai_names)}
        for lib in case.libs:
            text = lib["sam"]
            extra = ("@HD\tVN:1.6\n@SQ\tSN:x\tLN:5\n" + text +
                     "short\t4\t*\n" +
                     "q1  99 %s   17  255 5S90M5H  nosuch 250   -321  ACGT IIII\n" % case.fai_names[0] +
                     "q2\t147\tnosuch\t5\t300\t*\t%s\t9\t0\t*\t*" % case.fai_names[0])          # no trailing newline
            exp, cols = sam_io.decode(extra.splitlines(), sidx)
            recs, lb = gf.sam_pack(extra.encode(), case.fai_names)
            assert len(recs) == len(exp)
            assert recs.tobytes() == exp.tobytes()
            b = extra.encode()
            for i in (0, len(recs) // 2, len(recs) - 1):
                assert b[int(lb[i]):].split(None, 1)[0].decode() == cols[i][0]
            total += len(recs)
    assert total > 5000
