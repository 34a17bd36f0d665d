"""GPU BAM ingest (gf_bgzf_inflate + gf_bam_pack): the BGZF blocks inflated on the device equal zlib's output
byte for byte, and the records decoded from a BAM equal the records the SAM text path gives for the same alignments
(sam_io.decode = the host restatement of collect_reads_for_gaps.py:76-91, pinned to the reference's lists by the golden tests)."""
import numpy as np
import pytest

import bam_util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gf():
    from gappadder_amd.hip_api import GapFill
    g = GapFill(0)
    yield g
    g.close()


def _payloads():
    rng = np.random.RandomState(7)
    skew = bytes(np.minimum(255, rng.geometric(0.03, 150_000)).astype(np.uint8))   # skewed byte histogram: code lengths up to 15
    dna = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, 300_000)])
    text = b"".join(b"read%d\t%d\tchr%d\t%d\t60\t150M\t=\n" % (i, 99 + (i & 64), i % 23, i * 37) for i in range(20000))
    return {
        "empty": b"",
        "one_byte": b"x",
        "random": bytes(rng.randint(0, 256, 200_000, dtype=np.uint8)),          # incompressible: zlib falls back to stored blocks
        "dna": dna,
        "text": text,
        "runs": b"A" * 70000 + b"ab" * 40000 + b"xyz" * 30000 + bytes(1000),    # matches with distance 1, 2, 3 (< length)
        "far": dna[:32768] + dna[:32768] + dna[100:20000],                      # matches at the maximum distance
        "skew": skew,
    }


@pytest.mark.parametrize("levels", [(6,), (1,), (9,), (0,), ("fixed",), ("huffman",), ("rle",), (0, 6, "fixed", 9, 1, "huffman", "rle")])
def test_bgzf_inflate_equals_zlib(gf, levels):
    for name, data in _payloads().items():
        for block, seed in ((0xFF00, None), (20000, 11), (700, 5)):
            if block == 700 and len(data) > 100_000:
                continue
            z = U.bgzf_compress(data, block=block, seed=seed, levels=levels)
            assert U.bgzf_decompress(z) == data
            out, used = gf.bgzf_inflate(z)
            assert used == len(z), name
            assert out.tobytes() == data, (name, levels, block)


def test_bgzf_streaming_carry_and_partial_block(gf):
    data = _payloads()["text"]
    z = U.bgzf_compress(data, block=9000, seed=3, eof=False)
    cut = len(z) - 10                                  # the last block is incomplete
    out, used = gf.bgzf_inflate(z[:cut], carry=b"HELLO")
    assert used < cut and out[:5].tobytes() == b"HELLO"
    assert data.startswith(out[5:].tobytes()) and len(out) - 5 < len(data)
    out2, used2 = gf.bgzf_inflate(z[used:])
    assert used + used2 == len(z) and out[5:].tobytes() + out2.tobytes() == data
    out3, used3 = gf.bgzf_inflate(z[:10])              # less than a header
    assert used3 == 0 and len(out3) == 0


def test_corrupt_bgzf_is_an_error_not_garbage(gf):
    from gappadder_amd import _lib as B
    data = _payloads()["dna"][:50000]
    z = bytearray(U.bgzf_compress(data, levels=(6,)))
    for at in (30, len(z) // 2, len(z) - 40):
        bad = bytearray(z)
        bad[at] ^= 0x10
        with pytest.raises(B.GapFillError) as e:
            gf.bgzf_inflate(bytes(bad))
        assert e.value.code == B.GF_E_FORMAT
    bad = bytearray(z)
    bad[0] = 0x1e                                        # not a gzip member
    with pytest.raises(B.GapFillError) as e:
        gf.bgzf_inflate(bytes(bad))
    assert e.value.code == B.GF_E_FORMAT
    out, _ = gf.bgzf_inflate(bytes(z))                   # the ctx is still usable
    assert out.tobytes() == data


def _golden_sams():
    from golden_util import CASES, Case
    for name in CASES:
        case = Case(name)
        for lib in case.libs:
            yield case, lib["sam"]


def _check_bam(gf, sam_text, fai_names, ref_names, pieces, **bgzf):
    from gappadder_amd import bam_io, sam_io
    lines = [l for l in sam_text.splitlines() if l and l[0] != "@"]
    exp, cols = sam_io.decode(lines, {n: i for i, n in enumerate(fai_names)})
    bam = U.bgzf_compress(U.sam_to_bam_stream(lines, ref_names, [10 ** 6] * len(ref_names)), **bgzf)
    cuts = [0] + sorted(pieces(len(bam))) + [len(bam)]
    got, gcols, base = [], [], 0
    for recs, bc in bam_io.decode_chunks(gf, (bam[a:b] for a, b in zip(cuts, cuts[1:])), fai_names):
        assert np.array_equal(recs["read"], np.arange(len(recs)))
        got.append(recs)
        gcols += [(base + i, bc[i] + list(bc.seq_qual(i))) for i in list(range(0, len(recs), max(1, len(recs) // 50))) + [len(recs) - 1]]
        base += len(recs)
    got = np.concatenate(got) if got else np.zeros(0, dtype=exp.dtype)
    assert len(got) == len(exp)
    a, b = got.copy(), exp.copy()
    a["read"] = 0
    b["read"] = 0
    assert a.tobytes() == b.tobytes()
    for i, c in gcols:      # the 11 mandatory columns as `samtools view` would print them
        assert c[:9] == cols[i] and c[9:] == lines[i].split("\t")[9:11], (i, c, lines[i][:200])
    return len(got)


def test_bam_records_equal_the_sam_text_path_on_the_golden_alignments(gf):
    total = 0
    for case, sam in _golden_sams():
        extra = sam + ("qx\t77\t*\t0\t0\t*\t*\t0\t0\tACGT\tIIII\n"                                  # unmapped pair, no reference
                       "qy\t99\t%s\t17\t255\t5S90M5H\tnosuch\t250\t-321\tACGTN\t*\tNM:i:3\tXA:Z:foo\n" % case.fai_names[0])
        names = list(case.fai_names) + ["nosuch"]           # the BAM knows one reference the .fai does not
        total += _check_bam(gf, extra, case.fai_names, names, lambda n: [], levels=(6,))
        # reference order of the BAM differs from the .fai's; file cut at awkward places; mixed block kinds and sizes
        total += _check_bam(gf, extra, case.fai_names, names[::-1], lambda n: [1, 17, 18, n // 3, n // 3 + 1, n - 5],
                            block=5000, seed=9, levels=(0, 6, "fixed", 9))
    assert total > 10000


def test_bam_in_many_small_pieces_and_records_longer_than_a_segment(gf):
    case, sam = next(_golden_sams())
    lines = [l for l in sam.splitlines() if l and l[0] != "@"][:3000]
    ref = case.fai_names[0]
    rng = np.random.RandomState(5)
    long_seq = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, 250_000)]).decode()   # one record ~ 375 KB, spans 5 segments
    lines.insert(1000, "long1\t0\t%s\t500\t60\t100S249800M100S\t*\t0\t0\t%s\t*" % (ref, long_seq))
    lines.insert(2000, "long2\t16\t%s\t900\t0\t250000M\t=\t5\t-7\t%s\t*" % (ref, long_seq))
    text = "\n".join(lines) + "\n"
    n = _check_bam(gf, text, case.fai_names, case.fai_names, lambda n: list(range(4096, n, 4096)), block=30000, seed=2, levels=(6, 1))
    assert n == len(lines)


def test_bam_pack_from_host_bytes_and_argument_errors(gf):
    from gappadder_amd import _lib as B, bam_io
    case, sam = next(_golden_sams())
    lines = [l for l in sam.splitlines() if l and l[0] != "@"][:500]
    stream = U.sam_to_bam_stream(lines, case.fai_names, [10 ** 6] * len(case.fai_names))
    names, first = bam_io.parse_header(stream)
    assert names == list(case.fai_names)
    rmap = np.arange(len(names), dtype=np.uint32)
    recs, rb, used = gf.bam_pack(stream, first, rmap)
    assert len(recs) == len(lines) and used == len(stream) and int(rb[0]) == first
    recs2, _, used2 = gf.bam_pack(stream[:-7], first, rmap)              # the last record is incomplete
    assert len(recs2) == len(lines) - 1 and used2 == int(rb[-1])
    recs3, _, used3 = gf.bam_pack(stream[:first], first, rmap)           # header only
    assert len(recs3) == 0 and used3 == first
    with pytest.raises(B.GapFillError) as e:
        gf.bam_pack(None, first, rmap, n_bytes=len(stream) + 1)          # not the stream left on the device
    assert e.value.code == B.GF_E_STATE
    bad = bytearray(stream)
    bad[int(rb[100]):int(rb[100]) + 4] = (5).to_bytes(4, "little")     # a record cannot be shorter than its fixed fields
    with pytest.raises(B.GapFillError) as e:
        gf.bam_pack(bytes(bad), first, rmap)
    assert e.value.code == B.GF_E_FORMAT
    # slices of the stream the GPU holds
    gf.bam_pack(stream, first, rmap)
    got = gf.bam_fetch([0, int(rb[3]), len(stream)], [4, int(rb[4]), len(stream)]).tobytes()
    assert got == stream[:4] + stream[int(rb[3]):int(rb[4])]
    with pytest.raises(B.GapFillError) as e:
        gf.bam_fetch([0], [len(stream) + 1])
    assert e.value.code == B.GF_E_INVAL


def test_tagging_the_records_left_on_the_gpu_equals_tagging_a_host_copy(gf):
    """gf_tag_alignments_bam / gf_tag_low_mapq_bam (records stay in HBM after gf_bam_pack) == the host-array variants."""
    from gappadder_amd import _lib as B, bam_io, sam_io
    from golden_util import CASES, Case
    import records_util as RU
    case = Case(CASES[0])
    lib = case.libs[0]
    lines = [l for l in lib["sam"].splitlines() if l and l[0] != "@"]
    from oracle import gp_oracle as O
    names = list(case.fai_names)
    gaps = RU.gaps_array(names, O.gap_positions(case.fasta_records(), case.meta["min_gap"]))
    stream = U.sam_to_bam_stream(lines, names, [10 ** 6] * len(names))
    _, first = bam_io.parse_header(stream)
    gf.set_gaps(gaps, len(names))
    recs, _, _ = gf.bam_pack(stream, first, np.arange(len(names), dtype=np.uint32))
    a = gf.tag_alignments(recs, int(lib["is"]), int(lib["sd"]))
    b = gf.tag_alignments_bam(len(recs), int(lib["is"]), int(lib["sd"]))
    assert len(a) > 50 and a.tobytes() == b.tobytes()
    rows = sorted((int(recs[h["rec"]]["mate_ref"]), int(recs[h["rec"]]["mate_pos"]), int(gaps[h["gap"]]["scaffold"]),
                   int(gaps[h["gap"]]["idx_in_scaffold"])) for h in a if h["kind"] == B.KIND_DISCORDANT)
    table = RU.dpos_array(rows).astype(B.DPOS)
    assert gf.tag_low_mapq(recs, table).tobytes() == gf.tag_low_mapq_bam(len(recs), table).tobytes()


def test_bgzf_inflate_fuzz(gf):
    """40 seeded random payloads (mixtures of noise, runs, periodic text, sparse bytes) x random block sizes and DEFLATE
    flavours: inflated on the GPU == zlib."""
    kinds = [6, 1, 9, 0, "fixed", "huffman", "rle"]
    for seed in range(40):
        rng = np.random.RandomState(1000 + seed)
        parts = []
        for _ in range(rng.randint(1, 12)):
            n = int(rng.randint(1, 60000))
            t = rng.randint(5)
            if t == 0:
                parts.append(bytes(rng.randint(0, 256, n, dtype=np.uint8)))
            elif t == 1:
                parts.append(bytes([int(rng.randint(256))]) * n)
            elif t == 2:
                w = bytes(rng.randint(65, 91, int(rng.randint(1, 300)), dtype=np.uint8))
                parts.append((w * (n // len(w) + 1))[:n])
            elif t == 3:
                a = np.zeros(n, dtype=np.uint8)
                a[rng.randint(0, n, max(1, n // 50))] = rng.randint(1, 256)
                parts.append(bytes(a))
            else:
                parts.append(bytes(np.frombuffer(b"ACGTN", np.uint8)[rng.randint(0, 5, n)]))
        data = b"".join(parts)
        levels = tuple(kinds[i] for i in rng.randint(0, len(kinds), 5))
        z = U.bgzf_compress(data, block=int(rng.randint(100, 0xFF00)), seed=seed, levels=levels, eof=bool(seed & 1))
        out, used = gf.bgzf_inflate(z)
        assert used == len(z) and out.tobytes() == data, (seed, levels)


def test_bam_chain_survives_records_that_contain_plausible_records(gf):
    """The first record of a 64-KiB segment is GUESSED and then verified against the chain from the previous segment.  Here a
    record's quality bytes are themselves 300 KB of well-formed BAM records (so every guess inside it locks onto fake records) —
    the decoded chain must still be the true one."""
    import struct
    from gappadder_amd import bam_io
    case, sam = next(_golden_sams())
    lines = [l for l in sam.splitlines() if l and l[0] != "@"][:2500]
    names = list(case.fai_names)
    idx = {n: i for i, n in enumerate(names)}
    real = [U.sam_record(l, idx) for l in lines]
    decoy = b"".join(real[:1500])[:300_000]                  # valid records back to back, used as payload bytes
    l_seq = len(decoy)
    body = struct.pack("<iiBBHHHiiii", 0, 99, 6, 60, 4681, 1, 0, l_seq, -1, -1, 0) + b"decoy\0" + struct.pack("<I", (l_seq << 4) | 0)
    body += bytes((l_seq + 1) // 2) + decoy                  # SEQ (all '='), then QUAL = the decoy records
    big = struct.pack("<i", len(body)) + body
    header = U.sam_to_bam_stream([], names, [10 ** 6] * len(names))
    stream = header + b"".join(real[:700]) + big + b"".join(real[700:1900]) + big + b"".join(real[1900:])
    _, first = bam_io.parse_header(stream)
    true_rb, o = [], first
    while o < len(stream):
        true_rb.append(o)
        o += 4 + struct.unpack_from("<i", stream, o)[0]
    assert o == len(stream) and len(true_rb) == len(real) + 2
    recs, rb, used = gf.bam_pack(stream, first, np.arange(len(names), dtype=np.uint32))
    assert used == len(stream) and np.array_equal(rb, np.array(true_rb, dtype=np.uint64))
    assert int(recs[700]["pos"]) == 100 and int(recs[700]["mapq"]) == 60 and int(recs[1901]["pos"]) == 100
