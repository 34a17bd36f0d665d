"""The entry points that turn BAM + FASTQ files into a library resident in HBM (csrc/resident.hip: gf_fastq_index_dev, gf_bam_append_dev,
gf_read_join_dev, gf_fetch_slices, gf_gather_rows_dev) against their definitions in Python — the reference's join of alignments and reads
BY NAME (run_multi_threads_discordant.py:153-185, 209-241; id cut :212-214) restated as indices."""
import ctypes as C
import struct

import numpy as np
import pytest

import bam_util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gf():
    from gappadder_amd.hip_api import GapFill
    g = GapFill(0)
    yield g
    g.close()


def _ref_id(header_line):
    """The reference's cut of a FASTQ header line (run_multi_threads_discordant.py:212-214)."""
    return header_line.split()[0].split("/")[0][1:].rstrip()


def _fastq(ids, L, rng, tails=None):
    recs = []
    for i, rid in enumerate(ids):
        n = L - (i % 5 if tails else 0)
        seq = "".join("ACGTN"[x] for x in rng.integers(0, 5 if i % 7 == 0 else 4, size=n))
        recs.append("@%s\n%s\n+\n%s\n" % (rid, seq, "I" * n))
    return "".join(recs)


def _index(gf, text, L):
    import torch
    from gappadder_amd import _lib as B
    lib = B.lib()
    raw = text.encode()
    d_text = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).cuda()
    cap = raw.count(b"\n") // 4 + 2
    d_p = torch.zeros(cap * lib.gf_packed_read_bytes(L), dtype=torch.uint8, device="cuda")
    d_h = torch.zeros(cap + 1, dtype=torch.int64, device="cuda")
    d_c = torch.zeros(4, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert lib.gf_fastq_pack_dev(gf.handle, d_text.data_ptr(), len(raw), L, d_p.data_ptr(), cap, None, d_h.data_ptr(), d_c.data_ptr(), d_c.data_ptr() + 8) == 0
    gf.sync()
    n = int(d_c[0])
    d_id = torch.zeros(max(1, n), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert lib.gf_fastq_index_dev(gf.handle, d_text.data_ptr(), len(raw), d_h.data_ptr(), n, d_id.data_ptr(), d_c.data_ptr() + 16) == 0
    gf.sync()
    return n, d_id[:n], int(d_c[2]) & 0xFFFFFFFF


def test_fastq_ids_hash_like_the_reference_cuts_them(gf):
    """Headers that the reference maps to the same id ('@r7/1', '@r7/2 extra', '@r7 1:N:0', '@r7\\t…') hash alike; different ids differ;
    the longest sequence line is reported."""
    rng = np.random.default_rng(1)
    heads = ["r7/1", "r7/2 extra words", "r7 1:N:0:ACGT", "r7\tx", "r70/1", "r/7", "R7/1", "r7:1/1", "r7"]
    n, ids, mx = _index(gf, _fastq(heads, 40, rng, tails=True), 40)
    got = ids.cpu().numpy()
    assert n == len(heads) and mx == 40
    want = [_ref_id("@" + h) for h in heads]
    for i in range(n):
        for j in range(n):
            assert (got[i] == got[j]) == (want[i] == want[j]), (heads[i], heads[j])


def _bam(records, names, lens):
    """records: [(qname, flag, ref, pos1, mapq, cigar, mref, mpos1, tlen)] -> BGZF bytes."""
    lines = ["%s\t%d\t%s\t%d\t%d\t%s\t%s\t%d\t%d\t*\t*" % (q, fl, names[r] if r >= 0 else "*", p, mq, cg,
                                                       "*" if mr < 0 else "=" if mr == r else names[mr], mp, tl)
             for (q, fl, r, p, mq, cg, mr, mp, tl) in records]
    return bam_util.bgzf_compress(bam_util.sam_to_bam_stream(lines, names, lens), block=3000, seed=3, levels=(6, 1))


def test_bam_records_append_to_a_resident_array_with_names_and_hashes_and_join_to_fastq_records(gf):
    import torch
    from gappadder_amd import _lib as B
    from gappadder_amd import bam_io
    lib = B.lib()
    rng = np.random.default_rng(5)
    names, lens = ["sA", "sB", "sC"], [50000, 40000, 30000]
    n_pairs = 700
    fq_ids = ["read%d" % (i * 3) for i in range(n_pairs)]
    recs = []
    for i in range(1500):
        pair = int(rng.integers(0, n_pairs + 60))                       # some QNAMEs have no FASTQ record
        q = "read%d" % (pair * 3) if pair < n_pairs else "orphan%d" % pair
        ref = int(rng.integers(-1, 2))                                   # -1: unmapped; scaffold sC never appears
        flag = (0x40 if rng.integers(2) else 0x80) | 1 | (4 if ref < 0 else 0)
        recs.append((q, flag, ref, int(rng.integers(1, 30000)) if ref >= 0 else 0, int(rng.choice([0, 30, 60])), "10S90M" if i % 9 == 0 else "100M",
                     ref, int(rng.integers(1, 30000)) if ref >= 0 else 0, int(rng.integers(-500, 500))))
    recs.sort(key=lambda r: (r[2] if r[2] >= 0 else 99, r[3]))
    data = _bam(recs, names, lens)
    # two pieces, the second appended behind the first
    n_total, name_total = 0, 0
    cap = 4096
    d_recs = torch.zeros(cap * 4, dtype=torch.int64, device="cuda")
    d_qh = torch.zeros(cap, dtype=torch.int64, device="cuda")
    d_noff = torch.zeros(cap + 1, dtype=torch.int64, device="cuda")
    d_names = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    d_seen = torch.zeros(len(names), dtype=torch.int32, device="cuda")
    d_rb = torch.zeros(cap + 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    nr, nb, used = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    file_carry, rec_carry, first_done = b"", b"", False
    ref_map = np.arange(len(names), dtype=np.uint32)
    for piece in (data[:len(data) // 3], data[len(data) // 3:]):
        buf = file_carry + piece
        n_stream, consumed = gf.bgzf_inflate(buf, rec_carry, want_host=False)
        file_carry = buf[consumed:]
        first = 0
        if not first_done:
            _, first = bam_io.parse_header(gf.bam_fetch([0], [n_stream]).tobytes())
            first_done = True
        # a capacity that is too small is reported, nothing is written
        assert lib.gf_bam_append_dev(gf.handle, n_stream, first, B._p(ref_map), 3, d_recs.data_ptr(), n_total, n_total + 1, d_qh.data_ptr(), d_names.data_ptr(),
                                     name_total, 1 << 16, d_noff.data_ptr(), d_seen.data_ptr(), 3, None, C.byref(nr), C.byref(nb), C.byref(used)) == B.GF_E_NOSPACE
        assert nr.value > 1
        assert lib.gf_bam_append_dev(gf.handle, n_stream, first, B._p(ref_map), 3, d_recs.data_ptr(), n_total, cap, d_qh.data_ptr(), d_names.data_ptr(),
                                     name_total, 1 << 16, d_noff.data_ptr(), d_seen.data_ptr(), 3, d_rb.data_ptr(), C.byref(nr), C.byref(nb), C.byref(used)) == 0
        # the records' offsets in the inflated stream: each starts with its own block_size, the next one follows it
        rb = d_rb[:nr.value].cpu().numpy().astype(np.uint64)
        sizes = np.frombuffer(gf.bam_fetch(rb, rb + np.uint64(4)).tobytes(), dtype="<i4").astype(np.int64)
        assert rb[0] == first and (rb[1:] == rb[:-1] + np.uint64(4) + sizes[:-1].astype(np.uint64)).all() and int(rb[-1]) + 4 + int(sizes[-1]) == used.value
        n_total += nr.value
        name_total += nb.value
        rec_carry = gf.bam_fetch([used.value], [n_stream]).tobytes() if used.value < n_stream else b""
    assert n_total == len(recs) and not file_carry and not rec_carry
    got = np.frombuffer(d_recs[:4 * n_total].cpu().numpy().tobytes(), dtype=B.ALNREC)
    noff = d_noff[:n_total + 1].cpu().numpy()
    blob = d_names[:name_total].cpu().numpy().tobytes()
    for i, (q, fl, r, p, mq, cg, mr, mp, tl) in enumerate(recs):
        g = got[i]
        assert blob[noff[i]:noff[i + 1]].decode() == q
        assert (int(g["flag"]), int(g["mapq"]), int(g["tlen"])) == (fl, mq, tl)
        assert int(g["ref"]) == (r if r >= 0 else 0xFFFFFFFF) and int(g["clipflag"]) == (1 if cg.startswith("10S") else 0)
        assert int(g["read"]) == 0xFFFFFFFF                              # no read before the join
    seen = d_seen.cpu().numpy()
    assert [int(x) & 1 for x in seen] == [1, 1, 0]
    assert [bool(int(x) & 2) for x in seen] == [any(r[2] == s and r[4] == 0 for r in recs) for s in range(3)]
    # names fetched for a few records in one gather
    pick = np.array([0, 7, n_total - 1], dtype=np.int64)
    b, e = noff[pick].astype(np.uint64), noff[pick + 1].astype(np.uint64)
    dst = np.zeros(int((e - b).sum()), dtype=np.uint8)
    nn = C.c_size_t(0)
    assert lib.gf_fetch_slices(gf.handle, d_names.data_ptr(), name_total, B._p(b), B._p(e), 3, B._p(dst), len(dst), C.byref(nn)) == 0
    assert dst.tobytes().decode() == "".join(recs[i][0] for i in pick)
    # the join: QNAME -> number of the FASTQ record with that id; flag 0x40 -> mate 0, else mate 1; unknown names -> no read
    n, d_id, _ = _index(gf, _fastq([x + "/1" for x in fq_ids], 30, rng), 30)
    assert n == n_pairs
    d_stats = torch.zeros(4, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    assert lib.gf_read_join_dev(gf.handle, d_id.data_ptr(), n, d_recs.data_ptr(), d_qh.data_ptr(), n_total, d_stats.data_ptr()) == 0
    gf.sync()
    got = np.frombuffer(d_recs[:4 * n_total].cpu().numpy().tobytes(), dtype=B.ALNREC)
    where = {x: i for i, x in enumerate(fq_ids)}
    n_orphan = 0
    for i, (q, fl, *_rest) in enumerate(recs):
        if q in where:
            assert int(got[i]["read"]) == 2 * where[q] + (0 if fl & 0x40 else 1), q
        else:
            assert int(got[i]["read"]) == 0xFFFFFFFF
            n_orphan += 1
    st = d_stats.cpu().numpy()
    assert int(st[0]) == 0 and int(st[1]) == n_orphan > 0
    # an id that occurs twice is counted
    n2, d_id2, _ = _index(gf, _fastq(["a/1", "b/1", "a/1", "c/1", "b/1"], 30, rng), 30)
    assert lib.gf_read_join_dev(gf.handle, d_id2.data_ptr(), n2, d_recs.data_ptr(), d_qh.data_ptr(), 0, d_stats.data_ptr()) == 0
    gf.sync()
    assert int(d_stats[0]) == 2


def test_a_record_without_a_read_recruits_nothing(gf):
    """gf_pool_keys_all_dev: a tagger hit on a record whose QNAME is in no FASTQ record (read = 0xFFFFFFFF) yields no key (the reference
    finds no FASTQ record for such a name)."""
    import torch
    from gappadder_amd import _lib as B
    lib = B.lib()
    recs = np.zeros(4, dtype=B.ALNREC)
    recs["read"] = [5, 0xFFFFFFFF, 8, 0xFFFFFFFF]
    th = np.zeros(4, dtype=B.TAGHIT)
    th["rec"], th["gap"], th["to_mate"] = [0, 1, 2, 3], [0, 1, 1, 0], [0, 0, 1, 1]
    d_recs = torch.from_numpy(recs.view(np.uint8).copy()).cuda()
    d_th = torch.from_numpy(th.view(np.uint8).copy()).cuda()
    d_cnt = torch.tensor([0, 4, 0, 0], dtype=torch.int32, device="cuda")     # screen hits, tagger hits, second-hop hits, keys
    d_keys = torch.zeros(16, dtype=torch.int64, device="cuda")
    d_dummy = torch.zeros(64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    p = d_cnt.data_ptr()
    assert lib.gf_pool_keys_all_dev(gf.handle, d_dummy.data_ptr(), p, 4, 1, d_recs.data_ptr(), d_th.data_ptr(), p + 4, 4, None, p + 8, 0, None,
                                    d_keys.data_ptr(), 16, p + 12) == 0
    gf.sync()
    n = int(d_cnt[3])
    keys = d_keys[:n].cpu().numpy().view(np.uint64)
    real = sorted(int(k) for k in keys if (int(k) >> 32) != 0xFFFFFFFF)
    assert real == sorted([(0 << 32) | 5, (1 << 32) | 9])


def test_rows_gathered_by_index(gf):
    import torch
    from gappadder_amd import _lib as B
    lib = B.lib()
    rng = np.random.default_rng(3)
    src = rng.integers(0, 2 ** 31, size=(50, 5), dtype=np.int64).astype(np.uint32)
    ids = np.array([3, 3, 49, 0, 77, 12], dtype=np.uint32)                  # 77: beyond the source -> all-ones
    d_src = torch.from_numpy(src.view(np.int32).copy()).cuda()
    d_ids = torch.from_numpy(ids.view(np.int32).copy()).cuda()
    d_n = torch.tensor([5], dtype=torch.int64, device="cuda")                # only the first five count
    d_dst = torch.full((6 * 5,), 0x55555555, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    assert lib.gf_gather_rows_dev(gf.handle, d_src.data_ptr(), 50, 20, d_ids.data_ptr(), d_n.data_ptr(), 6, d_dst.data_ptr()) == 0
    gf.sync()
    got = d_dst.cpu().numpy().view(np.uint32).reshape(6, 5)
    assert (got[0] == src[3]).all() and (got[1] == src[3]).all() and (got[2] == src[49]).all() and (got[3] == src[0]).all()
    assert (got[4] == 0xFFFFFFFF).all() and (got[5] == 0x55555555).all()


@pytest.mark.parametrize("layout,is_,sd,n_pairs", [("small", 300, 30, 150_001), ("small", 5000, 500, 60_000), ("human", 300, 30, 1_500_000),
                                                     ("human", 5000, 500, 700_000), ("small", 300, 30, 0), ("small", 300, 30, 1)])
def test_tagger_on_the_key_column_equals_the_tagger_on_the_records(gf, layout, is_, sd, n_pairs):
    """gf_tag_alignments_keys_dev streams {pos, ref | mapq0} (8 bytes per record, gf_alnrec_keys_dev) and fetches the 32-byte record only of
    what passes the bin maps: same hits and the same MAPQ-0 list as gf_tag_alignments_low_dev — 4-wave and 16-wave forms, with and without
    the fine map, short- and long-insert branch, odd record counts, unplaced records."""
    import torch
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    lib = B.lib()
    if layout == "small":
        cfg = GapFill.synth_cfg(seed=77, scaffold_len=400_000, n_scaffolds=7, gaps_per_scaffold=6, gap_len=1500, insert_mean=is_, insert_sd=sd, library=1)
    else:      # the human-scale layout: 64-KiB bin map, one 16-wave workgroup per CU, fine map where it is selective
        cfg = GapFill.synth_cfg(seed=20260004, scaffold_len=5_000_000, n_scaffolds=620, gaps_per_scaffold=32, gap_len=2000, insert_mean=is_, insert_sd=sd,
                                library=1 if is_ > 1000 else 0)
    gaps, _ = GapFill.synth_layout(cfg)
    gf.set_gaps(gaps, int(cfg["n_scaffolds"][0]), None)
    n = 2 * n_pairs
    n_use = max(0, n - 1) if n_pairs == 150_001 else n          # an odd count once
    d_reads = torch.empty(max(1, n) * 38 + 64, dtype=torch.uint8, device="cuda")
    d_recs = torch.zeros(max(1, n) * 32, dtype=torch.uint8, device="cuda")
    if n_pairs:
        gf.synth_pairs_dev(cfg, 0, n_pairs, d_reads.data_ptr(), d_recs.data_ptr())
    gf.sync()
    if n_use >= 40:     # a few unplaced records ('*') and a scaffold beyond the .fai
        r = d_recs.view(torch.int32).view(-1, 8)
        r[5, 3] = -1
        r[17, 3] = int(cfg["n_scaffolds"][0]) + 3
    cap = max(1024, n // 4)
    outs = []
    for keyed in (False, True):
        d_hits = torch.zeros(cap * 12, dtype=torch.uint8, device="cuda")
        d_low = torch.zeros(cap * 12, dtype=torch.uint8, device="cuda")
        d_cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
        d_keys = torch.full((n_use + 1,), -1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        if keyed:
            assert lib.gf_alnrec_keys_dev(gf.handle, d_recs.data_ptr(), n_use, d_keys.data_ptr()) == 0
            assert lib.gf_tag_alignments_keys_dev(gf.handle, d_recs.data_ptr(), d_keys.data_ptr(), n_use, is_, sd, 250, 30, d_hits.data_ptr(), cap, d_cnt.data_ptr(),
                                                  d_low.data_ptr(), cap, d_cnt.data_ptr() + 4) == 0
        else:
            assert lib.gf_tag_alignments_low_dev(gf.handle, d_recs.data_ptr(), n_use, is_, sd, 250, 30, d_hits.data_ptr(), cap, d_cnt.data_ptr(),
                                                 d_low.data_ptr(), cap, d_cnt.data_ptr() + 4) == 0
        gf.sync()
        nh, nl = int(d_cnt[0]), int(d_cnt[1])
        assert nh <= cap and nl <= cap
        hits = np.frombuffer(d_hits[:nh * 12].cpu().numpy().tobytes(), dtype=B.TAGHIT)
        low = np.frombuffer(d_low[:nl * 12].cpu().numpy().tobytes(), dtype=np.dtype([("pos", "<u4"), ("ref", "<u4"), ("rec", "<u4")]))
        outs.append((np.sort(hits, order=["rec", "gap", "kind", "to_mate"]), np.sort(low, order=["rec"])))
    assert len(outs[0][0]) == len(outs[1][0]) and (outs[0][0] == outs[1][0]).all()
    assert len(outs[0][1]) == len(outs[1][1]) and (outs[0][1] == outs[1][1]).all()
    if n_pairs >= 60_000:
        assert len(outs[0][0]) > 100 and len(outs[0][1]) > 100


def test_pipeline_grows_hit_buffers_that_start_too_small(gf):
    """Pipeline.prepare(): a library that recruits more hits than its buffers hold (here: a capacity of 64 hits) makes the sizing pass grow them
    and run again — same pools, contigs and picks as with the default capacity; nothing is lost, nothing raises."""
    import torch
    from gappadder_amd.hip_api import GapFill
    from gappadder_amd.pipeline import DeviceLibrary, Pipeline
    seed, slen, nscf, gps, glen, L, n_pairs = 31, 300_000, 2, 5, 150, 150, 40_000
    cfg = GapFill.synth_cfg(seed=seed, scaffold_len=slen, n_scaffolds=nscf, gaps_per_scaffold=gps, gap_len=glen, read_len=L)
    gaps, flanks = GapFill.synth_layout(cfg)
    gf.set_gaps(gaps, nscf, flanks)
    out = []
    for cap in (None, 64):
        d_reads = torch.empty(2 * n_pairs * 38 + 64, dtype=torch.uint8, device="cuda")
        d_recs = torch.empty(2 * n_pairs * 32, dtype=torch.uint8, device="cuda")
        gf.synth_pairs_dev(cfg, 0, n_pairs, d_reads.data_ptr(), d_recs.data_ptr())
        gf.sync()
        pipe = Pipeline(gf, len(gaps), L, [(31, 29)], keep_read_ids=True)
        lb = pipe.add_library(DeviceLibrary("x", 300, 30, 2 * n_pairs, d_reads, d_recs), hit_cap=cap)
        pipe.prepare()
        pipe.finish()
        res = pipe.fetch(pools=True)
        assert lb.counts["tagger_hits"] > 64 and lb.counts["screen_hits"] > 64 and lb.hit_cap >= lb.counts["pool_keys"] // 4
        ctg = sorted((int(c["gap"]), res.seq[int(c["seq_off"]):int(c["seq_off"]) + int(c["length"])]) for c in res.contigs)
        from gappadder_amd.pipeline import decode_best
        picks = []                 # (the pick word names its contig by index in the device list, whose order is unspecified: compare the contigs)
        for g, b in enumerate(res.best.tolist()):
            if b:
                a_len, span1, ci, rev = decode_best(b)
                c = res.contigs[ci]
                picks.append((g, a_len, span1, rev, res.seq[int(c["seq_off"]):int(c["seq_off"]) + int(c["length"])]))
        out.append((lb.counts, res.pool_off.tolist(), res.pool_rows.tobytes(), ctg, picks))
        gf.set_option("asm_max_pool_reads", 0)
    assert out[0] == out[1] and len(out[0][3]) > 5
