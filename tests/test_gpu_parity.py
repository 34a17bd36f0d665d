"""GPU parity tests proper: the HIP path, called through the C ABI, against the oracle and the reference-generated
golden fixtures.  Bit-exact (integer / index work)."""
import os

import numpy as np
import pytest

from golden_util import CASES, Case
from oracle import c_oracle as CO
from oracle import gp_oracle as O
import records_util as RU
import synth_small as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gf():
    from gappadder_amd.hip_api import GapFill
    g = GapFill(0)
    yield g
    g.close()


@pytest.fixture(scope="module", params=CASES)
def case(request):
    return Case(request.param)


def _same(a, b):
    return len(a) == len(b) and a.tobytes() == np.ascontiguousarray(b).astype(a.dtype).tobytes()


# ---------------------------------------------------------------------------------- golden: tagger
def test_tagger_equals_reference_lists(gf, case):
    gaps = O.gap_positions(case.fasta_records(), case.meta["min_gap"])
    garr = RU.gaps_array(case.fai_names, gaps)
    gf.set_gaps(garr, len(case.fai_names))
    for lib in case.libs:
        recs, fields = RU.sam_to_records(lib["sam"], case.fai_names)
        hits = gf.tag_alignments(recs, lib["is"], lib["sd"], case.meta["clip_dist"], case.meta["anchor_mapq"])
        assert _same(hits, CO.tag_alignments(recs, garr, lib["is"], lib["sd"], case.meta["clip_dist"], case.meta["anchor_mapq"]))
        gf.set_option("tag_light", 1)      # the one-wave variant that reads the bin map through L1/L2 (runs beside the k-mer filter)
        try:
            assert _same(gf.tag_alignments(recs, lib["is"], lib["sd"], case.meta["clip_dist"], case.meta["anchor_mapq"]), hits)
        finally:
            gf.set_option("tag_light", 0)
        got = RU.hits_to_lines(hits, recs, fields, garr, case.fai_names)
        exp = case.exp_dir(lib["folder"] + "/scaffold_reads_list_all/")
        for scf in set(g[3] for g in gaps):
            for side in ("left", "right"):
                e = exp["%s_cluster_by_gap_reads_%s.list" % (scf, side)].splitlines()
                assert sorted(got.get(scf, {}).get(side, [])) == sorted(e), (lib["folder"], scf, side)


def test_low_mapq_equals_reference_lists(gf, case):
    gaps = O.gap_positions(case.fasta_records(), case.meta["min_gap"])
    gf.set_gaps(RU.gaps_array(case.fai_names, gaps), len(case.fai_names))
    for lib in case.libs:
        rows = [tuple(int(x) for x in l.split()) for l in case.exp_lines(lib["folder"] + "/discordant_reads_pos.txt.sorted.txt")]
        table = RU.dpos_array(rows)
        recs, fields = RU.sam_to_records(lib["sam"], case.fai_names)
        hits = gf.tag_low_mapq(recs, table)
        assert _same(hits, CO.tag_low_mapq(recs, table))
        got = RU.lowmapq_hits_to_lines(hits, fields, table, case.fai_names)
        exp = case.exp_dir(lib["folder"] + "/discordant_reads_list/")
        for name, txt in exp.items():
            scf, side = name.rsplit("_cluster_by_discordant_reads_", 1)
            assert got.get(scf, {}).get(side.split(".")[0], []) == txt.splitlines(), (lib["folder"], name)


# ---------------------------------------------------------------------------------- golden inputs: screen
@pytest.mark.parametrize("k,min_hits", [(31, 1), (41, 1), (51, 2), (30, 1), (16, 1), (64, 1)])
def test_screen_on_golden_reads(gf, case, k, min_hits):
    from gappadder_amd.hip_api import GapFill
    gaps = O.gap_positions(case.fasta_records(), case.meta["min_gap"])
    seqs = dict(case.fasta_records())
    flanks = [O.flank_seqs(seqs[scf], s, e, case.meta["flank"]) for (s, e, _, scf) in gaps]
    gf.set_gaps(RU.gaps_array(case.fai_names, gaps), len(case.fai_names), flanks)
    lib = case.libs[0]
    reads = RU.fastq_seqs(lib["fq1"]) + RU.fastq_seqs(lib["fq2"])
    L = len(reads[0])
    blob = "".join(reads).encode()
    packed, _ = GapFill.pack_reads(blob, L)
    hits = gf.screen_reads(packed, L, k, min_hits)
    exp = CO.screen_reads(blob, L, flanks, k, min_hits)
    assert _same(hits, exp)
    if k <= 51:
        assert len(exp) > 0


# ---------------------------------------------------------------------------------- seeded synthetic, larger
@pytest.mark.parametrize("seed,n_pairs,L,k", [(1, 30000, 150, 31), (2, 20000, 150, 41), (3, 20000, 100, 31), (4, 8000, 250, 51), (5, 2500, 600, 31), (6, 1500, 1000, 64)])
def test_screen_and_tagger_synthetic(gf, seed, n_pairs, L, k):
    from gappadder_amd.hip_api import GapFill
    c = S.small_case(seed=seed, n_pairs=n_pairs, L=L, insert=max(300, L + 100))
    gf.set_gaps(c["gaps"], c["n_scaffolds"], c["flanks"])
    packed, _ = GapFill.pack_reads(c["reads_blob"], L)
    hits = gf.screen_reads(packed, L, k)
    exp = CO.screen_reads(c["reads_blob"], L, c["flanks"], k)
    assert _same(hits, exp) and len(exp) > 100
    if L <= 250:      # the partitioned filters (packed reads up to 64 bytes) on the same reads, with a 2^27-bit bitmap:
        for variant, ext in ((16, 1), (17, 1), (16, 0)):  # 256 buckets with the workgroup sort and 4-byte pairs (whole-line stores where they fit; 17: unaligned runs); seeds with / without the neighbour check
            gf.set_option("screen_variant", variant)
            gf.set_option("screen_ext", ext)
            gf.set_option("bitmap_log2", 27)
            try:
                assert _same(gf.screen_reads(packed, L, k), exp), (variant, ext)
            finally:
                gf.set_option("screen_variant", 0)
                gf.set_option("screen_ext", 1)
                gf.set_option("bitmap_log2", 0)
    for (IS, sd) in ((max(300, L + 100), 30), (5000, 500)):
        th = gf.tag_alignments(c["recs"], IS, sd)
        assert _same(th, CO.tag_alignments(c["recs"], c["gaps"], IS, sd))
        assert len(th) > 50


def test_reads_with_N_use_the_mask(gf):
    from gappadder_amd.hip_api import GapFill
    c = S.small_case(seed=9, n_pairs=6000, n_frac=0.3)
    gf.set_gaps(c["gaps"], c["n_scaffolds"], c["flanks"])
    packed, nm = GapFill.pack_reads(c["reads_blob"], c["L"], with_mask=True)
    hits = gf.screen_reads(packed, c["L"], 31, 1, n_mask=nm)
    exp = CO.screen_reads(c["reads_blob"], c["L"], c["flanks"], 31)
    assert _same(hits, exp)
    # without the mask N is read as A (KmerUtils.cpp:25): a superset, never a subset
    loose = gf.screen_reads(packed, c["L"], 31, 1)
    assert set(map(tuple, exp.tolist())) <= set(map(tuple, loose.tolist()))


def test_max_gaps_per_kmer_rule(gf):
    from gappadder_amd.hip_api import GapFill
    c = S.small_case(seed=11, n_pairs=3000)
    flanks = list(c["flanks"])
    flanks[1] = flanks[0]          # two gaps share every flank k-mer
    flanks[2] = (flanks[0][0], flanks[2][1])
    gf.set_option("max_gaps_per_kmer", 2)
    try:
        gf.set_gaps(c["gaps"], c["n_scaffolds"], flanks)
        packed, _ = GapFill.pack_reads(c["reads_blob"], c["L"])
        hits = gf.screen_reads(packed, c["L"], 31)
        assert _same(hits, CO.screen_reads(c["reads_blob"], c["L"], flanks, 31, 1, 2))
    finally:
        gf.set_option("max_gaps_per_kmer", 0)
    gf.set_gaps(c["gaps"], c["n_scaffolds"], flanks)
    packed, _ = GapFill.pack_reads(c["reads_blob"], c["L"])
    assert _same(gf.screen_reads(packed, c["L"], 31), CO.screen_reads(c["reads_blob"], c["L"], flanks, 31))


# ---------------------------------------------------------------------------------- edges
def test_empty_and_ragged_inputs(gf):
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    c = S.small_case(seed=5, n_pairs=700)
    gf.set_gaps(c["gaps"], c["n_scaffolds"], c["flanks"])
    L = c["L"]
    assert len(gf.screen_reads(np.zeros((0, 38), np.uint8), L, 31)) == 0
    assert len(gf.tag_alignments(np.zeros(0, B.ALNREC), 300, 30)) == 0
    assert len(gf.tag_low_mapq(c["recs"], np.zeros(0, B.DPOS))) == 0
    for n in (1, 63, 64, 255, 256, 257, 1399):     # tails of the 256-read tile and of the 64-lane wave
        blob = c["reads_blob"][:n * L]
        packed, _ = GapFill.pack_reads(blob, L)
        assert _same(gf.screen_reads(packed, L, 31), CO.screen_reads(blob, L, c["flanks"], 31)), n
        assert _same(gf.tag_alignments(c["recs"][:n], 300, 30), CO.tag_alignments(c["recs"][:n], c["gaps"], 300, 30)), n
    # no gaps at all
    gf.set_gaps(np.zeros(0, B.GAP), c["n_scaffolds"], [])
    packed, _ = GapFill.pack_reads(c["reads_blob"], L)
    assert len(gf.screen_reads(packed, L, 31)) == 0
    assert len(gf.tag_alignments(c["recs"], 300, 30)) == 0


def test_nospace_reports_required_count(gf):
    import ctypes as C
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    c = S.small_case(seed=6, n_pairs=3000)
    gf.set_gaps(c["gaps"], c["n_scaffolds"], c["flanks"])
    packed, _ = GapFill.pack_reads(c["reads_blob"], c["L"])
    full = gf.screen_reads(packed, c["L"], 31)
    out = np.zeros(3, dtype=B.HIT)
    n = C.c_size_t(0)
    rc = B.lib().gf_screen_reads(gf.handle, B._p(packed), None, len(packed), c["L"], 31, 1, B._p(out), 3, C.byref(n))
    assert rc == B.GF_E_NOSPACE and n.value == len(full)
    rc = B.lib().gf_screen_reads(gf.handle, B._p(packed), None, len(packed), c["L"], 8, 1, B._p(out), 3, C.byref(n))
    assert rc == B.GF_E_UNSUPPORTED


def test_screen_properties_at_scale(gf):
    """Size-independent properties on 2 M reads (the oracle would take minutes): every read cut from a flank is
    recruited to its gap; reads from an unrelated random genome are never recruited; results are idempotent and
    independent of the read order."""
    from gappadder_amd.hip_api import GapFill
    c = S.small_case(seed=21, n_pairs=10)
    gf.set_gaps(c["gaps"], c["n_scaffolds"], c["flanks"])
    L, k = 150, 31
    rng = np.random.RandomState(3)
    n = 2_000_000
    arr = rng.randint(0, 4, (n, L)).astype(np.uint8)
    lut = np.frombuffer(b"ACGT", np.uint8)
    planted = {}
    for g, (l, r) in enumerate(c["flanks"]):
        for f in (l, r):
            for off in (0, 17, len(f) - 60):
                i = rng.randint(n)
                seg = np.frombuffer(f[off:off + 60].encode(), np.uint8)
                arr[i, 40:100] = np.searchsorted(lut, seg)
                planted[i] = g
    blob = lut[arr].tobytes()
    packed, _ = GapFill.pack_reads(blob, L)
    hits = gf.screen_reads(packed, L, k)
    got = {(int(h["gap"]), int(h["read"])) for h in hits}
    assert got == {(g, i) for i, g in planted.items()}
    perm = rng.permutation(n)
    hits2 = gf.screen_reads(packed[perm], L, k)
    assert {(int(h["gap"]), int(perm[h["read"]])) for h in hits2} == got
    assert _same(gf.screen_reads(packed, L, k), hits)


def test_synthetic_generator_matches_oracle_bit_for_bit(gf):
    """The HIP generator of the bench workload and the oracle's C generator share include/gf_synth.h."""
    import torch
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    cfg = GapFill.synth_cfg(scaffold_len=300000, n_scaffolds=5, gaps_per_scaffold=4, chimeric=0.05, mapq0=0.1)
    n_pairs, first = 70001, 12345
    d_reads = torch.zeros(2 * n_pairs * 38, dtype=torch.uint8, device="cuda:0")
    d_recs = torch.zeros(2 * n_pairs * 32, dtype=torch.uint8, device="cuda:0")
    gf.synth_pairs_dev(cfg, first, n_pairs, d_reads.data_ptr(), d_recs.data_ptr())
    gf.sync()
    ocfg = np.frombuffer(cfg.tobytes(), dtype=CO.SYNTH_CFG).copy()
    packed, recs = CO.synth_pairs(ocfg, first, n_pairs)
    assert d_reads.cpu().numpy().tobytes() == packed.tobytes()
    assert d_recs.cpu().numpy().tobytes() == recs.tobytes()
    g1, f1 = GapFill.synth_layout(cfg)
    g2, f2 = CO.synth_layout(ocfg)
    assert g1.tobytes() == g2.tobytes() and f1 == f2
    # and the hot path on it
    gf.set_gaps(g1, 5, f1)
    blob = CO.unpack_reads(packed, 150)
    assert _same(gf.screen_reads(packed, 150, 31), CO.screen_reads(blob, 150, f2, 31))
    assert _same(gf.tag_alignments(recs, 300, 30), CO.tag_alignments(recs, g2, 300, 30))


@pytest.mark.parametrize("n_big", [16385, 40000, 150001])
def test_device_pools_keep_gaps_with_more_keys_than_the_lds_sort_holds(gf, n_big):
    """A gap whose flank sits in a repeat recruits far more reads than the 16 384 keys one LDS sort holds (the reference has no
    bound: run_multi_threads_discordant.py:209-241 flushes any number of records).  Such a gap is sorted in place in global memory
    — same pools as numpy's sort + unique, no error flag, the neighbours untouched."""
    import torch
    from gappadder_amd import _lib as B
    c = S.small_case(seed=33, n_pairs=600)
    gf.set_gaps(c["gaps"], c["n_scaffolds"], c["flanks"])
    n_gaps, L, rb = len(c["gaps"]), 150, 38
    assert n_gaps >= 3
    rng = np.random.RandomState(n_big)
    n_reads = 400_000
    per_gap = {0: rng.randint(0, n_reads, 900), 1: rng.randint(0, n_reads, n_big), 2: rng.randint(0, n_reads, 17000)}
    per_gap[1][::7] = per_gap[1][0]                      # many duplicates of one key
    if n_big > 20000: per_gap[1][5000:12000] = np.arange(7000) * 2 + 1     # a run of right mates
    keys = np.concatenate([(np.uint64(g) << np.uint64(32)) | v.astype(np.uint64) for g, v in per_gap.items()])
    rng.shuffle(keys)
    dev = torch.device("cuda:0")
    d_reads = torch.randint(0, 256, (n_reads, rb), dtype=torch.uint8, device=dev)
    d_keys = torch.from_numpy(keys.view(np.int64)).to(dev)
    d_nk = torch.tensor([len(keys), 0, 0, 0], dtype=torch.int32, device=dev)
    pool_cap = len(keys) + 8
    d_pool = torch.zeros(pool_cap * rb, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n_gaps + 1, dtype=torch.int64, device=dev)
    d_ids = torch.zeros(pool_cap, dtype=torch.int32, device=dev)
    d_err = torch.zeros(4, dtype=torch.int32, device=dev)
    assert B.lib().gf_build_pools_dev(gf.handle, d_reads.data_ptr(), n_reads, L, d_keys.data_ptr(), d_nk.data_ptr(), len(keys), d_pool.data_ptr(),
                                      pool_cap, d_off.data_ptr(), d_ids.data_ptr(), d_err.data_ptr()) == 0
    gf.sync()
    assert int(d_err[0]) == 0
    off = d_off.cpu().numpy()
    ids = d_ids.cpu().numpy().astype(np.uint32)
    reads = d_reads.cpu().numpy()
    pool = d_pool.cpu().numpy().reshape(pool_cap, rb)
    for g in range(n_gaps):
        u = np.unique(per_gap[g]) if g in per_gap else np.zeros(0, np.int64)
        want = u[np.lexsort((u >> 1, u & 1))].astype(np.uint32)      # (mate, pair): left-file stream order, then right-file order
        got = ids[off[g]:off[g + 1]]
        assert len(got) == len(want) and (got == want).all(), g
        assert (pool[off[g]:off[g + 1]] == reads[want]).all(), g


@pytest.mark.parametrize("L_case", [150, 100, 101, 250])     # packed rows of 38 (the dword-and-a-half fast path), 25, 26 and 63 bytes
def test_device_pools_follow_the_reference_fastq_join_order(gf, L_case):
    """gf_build_pools_dev: keys from screen (+mates), tagger and second hop -> per-gap pools ordered (mate, pair)."""
    import torch
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    c = S.small_case(seed=31, n_pairs=9000, L=L_case, insert=max(300, L_case + 100))
    L, n = c["L"], c["n_reads"]
    gf.set_gaps(c["gaps"], c["n_scaffolds"], c["flanks"])
    packed, _ = GapFill.pack_reads(c["reads_blob"], L)
    recs = c["recs"]
    shits = gf.screen_reads(packed, L, 31)
    thits = gf.tag_alignments(recs, 300, 30)
    # second hop table from the discordant tagger hits (run_multi_threads_discordant.py:47-103)
    rows = sorted((int(recs[h["rec"]]["mate_ref"]), int(recs[h["rec"]]["mate_pos"]), int(c["gaps"][h["gap"]]["scaffold"]),
                   int(c["gaps"][h["gap"]]["idx_in_scaffold"])) for h in thits if h["kind"] == B.KIND_DISCORDANT)
    table = RU.dpos_array(rows).astype(B.DPOS)
    lhits = gf.tag_low_mapq(recs, table)
    assert len(lhits) > 0
    # expected: the set of (gap, read) keys, per gap sorted by (mate, pair)
    keys = set()
    for h in shits:
        keys.add((int(h["gap"]), int(h["read"])))
        keys.add((int(h["gap"]), int(h["read"]) ^ 1))
    for h in thits:
        keys.add((int(h["gap"]), int(recs[h["rec"]]["read"]) ^ int(h["to_mate"])))
    gidx = {(int(g["scaffold"]), int(g["idx_in_scaffold"])): i for i, g in enumerate(c["gaps"])}
    for h in lhits:
        keys.add((gidx[(int(table[h["gap"]]["src_scaffold"]), int(table[h["gap"]]["src_gap"]))], int(recs[h["rec"]]["read"])))
    exp = {}
    for g, r in keys:
        exp.setdefault(g, []).append(r)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.frombuffer(a.tobytes(), dtype=np.uint8).copy()).to(dev)
    d_reads, d_recs = t(packed), t(recs)
    d_sh, d_th, d_lh = t(shits), t(thits), t(lhits)
    cnts = torch.tensor([len(shits), len(thits), len(lhits), 0], dtype=torch.int32, device=dev)
    key_cap = 4 * (2 * len(shits) + len(thits) + len(lhits)) + 16
    d_keys = torch.zeros(key_cap, dtype=torch.int64, device=dev)
    d_nk = torch.zeros(4, dtype=torch.int32, device=dev)
    pool_cap = len(keys) + 8
    rb = packed.shape[1]
    d_pool = torch.zeros(pool_cap * rb, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(len(c["gaps"]) + 1, dtype=torch.int64, device=dev)
    d_ids = torch.zeros(pool_cap, dtype=torch.int32, device=dev)
    d_err = torch.zeros(4, dtype=torch.int32, device=dev)
    Lb, h = B.lib(), gf.handle
    assert Lb.gf_pool_keys_reset(h, d_nk.data_ptr()) == 0
    assert Lb.gf_pool_keys_from_screen_dev(h, d_sh.data_ptr(), cnts.data_ptr(), len(shits), 1, d_keys.data_ptr(), key_cap, d_nk.data_ptr()) == 0
    assert Lb.gf_pool_keys_from_tags_dev(h, d_recs.data_ptr(), d_th.data_ptr(), cnts.data_ptr() + 4, len(thits), None, 0,
                                         d_keys.data_ptr(), key_cap, d_nk.data_ptr()) == 0
    assert Lb.gf_pool_keys_from_tags_dev(h, d_recs.data_ptr(), d_lh.data_ptr(), cnts.data_ptr() + 8, len(lhits), B._p(table), len(table),
                                         d_keys.data_ptr(), key_cap, d_nk.data_ptr()) == 0
    assert Lb.gf_build_pools_dev(h, d_reads.data_ptr(), n, L, d_keys.data_ptr(), d_nk.data_ptr(), key_cap, d_pool.data_ptr(), pool_cap,
                                 d_off.data_ptr(), d_ids.data_ptr(), d_err.data_ptr()) == 0
    gf.sync()
    torch.cuda.synchronize()
    assert int(d_err[0]) == 0 and int(d_nk[0]) == 2 * len(shits) + len(thits) + len(lhits)
    off = d_off.cpu().numpy()
    ids = d_ids.cpu().numpy().astype(np.uint32)
    pool = d_pool.cpu().numpy().reshape(pool_cap, rb)
    assert off[-1] == len(keys)
    for g in range(len(c["gaps"])):
        want = sorted(exp.get(g, []), key=lambda r: (r & 1, r >> 1))
        got = ids[off[g]:off[g + 1]].tolist()
        assert got == want, g
        assert (pool[off[g]:off[g + 1]] == packed[want]).all()
    # and the pools assemble like the oracle says
    ctg, seq = gf.assemble(pool[:off[-1]], off.astype(np.uint64), L, [(31, 29)])
    blob = CO.unpack_reads(pool[:off[-1]], L)
    for g in range(len(c["gaps"])):
        e = CO.assemble_pool(blob[int(off[g]) * L:int(off[g + 1]) * L], L, 31, 29)
        gseqs = [seq[int(x["seq_off"]):int(x["seq_off"]) + int(x["length"])].decode() for x in ctg if x["gap"] == g]
        assert gseqs == [x[0] for x in e]


@pytest.mark.parametrize("variant", [13, 9, 16])
def test_screen_filter_variants_agree(gf, variant):
    """Every filter kernel (pipelined = 13, plain L2 bitmap = 9, partitioned = 16 / 17) gives the oracle's hits."""
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    c = S.small_case(seed=41, n_pairs=25000)
    gf.set_gaps(c["gaps"], c["n_scaffolds"], c["flanks"])
    packed, _ = GapFill.pack_reads(c["reads_blob"], c["L"])
    exp31 = CO.screen_reads(c["reads_blob"], c["L"], c["flanks"], 31)
    exp51 = CO.screen_reads(c["reads_blob"], c["L"], c["flanks"], 51)
    gf.set_option("screen_variant", variant)
    try:
        for n in (len(packed), 1000, 769, 1):
            assert _same(gf.screen_reads(packed[:n], c["L"], 31), exp31[exp31["read"] < n]), (variant, n)
        assert _same(gf.screen_reads(packed, c["L"], 51), exp51)
        for bl in (16, 19, 22, 26, 28, 29):  # 26: level-1 bitmap beyond the L2 -> the plain kernel also asks the 2^24-bit reduction;
            gf.set_option("bitmap_log2", bl)      # (the partitioned filter exists for 2^27 and 2^28 bits: the plain kernel otherwise)
            assert _same(gf.screen_reads(packed, c["L"], 31), exp31), (variant, bl)
        if variant == 16:      # the 256-bucket partitioned filter exists for 2^27- and 2^28-bit bitmaps
            exp41 = CO.screen_reads(c["reads_blob"], c["L"], c["flanks"], 41)
            for v256, ext in ((16, 1), (17, 1), (16, 0)):   # pass A with whole-line stores / with unaligned runs; seeds with / without the neighbour check
                gf.set_option("screen_variant", v256)
                gf.set_option("screen_ext", ext)
                for bl in (27, 28):
                    gf.set_option("bitmap_log2", bl)
                    for n in (len(packed), 1000, 769, 1):
                        assert _same(gf.screen_reads(packed[:n], c["L"], 31), exp31[exp31["read"] < n]), (v256, ext, bl, n)
                    assert _same(gf.screen_reads(packed, c["L"], 51), exp51), (v256, ext, bl)
                    assert _same(gf.screen_reads(packed, c["L"], 41), exp41), (v256, ext, bl)
            # the whole-line pass A is instantiated per probes per read (1 ... 4): k = 51 above has three; k = 61 two, k = 45 four, and
            # 100-base reads at k = 61 one
            gf.set_option("screen_variant", 16)
            gf.set_option("screen_ext", 1)
            gf.set_option("bitmap_log2", 28)
            for k in (61, 45):
                assert _same(gf.screen_reads(packed, c["L"], k), CO.screen_reads(c["reads_blob"], c["L"], c["flanks"], k)), k
                assert "pf4_scatter_lines_kernel<%du" % {61: 2, 45: 4}[k] in B.lib().gf_screen_kernels(gf.handle).decode()
            # five to eight probes per read (k = 41 ... 31 on 150-base reads): TWO groups of four per tile iteration through the same whole-line
            # kernel, the slots beyond the read's probe count dead
            for k in (31, 33, 36, 41):
                assert _same(gf.screen_reads(packed, c["L"], k), CO.screen_reads(c["reads_blob"], c["L"], c["flanks"], k)), k
                names = B.lib().gf_screen_kernels(gf.handle).decode()
                assert "pf4_scatter_lines_kernel<4u, " in names and ", 2u>" in names, (k, names)
                for n in (1000, 769, 1):
                    e_k = CO.screen_reads(c["reads_blob"][:n * c["L"]], c["L"], c["flanks"], k)
                    assert _same(gf.screen_reads(packed[:n], c["L"], k), e_k), (k, n)
            c2 = S.small_case(seed=42, n_pairs=12000, L=100)
            gf.set_gaps(c2["gaps"], c2["n_scaffolds"], c2["flanks"])
            packed2, _ = GapFill.pack_reads(c2["reads_blob"], 100)
            assert _same(gf.screen_reads(packed2, 100, 61), CO.screen_reads(c2["reads_blob"], 100, c2["flanks"], 61))
            assert "pf4_scatter_lines_kernel<1u" in B.lib().gf_screen_kernels(gf.handle).decode()
    finally:
        gf.set_option("screen_variant", 0)
        gf.set_option("screen_ext", 1)
        gf.set_option("bitmap_log2", 0)


def test_one_pass_tagger_with_mapq0_compaction_equals_two_passes(gf):
    """gf_tag_alignments_low_dev + gf_tag_low_mapq_compact_dev give the hits of gf_tag_alignments + gf_tag_low_mapq."""
    import torch
    from gappadder_amd import _lib as B
    c = S.small_case(seed=51, n_pairs=40000)
    gf.set_gaps(c["gaps"], c["n_scaffolds"], c["flanks"])
    recs = c["recs"]
    thits = gf.tag_alignments(recs, 300, 30)
    rows = sorted((int(recs[h["rec"]]["mate_ref"]), int(recs[h["rec"]]["mate_pos"]), int(c["gaps"][h["gap"]]["scaffold"]),
                   int(c["gaps"][h["gap"]]["idx_in_scaffold"])) for h in thits if h["kind"] == B.KIND_DISCORDANT)
    table = RU.dpos_array(rows).astype(B.DPOS)
    lhits = gf.tag_low_mapq(recs, table)
    dev = torch.device("cuda:0")
    d_recs = torch.from_numpy(np.frombuffer(recs.tobytes(), dtype=np.uint8).copy()).to(dev)
    n = len(recs)
    cap = 4 * (len(thits) + len(lhits)) + 64
    d_t = torch.zeros(cap * 12, dtype=torch.uint8, device=dev)
    d_l = torch.zeros(cap * 12, dtype=torch.uint8, device=dev)
    low_cap = n
    d_low = torch.zeros(low_cap * 12, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(8, dtype=torch.int32, device=dev)
    L, h, cp = B.lib(), gf.handle, d_cnt.data_ptr()
    assert L.gf_tag_alignments_low_dev(h, d_recs.data_ptr(), n, 300, 30, 250, 30, d_t.data_ptr(), cap, cp, d_low.data_ptr(), low_cap, cp + 4) == 0
    assert L.gf_tag_low_mapq_compact_dev(h, d_low.data_ptr(), cp + 4, low_cap, B._p(table), len(table), d_l.data_ptr(), cap, cp + 8) == 0
    gf.sync()
    torch.cuda.synchronize()
    cnt = d_cnt.cpu().numpy()
    valid = recs["ref"] < c["n_scaffolds"]
    assert int(cnt[1]) == int(((recs["mapq"] == 0) & valid).sum())
    got_t = np.sort(np.frombuffer(d_t[:int(cnt[0]) * 12].cpu().numpy().tobytes(), dtype=B.TAGHIT), order=["rec", "gap", "kind"])
    got_l = np.sort(np.frombuffer(d_l[:int(cnt[2]) * 12].cpu().numpy().tobytes(), dtype=B.TAGHIT), order=["rec", "gap", "kind"])
    assert _same(got_t, thits) and _same(got_l, lhits) and len(lhits) > 0


def test_mapq0_list_survives_many_flushes_of_the_staging_slices(gf):
    """The MAPQ-0 by-product of the tagger leaves through per-wave staging slices of 1 024 records (tagger.hip): with 12 M records,
    70 % of them MAPQ 0, every wave fills and empties its slice several times.  The list must hold exactly the MAPQ-0 records with a
    valid scaffold — each once, with its position and record index — and the tagger's hits must not change."""
    import torch
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    cfg = GapFill.synth_cfg(seed=9, scaffold_len=5_000_000, n_scaffolds=620, gaps_per_scaffold=32, gap_len=2000)   # (the 64-KiB bin map: 16-wave workgroups)
    gaps, flanks = GapFill.synth_layout(cfg)
    gf.set_gaps(gaps, 620, flanks)
    n_pairs = 6_000_000
    dev = torch.device("cuda:0")
    rb = B.lib().gf_packed_read_bytes(150)
    d_reads = torch.empty(2 * n_pairs * rb + 64, dtype=torch.uint8, device=dev)
    d_recs = torch.empty(2 * n_pairs * 32, dtype=torch.uint8, device=dev)
    gf.synth_pairs_dev(cfg, 0, n_pairs, d_reads.data_ptr(), d_recs.data_ptr())
    gf.sync()
    del d_reads
    recs = np.frombuffer(d_recs.cpu().numpy().tobytes(), dtype=B.ALNREC).copy()
    rng = np.random.RandomState(4)
    recs["mapq"][rng.rand(len(recs)) < 0.7] = 0
    recs["ref"][rng.rand(len(recs)) < 0.01] = 0xFFFFFFFF          # unplaced records: not listed
    d_recs.copy_(torch.from_numpy(np.frombuffer(recs.tobytes(), dtype=np.uint8).copy()))
    n = len(recs)
    cap = 1 << 22
    d_t = torch.zeros(cap * 12, dtype=torch.uint8, device=dev)
    d_low = torch.zeros(n * 12, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(8, dtype=torch.int32, device=dev)
    L, h, cp = B.lib(), gf.handle, d_cnt.data_ptr()
    assert L.gf_tag_alignments_low_dev(h, d_recs.data_ptr(), n, 300, 30, 250, 30, d_t.data_ptr(), cap, cp, d_low.data_ptr(), n, cp + 4) == 0
    gf.sync()
    cnt = d_cnt.cpu().numpy()
    want = np.nonzero((recs["mapq"] == 0) & (recs["ref"] < 620))[0]
    assert int(cnt[1]) == len(want) > 8_000_000
    low = np.frombuffer(d_low[:len(want) * 12].cpu().numpy().tobytes(), dtype=np.dtype([("pos", "<u4"), ("ref", "<u4"), ("rec", "<u4")]))
    order = np.argsort(low["rec"], kind="stable")
    assert (low["rec"][order] == want).all()
    assert (low["pos"][order] == recs["pos"][want]).all() and (low["ref"][order] == recs["ref"][want]).all()
    assert L.gf_tag_alignments_dev(h, d_recs.data_ptr(), n, 300, 30, 250, 30, d_low.data_ptr(), cap, cp + 8) == 0      # (d_low re-used as a hit buffer)
    gf.sync()
    cnt = d_cnt.cpu().numpy()
    assert int(cnt[0]) == int(cnt[2]) > 1000
    a = np.sort(np.frombuffer(d_t[:int(cnt[0]) * 12].cpu().numpy().tobytes(), dtype=B.TAGHIT), order=["rec", "gap", "kind"])
    b = np.sort(np.frombuffer(d_low[:int(cnt[2]) * 12].cpu().numpy().tobytes(), dtype=B.TAGHIT), order=["rec", "gap", "kind"])
    assert a.tobytes() == b.tobytes()


def test_every_read_a_hit_fills_the_verification_staging_slices(gf):
    """10 M reads cut out of the flanks themselves: every read hits (exactly) its gap, so every wave of the seed-and-extend verification
    collects more hits than its staging slice holds (1 024) and appends to the hit list several times."""
    from gappadder_amd.hip_api import GapFill
    cfg = GapFill.synth_cfg(seed=20260002, scaffold_len=5_000_000, n_scaffolds=10, gaps_per_scaffold=100, gap_len=1000)
    gaps, flanks = GapFill.synth_layout(cfg)
    gf.set_gaps(gaps, 10, flanks)
    L, n = 150, 10_000_000
    rng = np.random.RandomState(8)
    text = np.frombuffer("".join(f[0] + f[1] for f in flanks).encode(), dtype=np.uint8)
    flen = np.array([[len(f[0]), len(f[1])] for f in flanks], dtype=np.int64).reshape(-1)
    fstart = np.concatenate([[0], np.cumsum(flen)[:-1]])
    side = rng.randint(0, 2 * len(flanks), n)
    assert flen.min() >= L
    off = (rng.rand(n) * (flen[side] - L + 1)).astype(np.int64)
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    hits = []
    exp_gap = (side // 2).astype(np.uint32)
    chunk = 1_000_000
    packed = []
    for a in range(0, n, chunk):
        idx = (fstart[side[a:a + chunk]] + off[a:a + chunk])[:, None] + np.arange(L)[None, :]
        r = text[idx]
        flip = rng.randint(0, 2, len(r)).astype(bool)
        r[flip] = comp[r[flip][:, ::-1]]
        assert not (r == ord("N")).any()
        packed.append(GapFill.pack_reads(r.tobytes(), L)[0])
    packed = np.concatenate(packed)
    got = gf.screen_reads(packed, L, 31, cap=n + (1 << 20))
    assert len(got) == n, (len(got), n)      # (random flanks share no 31-mer: one gap per read)
    assert (got["read"][np.argsort(got["read"], kind="stable")] == np.arange(n, dtype=np.uint32)).all()
    order = np.argsort(got["read"], kind="stable")
    assert (got["gap"][order] == exp_gap).all()


def test_c3_full_size_recruit_matches_oracle(gf):
    """BASELINE.json configs[2] at full size (E. coli-scale: 1 scaffold of 4.6 Mb, 200 gaps x 1 kb, 5 M 150-bp reads, k=41):
    every screen hit and every tagger hit of the GPU equals the oracle's."""
    import torch
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    cfg = GapFill.synth_cfg(seed=20260003, scaffold_len=4_600_000, n_scaffolds=1, gaps_per_scaffold=200, gap_len=1000)
    gaps, flanks = GapFill.synth_layout(cfg)
    gf.set_gaps(gaps, 1, flanks)
    n_pairs = 2_500_000
    ocfg = np.frombuffer(cfg.tobytes(), dtype=CO.SYNTH_CFG).copy()
    packed, recs = CO.synth_pairs(ocfg, 0, n_pairs)
    hits = gf.screen_reads(packed, 150, 41, cap=1 << 21)
    exp = CO.screen_reads(CO.unpack_reads(packed, 150), 150, flanks, 41)
    assert _same(hits, exp) and len(exp) > 50_000
    th = gf.tag_alignments(recs, 300, 30, cap=1 << 21)
    assert _same(th, CO.tag_alignments(recs, gaps, 300, 30)) and len(th) > 20_000


def test_reads_matching_many_gaps_take_the_large_list_pass(gf):
    """Six gaps share one flank: a read inside it matches > 256 (position, gap) pairs, overflows the per-wave list of the
    first verify pass and is re-verified by the large-list pass; min_hits counts positions per gap."""
    from gappadder_amd.hip_api import GapFill
    c = S.small_case(seed=61, n_pairs=4000)
    flanks = [(c["flanks"][0][0], f[1]) for f in c["flanks"][:6]] + list(c["flanks"][6:])
    gf.set_gaps(c["gaps"], c["n_scaffolds"], flanks)
    packed, _ = GapFill.pack_reads(c["reads_blob"], c["L"])
    for mh in (1, 50, 121):
        hits = gf.screen_reads(packed, c["L"], 31, mh)
        exp = CO.screen_reads(c["reads_blob"], c["L"], flanks, 31, mh)
        assert _same(hits, exp), mh
    full = CO.screen_reads(c["reads_blob"], c["L"], flanks, 31, 1)
    per_read = np.bincount(full["read"])
    assert per_read.max() >= 6          # some read is recruited by all six gaps (>= 6 x 100 matches in the verify list)


@pytest.mark.parametrize("k,L", [(16, 60), (17, 80), (31, 100), (33, 100), (48, 150), (64, 150)])
def test_seed_and_extend_verification_edge_cases(gf, k, L):
    """The seed-and-extend verify kernel against the oracle on reads built to sit on its edges: matches of exactly k-1, k and
    k+1 bases on either strand, at the ends of the read and of the flank, next to a read N and next to a non-ACGT flank base,
    through palindromic 16-mers, and in flanks that repeat a segment (several occurrences per seed, also in other gaps);
    the same inputs through the k-mer table kernel must agree too."""
    from gappadder_amd.hip_api import GapFill
    from gappadder_amd import _lib as B
    rng = np.random.RandomState(1000 + k)
    lut = np.frombuffer(b"ACGT", np.uint8)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    rnd = lambda n: lut[rng.randint(0, 4, n)].tobytes().decode()
    pal = lambda: (lambda h: h + h.encode().translate(comp)[::-1].decode())(rnd(8))          # 16-mer == its reverse complement
    n_gaps = 6
    flanks = []
    for g in range(n_gaps):
        left = rnd(120) + pal() + rnd(60) + "N" + rnd(90) + "acgt" + rnd(70)       # palindrome, N, lower case (= not ACGT)
        rep = rnd(70)
        right = rnd(40) + rep + rnd(30) + rep + pal() + rnd(50)                     # a repeated 70-mer
        flanks.append((left, right))
    flanks[3] = (flanks[0][0][:200] + rnd(80), flanks[3][1])                        # gap 3 shares 200 bases with gap 0
    gaps = np.zeros(n_gaps, dtype=B.GAP)
    gaps["scaffold"] = 0
    gaps["start"] = (np.arange(n_gaps) + 1) * 10000
    gaps["end"] = gaps["start"] + 500
    gaps["idx_in_scaffold"] = np.arange(n_gaps) + 1
    reads = []
    for i in range(6000):
        g = rng.randint(n_gaps)
        src = flanks[g][rng.randint(2)].upper().replace("N", "A")
        m = int(rng.choice([k - 1, k, k + 1, k + 5, min(L, k + 40)]))                # length of the copied stretch
        m = min(m, L, len(src))
        a = rng.randint(0, len(src) - m + 1)
        if rng.rand() < 0.2:
            a = 0 if rng.rand() < 0.5 else len(src) - m                              # at a flank end
        piece = src[a:a + m]
        if rng.rand() < 0.5:
            piece = piece.encode().translate(comp)[::-1].decode()
        off = rng.randint(0, L - m + 1)
        if rng.rand() < 0.3:
            off = 0 if rng.rand() < 0.5 else L - m                                   # at a read end
        r = list(rnd(L))
        r[off:off + m] = piece
        if rng.rand() < 0.3:
            r[rng.randint(L)] = "N"                                                  # an N somewhere, often inside the stretch
        reads.append("".join(r))
    blob = "".join(reads).encode()
    packed, nm = GapFill.pack_reads(blob, L, with_mask=True)
    gf.set_gaps(gaps, 1, flanks)
    exp = CO.screen_reads(blob, L, flanks, k)
    got = gf.screen_reads(packed, L, k, 1, n_mask=nm)
    assert _same(got, exp) and len(exp) > 500
    gf.set_option("screen_verify_ext", 0)
    try:
        assert _same(gf.screen_reads(packed, L, k, 1, n_mask=nm), exp)
    finally:
        gf.set_option("screen_verify_ext", 1)


def test_tagger_on_human_scale_layout_uses_the_fine_bin_map(gf):
    """24 scaffolds x 130 Mb with 20 000 gaps (BASELINE.json configs[3]'s layout): the LDS bin map has ~24 kb bins, so the
    tagger's second, finer map in global memory decides which records reach the window search; hits must stay the oracle's."""
    from gappadder_amd.hip_api import GapFill
    cfg = GapFill.synth_cfg(seed=20260004, scaffold_len=130_000_000, n_scaffolds=24, gaps_per_scaffold=834, gap_len=1000)
    gaps, flanks = GapFill.synth_layout(cfg)
    gf.set_gaps(gaps, 24, flanks)
    ocfg = np.frombuffer(cfg.tobytes(), dtype=CO.SYNTH_CFG).copy()
    _, recs = CO.synth_pairs(ocfg, 0, 1_000_000)
    for d2, cd in ((300, 30), (2500, 100)):
        th = gf.tag_alignments(recs, d2, cd, cap=1 << 21)
        assert _same(th, CO.tag_alignments(recs, gaps, d2, cd)) and len(th) > 1000


def test_tagger_keeps_a_bin_map_per_library_and_starts_the_window_search_from_the_bin(gf):
    """One context tags several libraries in turn (a pipeline does: short inserts, then mate pairs, per batch): the bin map of every
    dist2 stays cached (six insert sizes: the oldest maps are dropped and rebuilt), and the window search starts at the gap the
    map's third part names for the record's bin — also where a bin holds several gaps (1-kb gaps every 4 kb, bins of 64 b to 8 kb)
    and where windows of 13 kb overlap a dozen gaps either side."""
    from gappadder_amd.hip_api import GapFill
    for scaffold_len, n_scaf, per, gap_len in ((400_000, 6, 90, 1000), (5_000_000, 40, 32, 2000)):
        cfg = GapFill.synth_cfg(seed=77, scaffold_len=scaffold_len, n_scaffolds=n_scaf, gaps_per_scaffold=per, gap_len=gap_len)
        gaps, flanks = GapFill.synth_layout(cfg)
        gf.set_gaps(gaps, n_scaf, flanks)
        ocfg = np.frombuffer(cfg.tobytes(), dtype=CO.SYNTH_CFG).copy()
        _, recs = CO.synth_pairs(ocfg, 0, 300_000)
        want = {}
        for rnd in range(2):
            for ins, sd in ((300, 30), (5000, 500), (800, 60), (300, 30), (10000, 1000), (2000, 100), (150, 10), (5000, 500)):
                if (ins, sd) not in want:
                    want[(ins, sd)] = CO.tag_alignments(recs, gaps, ins, sd)
                th = gf.tag_alignments(recs, ins, sd, cap=1 << 22)
                assert _same(th, want[(ins, sd)]), (scaffold_len, rnd, ins, sd)
        assert len(want[(10000, 1000)]) > len(want[(300, 30)]) > 400


def test_partitioned_filter_probes_in_place_when_a_bucket_part_runs_full(gf):
    """Degenerate input for the partitioned filter: 60 000 poly-A reads put every probe into ONE bucket, eight
    times what a writer's part of it holds, so most pairs take the in-place path; the hits must still be the oracle's (the flank
    of gap 0 ends in a poly-A run, so those reads are real hits)."""
    from gappadder_amd.hip_api import GapFill
    c = S.small_case(seed=43, n_pairs=3000)
    flanks = [list(f) for f in c["flanks"]]
    flanks[0][0] = flanks[0][0][:-60] + "A" * 60
    flanks = [tuple(f) for f in flanks]
    L = c["L"]
    blob = c["reads_blob"][:2000 * L] + b"A" * (60000 * L) + c["reads_blob"][2000 * L:]
    gf.set_gaps(c["gaps"], c["n_scaffolds"], flanks)
    packed, _ = GapFill.pack_reads(blob, L)
    exp = CO.screen_reads(blob, L, flanks, 31)
    assert len(exp) > 60000
    try:
        gf.set_option("bitmap_log2", 27)
        for v256 in (16, 17):                      # 256 buckets, 4-byte pairs (whole-line stores / unaligned runs): the same degenerate reads overflow a workgroup's part — a full part is tested on the spot and resolved by the lane itself
            gf.set_option("screen_pf4_cap8", 0)
            gf.set_option("screen_variant", v256)
            assert _same(gf.screen_reads(packed, L, 31, cap=1 << 18), exp), v256
            gf.set_option("screen_pf4_cap8", 256)      # ... and so is every pair beyond a (here: tiny) pair list
            assert _same(gf.screen_reads(packed, L, 31, cap=1 << 18), exp), v256
            for cut in (1, 5, 67):                     # read counts that end inside an octet / a tile
                blob2 = blob[:len(blob) - cut * L]
                packed2, _ = GapFill.pack_reads(blob2, L)
                assert _same(gf.screen_reads(packed2, L, 31, cap=1 << 18), CO.screen_reads(blob2, L, flanks, 31)), (v256, cut)
    finally:
        gf.set_option("screen_variant", 0)
        gf.set_option("bitmap_log2", 0)
        gf.set_option("screen_pf4_cap8", 0)


def test_human_scale_key_set_all_filter_kernels_agree_on_20M_reads(gf):
    """BASELINE.json configs[3]'s key set (19 840 gaps, 1.1e7 flank 16-mers, k=51: the level-1 bitmap gets 2^28 bits and the
    partitioned filter is chosen) on 20 M synthetic reads: the automatic choice, the forced partitioned filter and the plain
    kernel return the same hits, and every hit of the first 400 000 reads is the oracle's."""
    import torch
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    cfg = GapFill.synth_cfg(seed=20260004, scaffold_len=5_000_000, n_scaffolds=620, gaps_per_scaffold=32, gap_len=2000)
    gaps, flanks = GapFill.synth_layout(cfg)
    gf.set_gaps(gaps, 620, flanks)
    n_pairs, L, k = 10_000_000, 150, 51
    dev = torch.device("cuda:0")
    rb = B.lib().gf_packed_read_bytes(L)
    d_reads = torch.empty(2 * n_pairs * rb + 64, dtype=torch.uint8, device=dev)
    gf.synth_pairs_dev(cfg, 0, n_pairs, d_reads.data_ptr())
    gf.sync()
    packed = d_reads[:2 * n_pairs * rb].cpu().numpy().reshape(-1, rb)
    res = {}
    try:
        for variant in (0, 17, 9):
            gf.set_option("screen_variant", variant)
            res[variant] = gf.screen_reads(packed, L, k, cap=1 << 20)
        gf.set_option("screen_variant", 0)
        gf.set_option("screen_ext", 0)           # 16-base seeds as they are (no neighbour check in the resolve step)
        res[14] = gf.screen_reads(packed, L, k, cap=1 << 20)
    finally:
        gf.set_option("screen_variant", 0)
        gf.set_option("screen_ext", 1)
    assert len(res[0]) > 20_000 and _same(res[0], res[14]) and _same(res[0], res[9]) and _same(res[0], res[17])   # (0 = the 256-bucket 4-byte-pair filter, whole-line stores; 17 = unaligned runs)
    try:   # a short pair list: most pairs that are in the exact set take the serial path
        gf.set_option("screen_pf4_cap8", 1 << 16)
        assert _same(res[0], gf.screen_reads(packed, L, k, cap=1 << 20))
    finally:
        gf.set_option("screen_pf4_cap8", 0)
    m = 400_000
    exp = CO.screen_reads(CO.unpack_reads(packed[:m], L), L, flanks, k)
    head = res[0][res[0]["read"] < m]
    assert _same(head, exp) and len(exp) > 500


def test_device_second_hop_table_equals_the_host_sort(gf):
    """gf_second_hop_table_dev (rows cut from the tagger's hits and radix-sorted in HBM) == the reference's inversion + sort(1)
    (run_multi_threads_discordant.py:19-122) done with numpy on the same hits; the second hop and the pool keys through the
    device table give what the host-table variants give."""
    import torch
    from gappadder_amd import _lib as B
    c = S.small_case(seed=53, n_pairs=40000)
    gf.set_gaps(c["gaps"], c["n_scaffolds"], c["flanks"])
    recs = c["recs"]
    dev = torch.device("cuda:0")
    d_recs = torch.from_numpy(np.frombuffer(recs.tobytes(), dtype=np.uint8).copy()).to(dev)
    n, cap, row_cap = len(recs), 1 << 16, 1 << 12
    d_t = torch.zeros(cap * 12, dtype=torch.uint8, device=dev)
    d_l = torch.zeros(cap * 12, dtype=torch.uint8, device=dev)
    d_low = torch.zeros(n * 12, dtype=torch.uint8, device=dev)
    d_rows = torch.zeros(row_cap * 16, dtype=torch.uint8, device=dev)
    d_rg = torch.zeros(row_cap, dtype=torch.int32, device=dev)
    d_keys = torch.zeros(2, cap, dtype=torch.int64, device=dev)
    d_cnt = torch.zeros(16, dtype=torch.int32, device=dev)
    L, h, cp = B.lib(), gf.handle, d_cnt.data_ptr()
    assert L.gf_tag_alignments_low_dev(h, d_recs.data_ptr(), n, 300, 30, 250, 30, d_t.data_ptr(), cap, cp, d_low.data_ptr(), n, cp + 4) == 0
    assert L.gf_second_hop_table_dev(h, d_recs.data_ptr(), d_t.data_ptr(), cp, cap, d_rows.data_ptr(), d_rg.data_ptr(), row_cap, cp + 8) == 0
    assert L.gf_tag_low_mapq_table_dev(h, d_low.data_ptr(), cp + 4, n, d_rows.data_ptr(), cp + 8, row_cap, d_l.data_ptr(), cap, cp + 12) == 0
    gf.sync()
    cnt = d_cnt.cpu().numpy()
    n_t, n_rows, n_l = int(cnt[0]), int(cnt[2]), int(cnt[3])
    th = np.frombuffer(d_t[:n_t * 12].cpu().numpy().tobytes(), dtype=B.TAGHIT)
    disc = th[th["kind"] == B.KIND_DISCORDANT]
    exp = np.zeros(len(disc), dtype=B.DPOS)
    exp["mate_scaffold"], exp["mate_pos"] = recs["mate_ref"][disc["rec"]], recs["mate_pos"][disc["rec"]]
    exp["src_scaffold"], exp["src_gap"] = c["gaps"]["scaffold"][disc["gap"]], c["gaps"]["idx_in_scaffold"][disc["gap"]]
    order = ["mate_scaffold", "mate_pos", "src_scaffold", "src_gap"]
    exp = np.sort(exp, order=order)
    rows = np.frombuffer(d_rows[:n_rows * 16].cpu().numpy().tobytes(), dtype=B.DPOS)
    assert n_rows == len(exp) > 50
    key = rows["mate_scaffold"].astype(np.uint64) << np.uint64(32) | rows["mate_pos"].astype(np.uint64)
    assert np.all(key[1:] >= key[:-1])                                     # sorted by (scaffold, position) ...
    assert np.sort(rows, order=order).tobytes() == exp.tobytes()           # ... and the same rows as the host sort
    rg = d_rg[:n_rows].cpu().numpy()
    first = np.searchsorted(c["gaps"]["scaffold"], rows["src_scaffold"])   # gaps are grouped by scaffold
    assert np.array_equal(rg, first + rows["src_gap"] - 1)
    got_l = np.sort(np.frombuffer(d_l[:n_l * 12].cpu().numpy().tobytes(), dtype=B.TAGHIT), order=["rec", "gap", "kind"])
    assert _same(got_l, gf.tag_low_mapq(recs, rows.copy())) and n_l > 0
    # pool keys of the second-hop hits: through the device row->gap array == through the host table
    assert L.gf_pool_keys_reset(h, cp + 16) == 0 and L.gf_pool_keys_reset(h, cp + 20) == 0
    assert L.gf_pool_keys_from_second_hop_dev(h, d_recs.data_ptr(), d_l.data_ptr(), cp + 12, cap, d_rg.data_ptr(), d_keys[0].data_ptr(), cap, cp + 16) == 0
    rows_c = rows.copy()
    assert L.gf_pool_keys_from_tags_dev(h, d_recs.data_ptr(), d_l.data_ptr(), cp + 12, cap, B._p(rows_c), len(rows_c), d_keys[1].data_ptr(), cap, cp + 20) == 0
    gf.sync()
    cnt = d_cnt.cpu().numpy()
    assert int(cnt[4]) == int(cnt[5]) == n_l
    assert np.array_equal(np.sort(d_keys[0][:n_l].cpu().numpy()), np.sort(d_keys[1][:n_l].cpu().numpy()))
    # the three key producers in one launch (gf_pool_keys_all_dev): the same key multiset as the separate calls
    d_hits = torch.zeros(cap * 8, dtype=torch.uint8, device=dev)
    from gappadder_amd.hip_api import GapFill
    packed, _ = GapFill.pack_reads(c["reads_blob"], c["L"])
    d_reads = torch.from_numpy(np.ascontiguousarray(packed)).to(dev)
    assert L.gf_screen_reads_dev(h, d_reads.data_ptr(), None, len(packed), c["L"], 31, 1, d_hits.data_ptr(), cap, cp + 24) == 0
    d_k = torch.zeros(2, 4 * cap, dtype=torch.int64, device=dev)
    assert L.gf_pool_keys_reset(h, cp + 28) == 0
    assert L.gf_pool_keys_from_screen_dev(h, d_hits.data_ptr(), cp + 24, cap, 1, d_k[0].data_ptr(), 4 * cap, cp + 28) == 0
    assert L.gf_pool_keys_from_tags_dev(h, d_recs.data_ptr(), d_t.data_ptr(), cp, cap, None, 0, d_k[0].data_ptr(), 4 * cap, cp + 28) == 0
    assert L.gf_pool_keys_from_second_hop_dev(h, d_recs.data_ptr(), d_l.data_ptr(), cp + 12, cap, d_rg.data_ptr(), d_k[0].data_ptr(), 4 * cap, cp + 28) == 0
    assert L.gf_pool_keys_all_dev(h, d_hits.data_ptr(), cp + 24, cap, 1, d_recs.data_ptr(), d_t.data_ptr(), cp, cap, d_l.data_ptr(), cp + 12, cap,
                                  d_rg.data_ptr(), d_k[1].data_ptr(), 4 * cap, cp + 32) == 0
    gf.sync()
    cnt = d_cnt.cpu().numpy()
    assert int(cnt[7]) == int(cnt[8]) == 2 * int(cnt[6]) + n_t + n_l and int(cnt[6]) > 100
    assert np.array_equal(np.sort(d_k[0][:int(cnt[7])].cpu().numpy()), np.sort(d_k[1][:int(cnt[8])].cpu().numpy()))
    # a table that does not fit: the count says so
    assert L.gf_second_hop_table_dev(h, d_recs.data_ptr(), d_t.data_ptr(), cp, cap, d_rows.data_ptr(), d_rg.data_ptr(), 16, cp + 8) == 0
    gf.sync()
    assert int(d_cnt[2]) == n_rows > 16


def test_device_second_hop_table_without_discordant_hits(gf):
    """No DISCORDANT hit at all: the table is empty and the second hop finds nothing (and does not fault)."""
    import torch
    from gappadder_amd import _lib as B
    c = S.small_case(seed=54, n_pairs=2000)
    gf.set_gaps(c["gaps"], c["n_scaffolds"], c["flanks"])
    recs = c["recs"].copy()
    recs["mate_ref"] = recs["ref"]
    recs["tlen"] = 300                       # every pair concordant
    dev = torch.device("cuda:0")
    d_recs = torch.from_numpy(np.frombuffer(recs.tobytes(), dtype=np.uint8).copy()).to(dev)
    n, cap, row_cap = len(recs), 1 << 14, 256
    bufs = [torch.zeros(cap * 12, dtype=torch.uint8, device=dev) for _ in range(2)]
    d_low = torch.zeros(n * 12, dtype=torch.uint8, device=dev)
    d_rows = torch.zeros(row_cap * 16, dtype=torch.uint8, device=dev)
    d_rg = torch.zeros(row_cap, dtype=torch.int32, device=dev)
    d_cnt = torch.zeros(8, dtype=torch.int32, device=dev)
    L, h, cp = B.lib(), gf.handle, d_cnt.data_ptr()
    assert L.gf_tag_alignments_low_dev(h, d_recs.data_ptr(), n, 300, 30, 250, 30, bufs[0].data_ptr(), cap, cp, d_low.data_ptr(), n, cp + 4) == 0
    assert L.gf_second_hop_table_dev(h, d_recs.data_ptr(), bufs[0].data_ptr(), cp, cap, d_rows.data_ptr(), d_rg.data_ptr(), row_cap, cp + 8) == 0
    assert L.gf_tag_low_mapq_table_dev(h, d_low.data_ptr(), cp + 4, n, d_rows.data_ptr(), cp + 8, row_cap, bufs[1].data_ptr(), cap, cp + 12) == 0
    gf.sync()
    cnt = d_cnt.cpu().numpy()
    th = np.frombuffer(bufs[0][:int(cnt[0]) * 12].cpu().numpy().tobytes(), dtype=B.TAGHIT)
    assert not (th["kind"] == B.KIND_DISCORDANT).any() and int(cnt[1]) > 0
    assert int(cnt[2]) == 0 and int(cnt[3]) == 0


def test_device_built_flank_index_equals_the_host_built_one(gf):
    """The flank index is built on the device (csrc/index_dev.hip: extraction kernel, rocPRIM sorts, CAS-claimed tables); the host
    builder (option index_host = 1: std::sort + upload) is its comparator.  Same hits in every regime the index serves: k from 16
    to 64, min_hits > 1 (k-mer table) and == 1 (occurrence lists), the repeat mask, flanks with N, shorter than k, or empty,
    duplicated flanks — and both equal the oracle."""
    from gappadder_amd.hip_api import GapFill
    c = S.small_case(seed=61, n_pairs=6000)
    L = c["L"]
    flanks = [list(f) for f in c["flanks"]]
    flanks[1][0] = flanks[0][0]                                        # two gaps share a flank (k-mers with two gaps)
    flanks[2][1] = flanks[2][1][:100] + "N" + flanks[2][1][101:200] + "nnn" + flanks[2][1][203:]   # non-ACGT bytes split the runs
    flanks[3][0] = flanks[3][0][:20]                                   # shorter than every k > 20
    flanks[4][1] = ""                                                  # empty flank
    flanks[5][0] = "ACGT" * 80                                         # low complexity: the same k-mer many times in one flank
    flanks = [tuple(f) for f in flanks]
    packed, _ = GapFill.pack_reads(c["reads_blob"], L)
    os.environ["GF_DIAGNOSTICS"] = "1"                                 # the host builder is a test aid behind this switch
    plan = [(16, 1, 0), (31, 1, 0), (32, 1, 0), (33, 1, 0), (51, 1, 0), (64, 1, 0), (31, 3, 0), (41, 2, 0), (31, 1, 1), (31, 2, 2)]
    got = {}
    try:
        for host in (1, 0):
            gf.set_option("index_host", host)
            for k, mh, mg in plan:
                gf.set_option("max_gaps_per_kmer", mg)
                gf.set_gaps(c["gaps"], c["n_scaffolds"], flanks)
                got[(host, k, mh, mg)] = gf.screen_reads(packed, L, k, mh)
    finally:
        gf.set_option("index_host", 0)
        gf.set_option("max_gaps_per_kmer", 0)
        del os.environ["GF_DIAGNOSTICS"]
    total = 0
    for k, mh, mg in plan:
        a, b = got[(1, k, mh, mg)], got[(0, k, mh, mg)]
        assert _same(a, b), (k, mh, mg)
        assert _same(b, CO.screen_reads(c["reads_blob"], L, flanks, k, mh, mg)), (k, mh, mg)
        total += len(b)
    assert total > 2000


def test_flank_index_degenerate_inputs(gf):
    """No gaps, flanks without a single k-mer (all N, empty, lower case only), and homopolymer flanks: the device-built index
    must not fault and must agree with the oracle."""
    from gappadder_amd import _lib as B
    packed = np.zeros((1000, 38), dtype=np.uint8)            # 1000 poly-A reads
    blob = b"A" * (1000 * 150)
    gf.set_gaps(np.zeros(0, dtype=B.GAP), 1, [])
    assert len(gf.screen_reads(packed, 150, 31)) == 0
    g = np.zeros(3, dtype=B.GAP)
    g["start"], g["end"], g["idx_in_scaffold"] = [100, 1000, 5000], [200, 1100, 5100], [1, 2, 3]
    flanks = [("N" * 300, "ACGT"), ("", ""), ("acgt" * 10, "NNNN")]
    gf.set_gaps(g, 1, flanks)
    assert len(gf.screen_reads(packed, 150, 31)) == 0 and len(gf.screen_reads(packed, 150, 64, 2)) == 0
    flanks = [("A" * 300, "C" * 300), ("G" * 100, "T" * 100), ("ACGT" * 50, "A" * 40)]
    gf.set_gaps(g, 1, flanks)
    for k, mh in ((31, 1), (40, 1), (31, 5)):
        assert _same(gf.screen_reads(packed, 150, k, mh, cap=4096), CO.screen_reads(blob, 150, flanks, k, mh))
    assert len(gf.screen_reads(packed, 150, 31, cap=4096)) == 3000


def test_a_read_with_more_matches_than_the_verification_lists_is_reported_not_lost_silently(gf):
    """Reads that share their k-mers with hundreds of flanks: a poly-A read against 300 flanks that end in a poly-A run is listed in full
    by the third seed-and-extend pass (sets of 1 536 gaps per read) and equals the oracle.  The documented hard limit lies beyond that:
    against 1 700 such flanks the read falls to the k-mer-table pass, whose 15 000 (k-mer position, gap) entries it exceeds (120 x 1 700) —
    the host call says GF_E_UNSUPPORTED, the device call counts the read (gf_screen_last_overflow) —, and with the repeat mask
    (max_gaps_per_kmer) the same reads go through and equal the oracle."""
    import ctypes as C
    import torch
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    rng = np.random.RandomState(5)
    lut = np.frombuffer(b"ACGT", np.uint8)
    n_all, L = 1700, 150
    gaps = np.zeros(n_all, dtype=B.GAP)
    gaps["scaffold"] = 0
    gaps["start"] = (np.arange(n_all) + 1) * 20000
    gaps["end"] = gaps["start"] + 500
    gaps["idx_in_scaffold"] = np.arange(n_all) + 1
    flanks = [(lut[rng.randint(0, 4, 255)].tobytes().decode() + "A" * 40, lut[rng.randint(0, 4, 295)].tobytes().decode()) for _ in range(n_all)]
    reads = [lut[rng.randint(0, 4, L)].tobytes() for _ in range(2000)]
    for i in range(0, 64, 2):
        reads[i] = b"A" * L
    for g in range(0, 40):                                     # ordinary hits beside them
        reads[100 + g] = flanks[g][0][60:60 + L].encode()
    blob = b"".join(reads)
    packed, _ = GapFill.pack_reads(blob, L)
    nd = C.c_size_t(0)
    gf.set_gaps(gaps[:300], 1, flanks[:300])                   # 300 gaps per poly-A read: the seed-and-extend passes list them all
    exp300 = CO.screen_reads(blob, L, flanks[:300], 31)
    assert np.bincount(exp300["read"]).max() == 300
    assert _same(gf.screen_reads(packed, L, 31, cap=len(exp300) + 64), exp300)
    assert B.lib().gf_screen_last_overflow(gf.handle, C.byref(nd)) == 0 and nd.value == 0
    gf.set_gaps(gaps, 1, flanks)                               # 1 700: beyond every list
    with pytest.raises(B.GapFillError):
        gf.screen_reads(packed, L, 31, cap=1 << 20)
    d_reads = torch.from_numpy(packed.reshape(-1).copy()).cuda()
    d_out = torch.zeros(1 << 20, 2, dtype=torch.int32, device="cuda")
    d_n = torch.zeros(4, dtype=torch.int32, device="cuda")
    assert B.lib().gf_screen_reads_dev(gf.handle, d_reads.data_ptr(), None, len(reads), L, 31, 1, d_out.data_ptr(), d_out.shape[0], d_n.data_ptr()) == 0
    assert B.lib().gf_screen_last_overflow(gf.handle, C.byref(nd)) == 0 and nd.value == 32
    try:
        gf.set_option("max_gaps_per_kmer", 8)
        gf.set_gaps(gaps, 1, flanks)
        exp = CO.screen_reads(blob, L, flanks, 31, max_gaps_per_kmer=8)
        assert _same(gf.screen_reads(packed, L, 31), exp) and len(exp) >= 40
        assert B.lib().gf_screen_last_overflow(gf.handle, C.byref(nd)) == 0 and nd.value == 0
    finally:
        gf.set_option("max_gaps_per_kmer", 0)
