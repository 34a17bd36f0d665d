"""GPU parity of the per-gap assembly (gf_assemble / gf_count_kmers through the C ABI) against the oracle's definition."""
import numpy as np
import pytest

from oracle import c_oracle as CO
import synth_small as S
from test_assembly_oracle import LUT, rc, tiled_reads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gf():
    from gappadder_amd.hip_api import GapFill
    g = GapFill(0)
    yield g
    g.close()


def _pools_from_case(c, k=31):
    """Per-gap pools = reads the oracle's screen recruits (plus their mates), as the pipeline would build them."""
    hits = CO.screen_reads(c["reads_blob"], c["L"], c["flanks"], k)
    L = c["L"]
    pools = []
    for g in range(len(c["gaps"])):
        ids = sorted(set(int(h["read"]) for h in hits if h["gap"] == g) | set(int(h["read"]) ^ 1 for h in hits if h["gap"] == g))
        pools.append(b"".join(c["reads_blob"][i * L:(i + 1) * L] for i in ids))
    return pools


def _gpu_assemble(gf, pools, L, kk, **kw):
    from gappadder_amd.hip_api import GapFill
    blob = b"".join(pools)
    packed, nm = GapFill.pack_reads(blob, L, with_mask=True)
    off = np.cumsum([0] + [len(p) // L for p in pools]).astype(np.uint64)
    ctg, seq = gf.assemble(packed, off, L, kk, n_mask=nm if b"N" in blob else None, **kw)
    out = {}
    for c in ctg:
        out.setdefault((int(c["gap"]), int(c["k"]), int(c["kv"])), []).append(
            (seq[int(c["seq_off"]):int(c["seq_off"]) + int(c["length"])].decode(), int(c["n_nodes"]), int(c["cov_sum"])))
    return out, ctg


@pytest.mark.parametrize("seed,L,kk", [(1, 150, [(31, 29)]), (2, 150, [(41, 39), (41, 37)]), (3, 100, [(31, 29), (51, 49)]),
                                       (4, 150, [(30, 29), (40, 37), (50, 47)])])
def test_assemble_matches_oracle(gf, seed, L, kk):
    c = S.small_case(seed=seed, n_pairs=12000, L=L, insert=max(300, L + 100), n_frac=0.05 if seed == 3 else 0.0)
    pools = _pools_from_case(c)
    got, ctg = _gpu_assemble(gf, pools, L, kk)
    total = 0
    for g, p in enumerate(pools):
        for (k, kv) in kk:
            exp = CO.assemble_pool(p, L, k, kv)
            assert got.get((g, k, kv), []) == exp, (g, k, kv)
            total += len(exp)
    assert total > 20
    # output order: gap, then the (k, kv) pairs in the order given
    order = [(int(x["gap"]), kk.index((int(x["k"]), int(x["kv"])))) for x in ctg]
    assert order == sorted(order)


def test_known_answers_on_gpu(gf):
    rng = np.random.RandomState(5)
    g = LUT[rng.randint(0, 4, 3000)].tobytes()
    L = 150
    reads = tiled_reads(g, L, 700, rng) + [g[:L], g[-L:], g[:L], g[-L:]]
    pools = [b"".join(reads), b"", b"".join(reads[:3]), reads[0] * 2]
    got, _ = _gpu_assemble(gf, pools, L, [(31, 29)])
    assert got[(0, 31, 29)][0][0].encode() == min(g, rc(g)) and len(got[(0, 31, 29)]) == 1
    assert (1, 31, 29) not in got                       # empty pool
    for i, p in enumerate(pools):
        assert got.get((i, 31, 29), []) == CO.assemble_pool(p, L, 31, 29)
    assert got[(3, 31, 29)][0][0].encode() == min(reads[0], rc(reads[0]))   # a duplicated read assembles to itself


def test_min_count_and_min_contig_parameters(gf):
    c = S.small_case(seed=8, n_pairs=8000)
    pools = _pools_from_case(c)
    for mc, ml in ((1, 40), (3, 40), (4, 40), (6, 40), (2, 29), (2, 200)):
        got, _ = _gpu_assemble(gf, pools, c["L"], [(31, 29)], min_count=mc, min_contig=ml)
        for g, p in enumerate(pools):
            assert got.get((g, 31, 29), []) == CO.assemble_pool(p, c["L"], 31, 29, mc, ml), (mc, ml, g)


def test_pool_larger_than_the_lds_stage_uses_global_reads(gf):
    rng = np.random.RandomState(12)
    g = LUT[rng.randint(0, 4, 5000)].tobytes()
    L = 150
    reads = tiled_reads(g, L, 2500, rng)      # 2500 x 38 B = 95 KB > the LDS share of the staged pool
    for i in (3, 77, 500):
        b = bytearray(reads[i]); b[40] = ord("A") if b[40] != ord("A") else ord("G"); reads[i] = bytes(b)
    pool = b"".join(reads)
    got, _ = _gpu_assemble(gf, [pool], L, [(31, 29)])
    assert got[(0, 31, 29)] == CO.assemble_pool(pool, L, 31, 29)
    for kb in (8, 64):      # nothing fits in LDS / only the node arrays do
        gf.set_option("asm_lds_pool_kb", kb)
        try:
            got2, _ = _gpu_assemble(gf, [pool], L, [(31, 29)])
        finally:
            gf.set_option("asm_lds_pool_kb", 152)
        assert got2 == got


@pytest.mark.parametrize("n_reads,kk", [(1034, [(31, 29)]), (2100, [(31, 29)]), (1034, [(41, 39), (51, 49)]), (700, [(63, 61), (64, 61)])])
def test_deep_noisy_pool(gf, n_reads, kk):
    """150-400x depth with 1 % errors: tens of thousands of error k-mers pass min_count 2, the node arrays fill the LDS and
    the per-contig coverage sums fall back to the global walk records (once an ASM_ERR_WALKS_PAR overflow)."""
    rng = np.random.RandomState(n_reads)
    g = LUT[rng.randint(0, 4, 2600)].tobytes()
    L = 150
    reads = tiled_reads(g, L, n_reads, rng)
    for i in range(len(reads)):
        b = bytearray(reads[i])
        for pos in np.flatnonzero(rng.rand(L) < 0.01):
            b[pos] = LUT[(list(b"ACGT").index(b[pos]) + 1 + rng.randint(3)) % 4]
        reads[i] = bytes(b)
    pool = b"".join(reads)
    got, _ = _gpu_assemble(gf, [pool], L, kk)     # k <= 31 / 32 < k <= 63: key-slot count phase in the global table; 64: instance ids
    for (k, kv) in kk:
        exp = CO.assemble_pool(pool, L, k, kv)
        assert got[(0, k, kv)] == exp, (k, kv)
    gf.set_option("asm_simplify", 0)          # raw unitigs: the errors seen twice split the graph into dozens of pieces
    try:
        raw, _ = _gpu_assemble(gf, [pool], L, kk)
    finally:
        gf.set_option("asm_simplify", 8)
    for (k, kv) in kk:
        exp = CO.assemble_pool(pool, L, k, kv, simplify=0)
        assert len(exp) > 20, (k, kv)
        assert raw[(0, k, kv)] == exp, (k, kv)


@pytest.mark.parametrize("n_reads,kk,min_count", [(320, [(31, 29), (51, 49)], 2), (660, [(31, 29), (41, 39), (51, 49)], 2),
                                                   (320, [(31, 29), (51, 49)], 3), (1500, [(51, 49)], 3), (90, [(63, 61)], 2)])
def test_pre_count_leaves_the_counts_unchanged(gf, n_reads, kk, min_count):
    """The bit-array pre-count of the count phase (k-mers seen fewer than min_count times never reach the table) against the
    oracle and against the same launch without it, at the depths of C4 / C5 pools, with N bases in the reads."""
    rng = np.random.RandomState(n_reads + min_count)
    g = LUT[rng.randint(0, 4, 2600)].tobytes()
    L = 150
    reads = tiled_reads(g, L, n_reads, rng)
    for i in range(len(reads)):
        b = bytearray(reads[i])
        for pos in np.flatnonzero(rng.rand(L) < 0.01):
            b[pos] = LUT[(list(b"ACGT").index(b[pos]) + 1 + rng.randint(3)) % 4]
        if i % 37 == 0:
            b[rng.randint(L)] = ord("N")
        reads[i] = bytes(b)
    pool = b"".join(reads)
    got, _ = _gpu_assemble(gf, [pool, pool[:40 * L]], L, kk, min_count=min_count)
    gf.set_option("asm_precount", 0)
    try:
        plain, _ = _gpu_assemble(gf, [pool, pool[:40 * L]], L, kk, min_count=min_count)
    finally:
        gf.set_option("asm_precount", 1)
    assert got == plain
    for (k, kv) in kk:
        assert got.get((0, k, kv), []) == CO.assemble_pool(pool, L, k, kv, min_count=min_count), (k, kv)
        assert len(got.get((0, k, kv), [])) >= 1


def test_count_kmers_matches_oracle(gf):
    from gappadder_amd.hip_api import GapFill
    c = S.small_case(seed=9, n_pairs=6000, n_frac=0.1)
    pool = _pools_from_case(c)[2]
    L = c["L"]
    packed, nm = GapFill.pack_reads(pool, L, with_mask=True)
    for k, mc in ((31, 2), (41, 2), (64, 1), (16, 3)):
        km, cn = gf.count_kmers(packed, L, k, mc, n_mask=nm)
        hi, lo, cnt = CO.count_kmers(pool, L, k, mc)
        assert (km[:, 0] == hi).all() and (km[:, 1] == lo).all() and (cn == cnt).all() and len(hi) > 10


def test_even_kv_is_rejected(gf):
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    packed, _ = GapFill.pack_reads(b"ACGT" * 25 * 4, 100)
    with pytest.raises(B.GapFillError) as e:
        gf.assemble(packed, np.array([0, 4], np.uint64), 100, [(31, 28)])
    assert e.value.code == B.GF_E_UNSUPPORTED


def _kat_pools():
    """The hand-built error-removal cases of tests/test_assembly_oracle.py as pools: SNP bubble, short tip, long dead end,
    two-copy repeat, overlapping bubbles, three alleles, noisy 40x pool; (k, kv) per pool."""
    from test_assembly_oracle import _cover, _mut
    pools = []
    rng = np.random.RandomState(17)
    g = LUT[rng.randint(0, 4, 1000)].tobytes()
    pools.append((100, (31, 29), _cover(g, 100) + _cover(_mut(g, 500)[430:590], 100, step=3)))
    rng = np.random.RandomState(18)
    g = LUT[rng.randint(0, 4, 900)].tobytes()
    tip = g[340:420] + LUT[rng.randint(0, 4, 20)].tobytes()
    pools.append((100, (31, 29), _cover(g, 100) + [tip, tip]))
    rng = np.random.RandomState(19)
    g = LUT[rng.randint(0, 4, 900)].tobytes()
    pools.append((100, (31, 29), _cover(g, 100) + _cover(g[300:420] + LUT[rng.randint(0, 4, 80)].tobytes(), 100, step=5)))
    rng = np.random.RandomState(8)
    rep = LUT[rng.randint(0, 4, 60)].tobytes()
    a, b, c = (LUT[rng.randint(0, 4, 300)].tobytes() for _ in range(3))
    g = a + rep + b + rep + c
    pools.append((100, (31, 29), tiled_reads(g, 100, 500, rng) + [g[:100], g[-100:]] * 2))
    rng = np.random.RandomState(21)
    g = LUT[rng.randint(0, 4, 1000)].tobytes()
    pools.append((100, (31, 29), _cover(g, 100) + _cover(_mut(g, 500)[430:580], 100, step=3) + _cover(_mut(g, 512, 2)[440:600], 100, step=3)))
    rng = np.random.RandomState(22)
    g = LUT[rng.randint(0, 4, 800)].tobytes()
    pools.append((100, (31, 29), _cover(g, 100) + _cover(_mut(g, 400, 1)[330:490], 100, step=3) + _cover(_mut(g, 400, 2)[330:490], 100, step=3)))
    rng = np.random.RandomState(23)
    g = LUT[rng.randint(0, 4, 1200)].tobytes()
    tip = g[500:630] + LUT[rng.randint(0, 4, 20)].tobytes()
    pools.append((150, (51, 49), _cover(g, 150) + _cover(_mut(g, 300)[200:420], 150, step=3) + [tip, tip, rc(tip)]))
    rng = np.random.RandomState(24)
    g = LUT[rng.randint(0, 4, 2600)].tobytes()
    reads = []
    for _ in range(int(40 * len(g) / 150)):
        s = rng.randint(0, len(g) - 150 + 1)
        r = bytearray(g[s:s + 150])
        for p in np.nonzero(rng.rand(150) < 0.005)[0]:
            r[p] = b"ACGT"[(b"ACGT".index(bytes([r[p]])) + 1 + rng.randint(3)) % 4]
        reads.append(rc(bytes(r)) if rng.randint(2) else bytes(r))
    pools.append((150, (31, 29), reads))
    pools.append((150, (41, 39), reads))
    return pools


@pytest.mark.parametrize("simplify", [0, 1, 2, 4])
def test_error_removal_known_answers_on_gpu(gf, simplify):
    """Tip clipping + bubble popping (Velvet's defaults as oracle/gp_oracle.c defines them) in the assembly kernel: every
    hand-built case, every number of rounds, bit-exact vs the oracle; the LDS plan and the all-global plan agree."""
    gf.set_option("asm_simplify", simplify)
    try:
        for L in (100, 150):
            for kk in ((31, 29), (51, 49), (41, 39)):
                group = [b"".join(r) for (l, k2, r) in _kat_pools() if l == L and k2 == kk]
                if not group:
                    continue
                got, _ = _gpu_assemble(gf, group, L, [kk])
                for i, p in enumerate(group):
                    exp = CO.assemble_pool(p, L, kk[0], kk[1], simplify=simplify)
                    assert got.get((i, kk[0], kk[1]), []) == exp, (L, kk, i, simplify)
                gf.set_option("asm_lds_pool_kb", 8)        # nothing fits in LDS: graph, pairs and table in global memory
                try:
                    got2, _ = _gpu_assemble(gf, group, L, [kk])
                finally:
                    gf.set_option("asm_lds_pool_kb", 152)
                assert got2 == got, (L, kk, simplify)
        if simplify >= 1:
            snp = [b"".join(r) for (l, k2, r) in _kat_pools() if l == 100][0]
            assert len(_gpu_assemble(gf, [snp], 100, [(31, 29)])[0][(0, 31, 29)]) == 1     # the SNP bubble is one contig now
    finally:
        gf.set_option("asm_simplify", 8)


def test_random_error_graphs_match_oracle(gf):
    """Many small pools with planted substitutions seen twice at random places (bubbles, tips at read ends, clusters closer than
    kv): GPU == oracle at the default two rounds, pool by pool."""
    rng = np.random.RandomState(99)
    L = 100
    pools = []
    for _ in range(60):
        g = LUT[rng.randint(0, 4, rng.randint(300, 900))].tobytes()
        reads = []
        for s in range(0, len(g) - L + 1, 5):
            reads += [g[s:s + L], rc(g[s:s + L])]
        for _e in range(rng.randint(1, 6)):
            s = rng.randint(0, len(g) - L + 1)
            r = bytearray(g[s:s + L])
            for _m in range(rng.randint(1, 3)):
                p = rng.randint(0, L)
                r[p] = b"ACGT"[(b"ACGT".index(bytes([r[p]])) + 1 + rng.randint(3)) % 4]
            reads += [bytes(r)] * 2
        pools.append(b"".join(reads))
    got, _ = _gpu_assemble(gf, pools, L, [(31, 29), (41, 39)])
    gf.set_option("asm_tiebreak", 0)
    try:
        got_none, _ = _gpu_assemble(gf, pools, L, [(31, 29), (41, 39)])      # the reference-shaped mode: no counts in the tie-break
    finally:
        gf.set_option("asm_tiebreak", 1)
    n_diff_raw = n_diff_mode = 0
    for i, p in enumerate(pools):
        for k, kv in ((31, 29), (41, 39)):
            exp = CO.assemble_pool(p, L, k, kv)
            assert got.get((i, k, kv), []) == exp, (i, k, kv)
            exp_none = CO.assemble_pool(p, L, k, kv, tiebreak="none")
            assert got_none.get((i, k, kv), []) == exp_none, (i, k, kv, "none")
            n_diff_raw += exp != CO.assemble_pool(p, L, k, kv, simplify=0)
            n_diff_mode += exp != exp_none
    assert n_diff_raw > 30      # the removal did something in most pools
    assert n_diff_mode > 5      # ... and the two tie-break modes part ways in some of them


@pytest.mark.parametrize("kk,L", [((31, 29), 100), ((51, 49), 150), ((41, 37), 150)])
def test_weak_kmers_lose_the_ties_of_the_error_removal(gf, kk, L):
    """An error seen two, three and four times against the true allele (oracle: test_an_error_seen_at_most_min_count_plus_one_...):
    the kernel keeps the true allele like the oracle — key-slot, fingerprint and instance-id count tables (k 31 / 51, kv = k - 4 runs
    without node fingerprints), LDS and global plans, min_count 2, 3, 4 (weakness on the true count: the slot forms with their 2-bit
    counters run at min_count <= 2 only) and 1 — and in the reference-shaped mode (asm_tiebreak = 0: no counts) like the oracle's
    tiebreak="none"."""
    from test_assembly_oracle import _cover, _mut
    rng = np.random.RandomState(41)
    pools, truth = [], []
    for trial in range(16):
        g = LUT[rng.randint(0, 4, 900)].tobytes()
        h = _mut(g, 450, 1 + trial % 3)
        err = h[450 - L // 2:450 + L // 2]
        reads = _cover(g, L) + [err, rc(err)] + [err] * (0 if trial < 6 else 1 if trial < 12 else 2)    # seen 2 / 3 / 4 times
        pools.append(b"".join(reads))
        truth.append(g)
    n_modes_differ = 0
    for mc, tb in ((2, "counts"), (3, "counts"), (4, "counts"), (1, "counts"), (2, "none"), (3, "none")):
        for lds_kb in (152, 8):
            gf.set_option("asm_lds_pool_kb", lds_kb)
            gf.set_option("asm_tiebreak", 1 if tb == "counts" else 0)
            try:
                got, _ = _gpu_assemble(gf, pools, L, [kk], min_count=mc)
            finally:
                gf.set_option("asm_lds_pool_kb", 152)
                gf.set_option("asm_tiebreak", 1)
            for i, p in enumerate(pools):
                exp = CO.assemble_pool(p, L, kk[0], kk[1], min_count=mc, tiebreak=tb)
                assert got.get((i, kk[0], kk[1]), []) == exp, (i, mc, tb, lds_kb)
                if mc == 2 and tb == "counts" and i < 12:
                    assert len(exp) == 1 and exp[0][0].encode() in (truth[i], rc(truth[i]))
                if mc == 3 and tb == "counts" and 6 <= i < 12:       # seen three times at min_count 3: survives, weak (<= 4), loses
                    assert any(t[400:500] in exp[0][0].encode() for t in (truth[i], rc(truth[i]))), i
                if tb == "none" and lds_kb == 152:
                    n_modes_differ += exp != CO.assemble_pool(p, L, kk[0], kk[1], min_count=mc)
    assert n_modes_differ > 0       # the switch is alive: somewhere the sequence order picks the error allele


@pytest.mark.parametrize("threads", [512, 256])
def test_two_and_four_gaps_per_cu_give_the_same_contigs(gf, threads):
    """option asm_threads: 512 / 256 threads per gap (two / four gaps per CU, 76 / 38 KiB of LDS each) against the oracle — small pools
    in LDS, the noisy deep pools that fall back to the global slice at those budgets, k <= 32 and k > 32."""
    c = S.small_case(seed=5, n_pairs=12000, L=150, insert=300)
    pools = _pools_from_case(c)
    deep = [b"".join(r) for (l, k2, r) in _kat_pools() if l == 150]
    gf.set_option("asm_threads", threads)
    try:
        got, _ = _gpu_assemble(gf, pools, 150, [(31, 29), (51, 49)])
        got2, _ = _gpu_assemble(gf, deep, 150, [(31, 29), (41, 39)])
    finally:
        gf.set_option("asm_threads", 0)
    n = 0
    for g, p in enumerate(pools):
        for k, kv in ((31, 29), (51, 49)):
            exp = CO.assemble_pool(p, 150, k, kv)
            assert got.get((g, k, kv), []) == exp, (g, k, kv)
            n += len(exp)
    for g, p in enumerate(deep):
        for k, kv in ((31, 29), (41, 39)):
            assert got2.get((g, k, kv), []) == CO.assemble_pool(p, 150, k, kv), (g, k, kv)
    assert n > 20


@pytest.mark.parametrize("kk", [(31, 29), (51, 49)])
def test_pools_beyond_the_callers_bound_take_the_second_launch(gf, kk):
    """asm_max_pool_reads bounds the workspace slice of the main launch; a deeper pool (a flank inside a repeat) is not an error:
    it is listed and assembled by the second launch with slices of asm_big_pool_reads rows — same contigs as the oracle, no
    gap_error.  A pool beyond asm_big_pool_reads is the documented limit: its gap_error is set and the host call reports the pool."""
    rng = np.random.RandomState(kk[0])
    L = 150
    pools = []
    for i in range(24):
        g = LUT[rng.randint(0, 4, 600 + 40 * i)].tobytes()
        depth = 4 if i % 5 else 40                      # every fifth pool is ten times as deep
        reads = []
        for _ in range(int(depth * len(g) / L) + 4):
            s0 = rng.randint(0, len(g) - L + 1)
            r = bytearray(g[s0:s0 + L])
            for p in np.nonzero(rng.rand(L) < 0.004)[0]:
                r[p] = b"ACGT"[(b"ACGT".index(bytes([r[p]])) + 1 + rng.randint(3)) % 4]
            reads.append(rc(bytes(r)) if rng.randint(2) else bytes(r))
        pools.append(b"".join(reads))
    sizes = [len(p) // L for p in pools]
    assert min(sizes) < 40 and max(sizes) > 300
    want = [CO.assemble_pool(p, L, kk[0], kk[1]) for p in pools]
    try:
        for bound in (64, 1):                            # some / all pools beyond the bound
            gf.set_option("asm_max_pool_reads", bound)
            got, _ = _gpu_assemble(gf, pools, L, [kk])
            for i in range(len(pools)):
                assert got.get((i, kk[0], kk[1]), []) == want[i], (bound, i, sizes[i])
        # The deep pools are beyond the last launch's slices too.  The HOST call knows its pools and sizes those slices for its deepest one
        # (an option a device pipeline left on the context only ever gets raised: the CLI's later rounds share the context with it) ...
        gf.set_option("asm_big_pool_reads", 200)
        gf.set_option("asm_max_pool_reads", 64)
        got, _ = _gpu_assemble(gf, pools, L, [kk])
        for i in range(len(pools)):
            assert got.get((i, kk[0], kk[1]), []) == want[i], ("host, small option", i, sizes[i])
        # ... the DEVICE entry point takes the option as it stands: a pool beyond it is the documented limit — its gap_error is set, the
        # others are assembled
        import ctypes as C
        import torch
        from gappadder_amd import _lib as B
        lib = B.lib()
        from gappadder_amd.hip_api import GapFill
        packed, _ = GapFill.pack_reads(b"".join(pools), L)
        off = np.cumsum([0] + sizes).astype(np.int64)
        d_pool = torch.from_numpy(packed.reshape(-1).copy()).cuda()
        d_off = torch.from_numpy(off).cuda()
        d_ctg = torch.zeros(8192 * 32, dtype=torch.uint8, device="cuda")
        d_seq = torch.zeros(1 << 22, dtype=torch.uint8, device="cuda")
        d_cnt = torch.zeros(8, dtype=torch.int32, device="cuda")
        d_err = torch.zeros(len(pools), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        assert lib.gf_assemble_dev(gf.handle, d_pool.data_ptr(), None, d_off.data_ptr(), len(pools), int(off[-1]), L, kk[0], kk[1], 2, 40,
                                   d_ctg.data_ptr(), 8192, d_cnt.data_ptr(), d_seq.data_ptr(), 1 << 22, d_cnt.data_ptr() + 8, d_err.data_ptr()) == 0
        gf.sync()
        err = d_err.cpu().numpy()
        assert [bool(e) for e in err] == [n > 200 for n in sizes] and any(err) and not all(err)
    finally:
        gf.set_option("asm_max_pool_reads", 0)
        gf.set_option("asm_big_pool_reads", 131072)


def test_gaps_that_do_not_fit_half_a_cu_move_to_the_whole_cu_launch(gf):
    """With thousands of gaps the main launch runs two gaps per CU (512 threads, 76 KiB of LDS each); a pool too deep for its count
    table to stay in that LDS is listed unseen, a pool whose table runs full or whose graph will not fit (shallow coverage of a long
    region: many distinct k-mers per read) when that shows, and a middle launch with 1 024 threads and a whole CU's LDS per gap takes
    the list — instead of the global-memory plans that cost three to six times as much.  Same contigs as the launch that gives every
    gap a whole CU, and as the oracle."""
    import ctypes as C
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    rng = np.random.RandomState(77)
    L, n_pools = 150, 2304
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    pools = []
    for i in range(n_pools):
        kind = i % 8
        if kind < 5:   glen, depth = 1200 + 100 * (i % 13), 25          # the usual gap: 200-400 reads
        elif kind < 7: glen, depth = 2400, 32                           # a deep pool: > 484 reads (listed unseen at k = 51)
        else:          glen, depth = 9000, 6                            # shallow coverage of a long region: 360 reads, > 8 000 distinct k-mers
        g = LUT[rng.randint(0, 4, glen)]
        n = int(depth * glen / L)
        st = rng.randint(0, glen - L + 1, n)
        r = g[st[:, None] + np.arange(L)[None, :]]
        err = rng.rand(n, L) < 0.004
        r = np.where(err, LUT[(np.searchsorted(LUT, r) + 1 + rng.randint(0, 3, (n, L))) % 4], r)
        flip = rng.randint(0, 2, n).astype(bool)
        r[flip] = comp[r[flip][:, ::-1]]
        pools.append(np.ascontiguousarray(r).tobytes())
    sizes = np.array([len(p) // L for p in pools])
    assert sizes.max() < 640 and (sizes > 484).sum() > 400
    kk = [(51, 49)]
    got, _ = _gpu_assemble(gf, pools, L, kk)
    t, m, l = C.c_int(0), C.c_uint32(0), C.c_uint32(0)
    assert B.lib().gf_assemble_last_launch(gf.handle, C.byref(t), C.byref(m), C.byref(l)) == 0
    import torch
    if torch.cuda.get_device_properties(0).multi_processor_count * 8 <= n_pools:      # (MI355X: 256 CUs)
        assert t.value == 512 and m.value > (sizes > 484).sum() + 100 and l.value == 0, (t.value, m.value, l.value)
    gf.set_option("asm_threads", 1024)
    try:
        ref, _ = _gpu_assemble(gf, pools, L, kk)
    finally:
        gf.set_option("asm_threads", 0)
    assert got == ref and len(got) >= n_pools
    for i in list(range(0, 64)) + list(range(n_pools - 16, n_pools)):
        assert got.get((i, 51, 49), []) == CO.assemble_pool(pools[i], L, 51, 49), (i, int(sizes[i]))
