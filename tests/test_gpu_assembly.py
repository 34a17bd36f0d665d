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
    for mc, ml in ((1, 40), (3, 40), (2, 29), (2, 200)):
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
        assert len(exp) > 20, (k, kv)
        assert got[(0, k, kv)] == exp, (k, kv)


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
