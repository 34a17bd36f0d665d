"""GPU tests of the multi-library / multi-GPU plumbing and of the BASELINE.json configurations at their full sizes
(§8e): owner-rank exchange kernels, library merge order, multi-k assembly call, device flank
anchoring vs the host picker, the 2-rank owner assembly vs the single-process run, and bench.py's sample parity at full size."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _dev(t):
    import torch
    return torch.from_numpy(np.array(t, copy=True, order="C")).cuda()


def _random_pools(rng, n_gaps, rb, max_rows):
    cnt = rng.integers(0, max_rows + 1, size=n_gaps)
    cnt[rng.integers(0, n_gaps, size=max(1, n_gaps // 5))] = 0          # empty pools
    off = np.zeros(n_gaps + 1, dtype=np.uint64)
    off[1:] = np.cumsum(cnt)
    rows = rng.integers(0, 256, size=(int(off[-1]), rb), dtype=np.uint8)
    return rows, off


@pytest.mark.parametrize("world,n_lib,n_gaps,batch,L", [(3, 1, 37, 4, 150), (4, 2, 1030, 256, 150), (2, 2, 19, 3, 101), (1, 3, 50, 256, 150)])
def test_owner_exchange_and_library_merge_kernels(world, n_lib, n_gaps, batch, L):
    """gf_pools_pack_for_owners_dev + (the all-to-all, done here by slicing) + gf_pools_merge_dev == the definition: every gap's
    rows at its owner, libraries in order, inside a library the source ranks in order (merge_reads.py:43-51)."""
    import torch
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    lib = B.lib()
    rb = lib.gf_packed_read_bytes(L)
    rng = np.random.default_rng(world * 100 + n_lib)
    gf = GapFill(0)
    h = gf.handle
    pools = [[_random_pools(rng, n_gaps, rb, 40) for _ in range(n_lib)] for _ in range(world)]     # [rank][lib] -> (rows, off)
    owner = (np.arange(n_gaps) // batch) % world
    cap = 1 + max(int(sum(int(off[g + 1] - off[g]) for g in range(n_gaps) if owner[g] == d))
                  for r in range(world) for (_, off) in pools[r] for d in range(world))
    sends, cnts = [], []
    for r in range(world):
        d_send = torch.zeros(world * n_lib * cap * rb, dtype=torch.uint8, device="cuda")
        d_cnt = torch.zeros(n_lib * n_gaps, dtype=torch.int32, device="cuda")
        d_err = torch.zeros(1, dtype=torch.int32, device="cuda")
        for l, (rows, off) in enumerate(pools[r]):
            d_rows, d_off = _dev(rows.reshape(-1) if len(rows) else np.zeros(1, np.uint8)), _dev(off.view(np.int64))
            assert lib.gf_pools_pack_for_owners_dev(h, d_rows.data_ptr(), d_off.data_ptr(), n_gaps, L, world, batch, l, n_lib,
                                                    d_send.data_ptr(), cap, d_cnt.data_ptr() + 4 * l * n_gaps, d_err.data_ptr()) == 0
        gf.sync()
        assert int(d_err[0]) == 0
        sends.append(d_send.cpu().numpy().reshape(world, n_lib * cap * rb))
        cnts.append(d_cnt.cpu().numpy())
        for l, (rows, off) in enumerate(pools[r]):
            assert (cnts[-1][l * n_gaps:(l + 1) * n_gaps] == np.diff(off.astype(np.int64))).all()
    all_cnt = np.concatenate(cnts)                                          # [rank][lib][gap] = all_gather_into_tensor
    total = 0
    for me in range(world):
        recv = np.concatenate([sends[src][me] for src in range(world)])    # all_to_all_single with equal slots
        want_rows, want_off = [], [0]
        for g in range(n_gaps):
            if owner[g] == me:
                for l in range(n_lib):
                    for r in range(world):
                        rows, off = pools[r][l]
                        want_rows.append(rows[int(off[g]):int(off[g + 1])])
            want_off.append(sum(len(x) for x in want_rows))
        want = np.concatenate(want_rows) if want_rows else np.zeros((0, rb), np.uint8)
        mcap = len(want) + 3
        d_merged = torch.zeros(mcap * rb + 8, dtype=torch.uint8, device="cuda")
        d_moff = torch.zeros(n_gaps + 1, dtype=torch.int64, device="cuda")
        d_err = torch.zeros(1, dtype=torch.int32, device="cuda")
        assert lib.gf_pools_merge_dev(h, _dev(recv).data_ptr(), cap, _dev(all_cnt).data_ptr(), n_lib, world, n_gaps, L, me, world, batch,
                                      d_merged.data_ptr(), mcap, d_moff.data_ptr(), d_err.data_ptr()) == 0
        gf.sync()
        assert int(d_err[0]) == 0
        assert d_moff.cpu().numpy().tolist() == want_off
        assert d_merged.cpu().numpy()[:len(want) * rb].tobytes() == want.tobytes()
        total += len(want)
        # a merged buffer that is too small is flagged, never overrun
        if len(want) > 4:
            small = torch.full(((len(want) - 2) * rb + 64,), 0xEE, dtype=torch.uint8, device="cuda")
            assert lib.gf_pools_merge_dev(h, _dev(recv).data_ptr(), cap, _dev(all_cnt).data_ptr(), n_lib, world, n_gaps, L, me, world, batch,
                                          small.data_ptr(), len(want) - 2, d_moff.data_ptr(), d_err.data_ptr()) == 0
            gf.sync()
            assert int(d_err[0]) & 0x80000000 and (small.cpu().numpy()[(len(want) - 2) * rb:] == 0xEE).all()
    assert total == sum(len(rows) for r in range(world) for rows, _ in pools[r])


def test_pack_flags_a_send_slot_that_is_too_small():
    import torch
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    lib = B.lib()
    gf = GapFill(0)
    rng = np.random.default_rng(5)
    rows, off = _random_pools(rng, 64, 38, 30)
    d_send = torch.full((2 * 8 * 38 + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros(64, dtype=torch.int32, device="cuda")
    d_err = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert lib.gf_pools_pack_for_owners_dev(gf.handle, _dev(rows.reshape(-1)).data_ptr(), _dev(off.view(np.int64)).data_ptr(), 64, 150, 2, 4, 0, 1,
                                            d_send.data_ptr(), 8, d_cnt.data_ptr(), d_err.data_ptr()) == 0
    gf.sync()
    assert int(d_err[0]) & 0x40000000 and (d_send.cpu().numpy()[2 * 8 * 38:] == 0xEE).all()


def _rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))


def decode_best(b):
    """gap_best word -> (anchor length, span + 1, contig index, reverse?)."""
    b = int(b)
    return b >> 56, (b >> 32) & 0xFFFFFF, 0x7FFFFFFF - ((b >> 1) & 0x7FFFFFFF), b & 1


def oracle_pick(gid, contigs, left, right, scores=(30, 15)):
    """The pipeline's pick of one gap — the first score that picks something (assemble_gaps.py:336-366) — by the oracle:
    (score, contig index, gap sequence, reverse?) or None.  contigs = [(name, seq)] with unique names."""
    from oracle import gp_oracle as O
    for a in scores:
        seqs, ctgs = O.pick_gap(gid, contigs, left, right, a)
        if seqs:
            hdr, body = seqs.split("\n")[:2]
            name = hdr[len(">" + gid + "_"):]
            ci = [n for n, _ in contigs].index(name)
            written = ctgs.split("\n")[1]
            rev = written != contigs[ci][1]
            assert rev or written == contigs[ci][1]
            return a, ci, body, rev
    return None


def test_device_flank_anchoring_equals_the_oracle_picker():
    """gf_pick_anchored_dev vs oracle/gp_oracle.py::pick_gap (the reference's selection, pick_contigs.py:97-358, pinned on its own
    answers, on the exact-anchor stand-in's hits): contig, strand and span of every gap — both orientations, repeated anchors,
    anchors in the wrong order / overlapping / back to back, forward and reverse pair in one contig, flanks shorter than or exactly
    as long as the anchor, non-ACGT in an anchor; the pick at 30 outranks any pick at 15."""
    import torch
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    import pick_util as PK
    cases = PK.picker_cases(11, 300)
    n_gaps = len(cases)
    flanks = [(l, r) for l, r, _ in cases]
    gaps = np.zeros(n_gaps, dtype=B.GAP)
    for g in range(n_gaps):
        gaps[g] = (0, 1000 * (g + 1), 1000 * (g + 1) + 100, g + 1)
    rng = np.random.default_rng(12)
    contigs = [(g, s) for g, (_, _, seqs) in enumerate(cases) for s in seqs]
    order = rng.permutation(len(contigs))
    contigs = [contigs[i] for i in order]
    ctg = np.zeros(len(contigs), dtype=B.CONTIG)
    seq = "".join(s for _, s in contigs)
    o = 0
    for i, (g, s) in enumerate(contigs):
        ctg[i] = (g, 31, 29, max(1, len(s) - 28), len(s), 0, 0, o)
        o += len(s)
    gf = GapFill(0)
    gf.set_gaps(gaps, 1, flanks)
    lib = B.lib()
    d_ctg, d_seq = _dev(ctg.view(np.uint8)), _dev(np.frombuffer(seq.encode(), dtype=np.uint8))
    d_n = torch.tensor([len(contigs)], dtype=torch.int32, device="cuda")
    for scores in ((30, 15), (15, 30), "one pass"):         # the order of the calls does not matter: the longer anchor outranks
        d_best = torch.zeros(n_gaps, dtype=torch.int64, device="cuda")
        d_closed = torch.zeros(1, dtype=torch.int32, device="cuda")
        if scores == "one pass":                            # both scores in one launch (what bench.py's step runs)
            assert lib.gf_pick_anchored2_dev(gf.handle, d_ctg.data_ptr(), d_n.data_ptr(), len(contigs), d_seq.data_ptr(), 30, 15, d_best.data_ptr(),
                                             d_closed.data_ptr()) == 0
        else:
            for a in scores:
                assert lib.gf_pick_anchored_dev(gf.handle, d_ctg.data_ptr(), d_n.data_ptr(), len(contigs), d_seq.data_ptr(), a, d_best.data_ptr(),
                                                d_closed.data_ptr()) == 0
        gf.sync()
        best = d_best.cpu().numpy().view(np.uint64)
        n_closed = n_rev = n_15 = 0
        for g in range(n_gaps):
            mine = [("c%d" % i, s) for i, (gg, s) in enumerate(contigs) if gg == g]
            idx = [i for i, (gg, _) in enumerate(contigs) if gg == g]
            want = oracle_pick("0_%d" % (g + 1), mine, flanks[g][0], flanks[g][1])
            if want is None:
                assert int(best[g]) == 0, g
                continue
            a, ci, body, rev = want
            assert decode_best(best[g]) == (a, len(body), idx[ci], int(rev)), (g, decode_best(best[g]), want)
            n_closed += 1
            n_rev += rev
            n_15 += a == 15
        assert int(d_closed[0]) == n_closed and n_closed > 100 and n_rev > 20 and n_15 > 20 and n_closed < n_gaps - 20


def test_multi_k_call_equals_one_call_per_pair():
    import torch
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    import synth_small as S
    case = S.small_case(seed=3, n_pairs=6000)
    L = case["L"]
    gf = GapFill(0)
    lib = B.lib()
    packed, _ = GapFill.pack_reads(case["reads_blob"], L)
    n = packed.shape[0]
    off = np.array([0, n // 3, n // 3, n], dtype=np.uint64)       # three pools, the middle one empty
    kk = [(31, 29), (41, 39), (51, 49)]
    single = []
    for k, kv in kk:
        c, s = gf.assemble(packed, off, L, [(k, kv)])
        single += sorted((int(x["gap"]), k, kv, s[int(x["seq_off"]):int(x["seq_off"]) + int(x["length"])].decode(), int(x["n_nodes"]), int(x["cov_sum"])) for x in c)
    d_pool, d_off = _dev(packed.reshape(-1)), _dev(off.view(np.int64))
    ccap, scap = 1 << 16, 1 << 24
    d_ctg = torch.zeros(ccap * 32, dtype=torch.uint8, device="cuda")
    d_seq = torch.zeros(scap, dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros(8, dtype=torch.int32, device="cuda")
    d_err = torch.zeros(3, dtype=torch.int32, device="cuda")
    ks, kvs = (C.c_int * 3)(*[a for a, _ in kk]), (C.c_int * 3)(*[b for _, b in kk])
    assert lib.gf_assemble_multi_dev(gf.handle, d_pool.data_ptr(), None, d_off.data_ptr(), 3, n, L, ks, kvs, 3, 2, 40, d_ctg.data_ptr(), ccap,
                                     d_cnt.data_ptr(), d_seq.data_ptr(), scap, d_cnt.data_ptr() + 8, d_err.data_ptr()) == 0
    gf.sync()
    cnt = d_cnt.cpu().numpy()
    nc, ns = int(cnt[0]), int(cnt[2:4].view(np.uint64)[0])
    assert int(d_err.sum()) == 0 and nc <= ccap and ns <= scap
    c = np.frombuffer(d_ctg[:nc * 32].cpu().numpy().tobytes(), dtype=B.CONTIG)
    s = d_seq[:ns].cpu().numpy().tobytes()
    multi = sorted((int(x["gap"]), int(x["k"]), int(x["kv"]), s[int(x["seq_off"]):int(x["seq_off"]) + int(x["length"])].decode(), int(x["n_nodes"]),
                    int(x["cov_sum"])) for x in c)
    assert multi == sorted(single) and len(single) > 10 and {x[1] for x in multi} == {31, 41, 51}


def test_assembly_refuses_pool_offsets_beyond_the_pool_array():
    """an overflowed pool_off must not make the kernel touch memory outside its workspace."""
    import torch
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    import synth_small as S
    case = S.small_case(seed=4, n_pairs=2000)
    L = case["L"]
    gf = GapFill(0)
    lib = B.lib()
    packed, _ = GapFill.pack_reads(case["reads_blob"], L)
    n = 600
    off = np.array([0, 300, 600, 5000], dtype=np.uint64)          # third pool claims rows beyond total_reads = 600
    d_pool, d_off = _dev(packed[:n].reshape(-1)), _dev(off.view(np.int64))
    d_ctg = torch.zeros(4096 * 32, dtype=torch.uint8, device="cuda")
    d_seq = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros(8, dtype=torch.int32, device="cuda")
    d_err = torch.zeros(3, dtype=torch.int32, device="cuda")
    assert lib.gf_assemble_dev(gf.handle, d_pool.data_ptr(), None, d_off.data_ptr(), 3, n, L, 31, 29, 2, 40, d_ctg.data_ptr(), 4096, d_cnt.data_ptr(),
                               d_seq.data_ptr(), 1 << 20, d_cnt.data_ptr() + 8, d_err.data_ptr()) == 0
    gf.sync()
    e = d_err.cpu().numpy()
    assert e[0] == 0 and e[1] == 0 and e[2] != 0


def test_pool_builder_flags_a_pool_buffer_that_is_too_small():
    import torch
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    import synth_small as S
    case = S.small_case(seed=9, n_pairs=3000)
    L = case["L"]
    gf = GapFill(0)
    gf.set_gaps(case["gaps"], case["n_scaffolds"], case["flanks"])
    lib = B.lib()
    packed, _ = GapFill.pack_reads(case["reads_blob"], L)
    hits = gf.screen_reads(packed, L, 31)
    keys = (hits["gap"].astype(np.uint64) << np.uint64(32)) | hits["read"].astype(np.uint64)
    d_reads, d_keys = _dev(packed.reshape(-1)), _dev(keys.view(np.int64))
    d_nk = torch.tensor([len(keys)], dtype=torch.int32, device="cuda")
    d_off = torch.zeros(len(case["gaps"]) + 1, dtype=torch.int64, device="cuda")
    d_err = torch.zeros(1, dtype=torch.int32, device="cuda")
    for cap, flagged in ((len(keys) + 8, False), (max(1, len(keys) // 2), True)):
        d_pool = torch.full((cap * 38 + 64,), 0xEE, dtype=torch.uint8, device="cuda")
        assert lib.gf_build_pools_dev(gf.handle, d_reads.data_ptr(), packed.shape[0], L, d_keys.data_ptr(), d_nk.data_ptr(), len(keys),
                                      d_pool.data_ptr(), cap, d_off.data_ptr(), None, d_err.data_ptr()) == 0
        gf.sync()
        assert bool(int(d_err[0]) & 0x80000000) == flagged and int(d_off[-1]) == len(set(keys.tolist()))
        assert (d_pool.cpu().numpy()[cap * 38:] == 0xEE).all()


def _bench(argv, env_extra=None, timeout=1500, nproc=1):
    env = dict(os.environ)
    env.update(env_extra or {})
    if nproc == 1:
        env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + argv
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", "29611", os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + argv
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, env=env, cwd=ROOT)
    if r.returncode != 0 and nproc > 1:      # a multi-process launch can lose its rendezvous (port still in TIME_WAIT, a slow peer): once more, loudly
        sys.stderr.write("multi-rank launch failed once, retrying:\n" + r.stderr.decode()[-1500:] + "\n")
        import time
        time.sleep(3)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    return json.loads([l for l in r.stdout.decode().strip().splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("config,extra", [("C2", ["--reads", "6000000"]), ("C5", ["--reads", "4000000", "--mp-reads", "2000000"])])
def test_two_rank_owner_assembly_equals_the_single_process_run(tmp_path, config, extra):
    """SURVEY.md §8e / north_star: the HIP kernels on two ranks (sharing this box's one GPU, gloo for the collectives), reads split,
    pools exchanged to the gap owners — the gathered contigs and the closed count equal the single-process run over all reads."""
    one, two = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    a = _bench(["--config", config, "--steps", "1", "--warmup", "0", "--no-cpu", "--no-extras", "--dump-contigs", one] + extra)
    b = _bench(["--config", config, "--steps", "1", "--warmup", "0", "--no-cpu", "--no-extras", "--dump-contigs", two] + extra,
               env_extra={"GF_BENCH_BACKEND": "gloo", "GF_BENCH_ONE_GPU": "1"}, nproc=2)
    ja, jb = json.load(open(one)), json.load(open(two))
    assert ja["contigs"] == jb["contigs"] and len(ja["contigs"]) > 100
    assert ja["gaps_closed"] == jb["gaps_closed"]
    assert a["counts"]["gaps_closed_correct"] == b["counts"]["gaps_closed_correct"] <= a["counts"]["gaps_closed"]   # every closed gap of both runs went through the truth check
    assert a["counts"]["assembled_pool_reads"] == b["counts"]["assembled_pool_reads"]
    assert b["ranks"] == 2 and b["n_gpus"] == 1 and b["functional_mode"] and b["scaling"] is None      # two ranks on ONE GPU are not a scaling point
    assert b["fixed_ms"]["second_hop_union"] > 0 and b["fixed_ms"]["owner_exchange"] > 0
    assert abs(a["gaps_per_s"] * a["ms_per_step"] - b["gaps_per_s"] * b["ms_per_step"]) < 1e-3 * a["gaps_per_s"] * a["ms_per_step"]   # same gaps, counted once


def _n_gpus():
    import torch
    return torch.cuda.device_count()       # (counting devices does not initialise the GPU)


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs: one RCCL rank per GPU (the pool's boxes have one; the test runs the day a box has two)")
@pytest.mark.parametrize("config,extra", [("C2", ["--reads", "6000000"]), ("C5", ["--reads", "4000000", "--mp-reads", "2000000"])])
def test_two_rank_rccl_equals_single_process(tmp_path, config, extra):
    """The real multi-GPU path: two ranks, ONE GPU EACH, backend `nccl` (= RCCL over xGMI) — the second hop's packed all-gather, the
    exact-size all-to-all of the owner exchange, the all-reduces of the results and the final gather on device tensors.  Same contigs,
    pools and closed gaps as the single-process run over all reads (one owner per gap, assemble_gaps.py:296-299)."""
    one, two = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    argv = ["--config", config, "--steps", "2", "--warmup", "1", "--no-cpu", "--no-extras"] + extra
    a = _bench(argv + ["--dump-contigs", one])
    b = _bench(argv + ["--dump-contigs", two], env_extra={"GF_BENCH_BACKEND": "nccl", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, nproc=2)
    ja, jb = json.load(open(one)), json.load(open(two))
    assert ja["contigs"] == jb["contigs"] and len(ja["contigs"]) > 100 and ja["gaps_closed"] == jb["gaps_closed"]
    assert a["counts"]["assembled_pool_reads"] == b["counts"]["assembled_pool_reads"]
    assert a["counts"]["gaps_closed_correct"] == b["counts"]["gaps_closed_correct"]
    assert b["ranks"] == 2 and b["n_gpus"] == 2 and not b["functional_mode"] and b["scaling"] == "strong" and "RCCL" in b["config"]["collectives"]
    assert b["fixed_ms"]["second_hop_union"] > 0 and b["fixed_ms"]["owner_exchange"] > 0 and b["final_gather_ms"] >= 0


def test_exact_size_exchange_equals_the_padded_slots(tmp_path):
    """The owner exchange with exact split sizes and the counts inside the one all-to-all (default) against the equal-slot form with
    its all-gather of counts (GF_XCHG=slots), two ranks on this box's GPU over gloo: same contigs; the exact form sends fewer bytes."""
    one, two = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    argv = ["--config", "C5", "--reads", "4000000", "--mp-reads", "2000000", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-extras"]
    env = {"GF_BENCH_BACKEND": "gloo", "GF_BENCH_ONE_GPU": "1"}
    a = _bench(argv + ["--dump-contigs", one], env_extra=env, nproc=2)
    b = _bench(argv + ["--dump-contigs", two], env_extra=dict(env, GF_XCHG="slots"), nproc=2)
    assert json.load(open(one))["contigs"] == json.load(open(two))["contigs"] and a["counts"] == b["counts"]
    xa, xb = a["exchange"], b["exchange"]
    assert xa["form"].startswith("exact") and xb["form"].startswith("equal")
    assert 0 < xa["bytes_sent_per_rank_and_step"] < xb["bytes_sent_per_rank_and_step"] and xa["collectives_per_step"] < xb["collectives_per_step"]


@pytest.mark.parametrize("config,extra", [("C2", ["--reads", "6000000"]), ("C5", ["--reads", "4000000", "--mp-reads", "2000000"])])
def test_rccl_branch_runs_at_world_one_and_changes_nothing(tmp_path, config, extra):
    """A one-GPU box cannot run two RCCL ranks, but it can run ONE: GF_BENCH_FORCE_EXCHANGE=1 takes the multi-rank code path — process
    group on the `nccl` backend (= RCCL), side stream, all-gather of the second-hop rows + their merge, OwnerExchange (device pack,
    all_gather_into_tensor of the counts, all_to_all_single of the slots, device merge), all-reduces of the results, final gather —
    with world size 1, on device tensors.  Same contigs, pools and closed gaps as the plain single-process run."""
    one, two = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    argv = ["--config", config, "--steps", "2", "--warmup", "0", "--no-cpu", "--no-extras"] + extra
    a = _bench(argv + ["--dump-contigs", one])
    b = _bench(argv + ["--dump-contigs", two], env_extra={"GF_BENCH_FORCE_EXCHANGE": "1", "GF_BENCH_BACKEND": "nccl", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29633"})
    ja, jb = json.load(open(one)), json.load(open(two))
    assert ja["contigs"] == jb["contigs"] and len(ja["contigs"]) > 100 and ja["gaps_closed"] == jb["gaps_closed"]
    assert a["counts"] == b["counts"]
    assert "RCCL" in b["config"]["collectives"] and b["config"]["forced_exchange_at_world_1"] and b["n_gpus"] == 1
    assert b["fixed_ms"]["second_hop_union"] > 0 and b["fixed_ms"]["owner_exchange"] > 0 and b["final_gather_ms"] >= 0


@pytest.mark.parametrize("env", [{"GF_BENCH_TAG_AHEAD": "1"}, {"GF_BENCH_TWO_STREAMS": "1"}, {"GF_BENCH_TAG_KEYS": "0"}])
def test_stream_and_layout_variants_of_the_step_change_nothing(tmp_path, env):
    """The step's options that move work between streams (Pipeline.tag_ahead: the next step's tagger beside this step's assembly; the tagger
    on a second stream beside the filter) or change what the tagger streams (the 32-byte records instead of their key column) leave contigs,
    counts and closed gaps as they are — three steps in a row, so that the pipelined tagger's hand-over between steps is exercised."""
    one, two = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    argv = ["--config", "C5", "--reads", "4000000", "--mp-reads", "2000000", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-extras"]
    a = _bench(argv + ["--dump-contigs", one])
    b = _bench(argv + ["--dump-contigs", two], env_extra=env)
    ja, jb = json.load(open(one)), json.load(open(two))
    assert ja["contigs"] == jb["contigs"] and len(ja["contigs"]) > 100 and ja["gaps_closed"] == jb["gaps_closed"]
    assert a["counts"] == b["counts"]


def test_bench_launches_its_own_ranks_from_a_bare_shell(tmp_path):
    """`python bench.py --gpus 2` without torch.distributed.run around it (no WORLD_SIZE in the environment): bench.py starts the
    ranks as a child process; on this one-GPU box they share cuda:0 and talk over gloo.  Same contigs as the single-process run."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GF_BENCH_BACKEND", "GF_BENCH_ONE_GPU")}
    one, two = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    argv = ["--config", "C2", "--reads", "4000000", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-extras"]
    a = _bench(argv + ["--dump-contigs", one])
    for attempt in range(2):      # (a lost rendezvous of the child launcher: once more, loudly)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + argv + ["--dump-contigs", two], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=900, env=env, cwd=ROOT)
        if r.returncode == 0:
            break
        sys.stderr.write("bench.py --gpus 2 failed (attempt %d):\n%s\n" % (attempt + 1, r.stderr.decode()[-1500:]))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    b = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert b["ranks"] == 2 and b["n_gpus"] == 1 and b["functional_mode"] and "gloo" in b["config"]["collectives"] and "ranks share cuda:0" in r.stderr.decode()
    assert json.load(open(one))["contigs"] == json.load(open(two))["contigs"]
    assert a["counts"]["assembled_pool_reads"] == b["counts"]["assembled_pool_reads"]


@pytest.mark.parametrize("config", ["C2", "C3", "C4", "C5"])
def test_full_size_config_sample_parity(config):
    """BASELINE.json configs[1], [2], [3], [4] at FULL size on this GPU (reads generated on the device): all screen and tagger hits of
    the oracle's sample (every library) — STRIPES over the whole library: its first and last reads, the reads whose bytes straddle
    offset 4 GiB of the packed reads / the records / the key column, the shard boundaries of 2 / 4 / 8 ranks, seeded places —, the
    merged pools' assembly of 256 gaps drawn over the whole gap list (first and last included) at every (k, kv) and the closed flags
    equal the oracle / the host picker; every gap recruits reads and yields contigs."""
    d = _bench(["--config", config, "--steps", "1", "--warmup", "0", "--no-extras"])
    cb = d["cpu_baseline"]
    assert cb["parity_recruit"] and cb["parity_assembly"] and cb["parity_pick"] and cb["parity_on_sample"], cb
    assert cb["sample_hits"] > 1000 and cb["sample_contigs"] > 100
    n_gaps = d["config"]["gaps"]
    assert cb["parity_sample"] == "striped"
    rg = cb["parity_sample_ranges"]
    assert rg["gaps"]["first"] == 0 and rg["gaps"]["last"] == n_gaps - 1 and rg["gaps"]["n"] >= min(n_gaps, 250)
    assert rg["gaps"]["scaffolds_touched"] >= min(200, {"C2": 50, "C3": 1, "C4": 620, "C5": 620}[config])      # (256 gaps drawn over 620 scaffolds: ~247 distinct ones)
    for name, w in rg.items():
        if name == "gaps":
            continue
        r = w["read_ranges"]
        assert r[0][0] == 0 and r[-1][1] == w["of"] and (w["complete"] or w["stripes"] >= 8), (name, w)       # from the first read to the last
        for stride in (38, 32, 8):          # a library whose arrays pass 4 GiB: the read / record at that byte offset is in the sample
            at = (1 << 32) // stride
            assert w["of"] * stride <= (1 << 32) or any(a <= at < b for a, b in r), (name, stride)
    assert d["counts"]["gaps_with_contig"] == n_gaps
    want_reads = {"C2": 50_000_000, "C3": 5_000_000, "C4": 900_000_000, "C5": 1_300_000_000}[config]     # C5: 900 M + the 400 M mate-pair records that make 2-kb gaps closable (bench.py)
    assert d["config"]["reads_total"] == want_reads and d["n_gpus"] == 1
    if config == "C5":      # the mate-pair geometry closes the gaps in one pass (tip clipping + bubble popping on)
        assert d["counts"]["gaps_closed"] > 0.9 * n_gaps and d["gaps_closed_per_s"] > 0
        # ... and closes them with the TRUE sequence (bench.py::truth_check compares every picked sequence with the generator's genome):
        # a tie-break that lets error alleles win shows up here (round 2's build: 53 %)
        tc = d["closed_truth_check"]
        assert tc["closed"] == d["counts"]["gaps_closed"] and d["counts"]["gaps_closed_correct"] >= 0.999 * tc["closed"], tc
        # the gaps the step's first pick leaves open go through the contig merger ON THE DEVICE, inside the step, and a second pick
        # (Pipeline.assemble; the reference merges before it picks, assemble_gaps.py:301-306, 335-339): most of them close, nearly all with
        # the true sequence; the merged contigs of a sample of those gaps equal the oracle's merger (cpu_baseline.parity_merge_round)
        mr = d["contig_merge_round"]
        assert mr["inside_the_timed_step"] and mr["gaps_tried"] <= n_gaps - mr["gaps_closed_without_merging"], mr
        assert mr["gaps_closed_by_merging"] >= 0.5 * mr["gaps_tried"] > 0 and mr["closed_correct"] >= 0.9 * mr["gaps_closed_by_merging"], mr
        assert d["counts"]["gaps_closed"] == mr["gaps_closed_without_merging"] + mr["gaps_closed_by_merging"] >= 0.998 * n_gaps, mr
        assert cb["parity_merge_round"] is True and cb["merge_round_gaps_checked"] >= 16, cb
    else:
        # no library spans these gaps: the step is timed without the merge round, which runs once behind the timed region over ALL open gaps;
        # a seeded sample of its merged contigs equals the oracle's merger
        ma = d["contig_merge_round_all_gaps"]
        assert "error" not in ma and not ma["inside_the_timed_step"] and ma["gaps_tried"] > 0.9 * n_gaps and ma["new_contigs"] > 0, ma
        assert ma["parity"]["gaps_checked"] >= 16 and ma["parity"]["merged_contigs_equal_the_oracles"], ma
        assert d["counts"]["gaps_closed_correct"] == d["counts"]["gaps_closed"] or d["counts"]["gaps_closed"] == 0 or \
            d["counts"]["gaps_closed_correct"] >= 0.99 * d["counts"]["gaps_closed"], d["closed_truth_check"]


def test_c2_full_size_complete_parity():
    """BASELINE.json configs[1] ("bit-exact read-ID check") with NOTHING sampled: every one of the 50 M reads and records through the
    oracle's k-mer screen and alignment tagger, every one of the 1 000 gaps' pools through the oracle's assembly and the host picker —
    hit lists, contigs and pick words of the GPU step equal them (collect_reads_for_gaps.py:68-263, collect_discordant_low_mapq_reads.py:31-84
    for the recruit rules)."""
    d = _bench(["--config", "C2", "--steps", "1", "--warmup", "0", "--no-extras", "--cpu-sample-reads", "-1", "--cpu-sample-gaps", "-1"], timeout=2400)
    cb = d["cpu_baseline"]
    assert cb["parity_sample"] == "complete", cb["parity_sample"]
    w = cb["parity_sample_ranges"]
    assert w["short-insert"]["complete"] and w["short-insert"]["reads"] == 50_000_000 and w["gaps"]["n"] == w["gaps"]["of"] == 1000
    assert cb["parity_recruit"] and cb["parity_assembly"] and cb["parity_pick"] and cb["parity_on_sample"], cb
    assert cb["sample_hits"] == d["counts"]["libraries"]["short-insert"]["screen_hits"] > 100_000
