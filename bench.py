#!/usr/bin/env python3
"""bench.py — reads screened/s (+ gaps/s) of the recruit + local-assembly hot path on MI355X.

A step = one pass of the hot path over one batch of the seeded synthetic workload (include/gf_synth.h),
inputs already resident in HBM when the timed region starts.  Workload at every N: BASELINE.json configs[1]
("C2": 1 000 gaps x 2 kb, 50 M 2x150-bp read records, k=31) PER GPU — gaps are replicated, reads sharded
(rank r owns pairs [r*P, (r+1)*P)), no data-path collective; scaling = weak.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel = the screen
filter; algorithmic bytes = ceil(2L/8) per read) and `cpu_baseline` (the oracle's C restatement —
kind "port" — timed on the host cores on a bounded sample of the same workload; that sample is also checked
bit-for-bit against the GPU's hits).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C2", choices=["C2", "C3", "C4"],
                    help="BASELINE.json workload: C2 (default, the metric's configuration), C3 (E. coli scale, k=41), "
                         "C4 (human scale, k=51; --reads is the PER-GPU shard of the 900 M reads)")
    ap.add_argument("--reads", type=int, default=0, help="read records per GPU (default: C2 50 M, C3 5 M, C4 112.5 M)")
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--cpu-sample-reads", type=int, default=4_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-human-scale", action="store_true",
                    help="skip the extra C4-shard measurement that the default (N=1, C2) run appends as `human_scale_shard`")
    ap.add_argument("--exchange", action="store_true",
                    help="N > 1: all-to-all-v of the per-gap pools to one owner rank per gap before the assembly (off: every rank "
                         "assembles the gaps from its own shard of the reads; the only collective is the final gather)")
    args = ap.parse_args()
    args.reads_given = bool(args.reads)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # GF_BENCH_BACKEND=gloo + GF_BENCH_ONE_GPU=1: smoke-test of the multi-rank code path with every rank on cuda:0
    # (single-GPU boxes); the real runs use nccl (= RCCL) with one GPU per rank
    backend = os.environ.get("GF_BENCH_BACKEND", "nccl")
    if os.environ.get("GF_BENCH_ONE_GPU"):
        local = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    from gappadder_amd import _lib as B
    from gappadder_amd import sharding as SH
    from gappadder_amd.hip_api import GapFill

    presets = {   # (seed, scaffold_len, n_scaffolds, gaps_per_scaffold, gap_len, reads per GPU, k)
        "C2": (20260002, 5_000_000, 50, 20, 2000, 50_000_000, 31),
        "C3": (20260003, 4_600_000, 1, 200, 1000, 5_000_000, 41),
        "C4": (20260004, 5_000_000, 620, 32, 2000, 112_500_000, 51),
    }
    seed, slen, nscf, gps, glen, dreads, dk = presets[args.config]
    args.reads = args.reads or dreads
    L, k = 150, (args.k or dk)
    n_pairs = args.reads // 2
    n_reads = 2 * n_pairs
    cfg = GapFill.synth_cfg(seed=seed, scaffold_len=slen, n_scaffolds=nscf, gaps_per_scaffold=gps, gap_len=glen,
                            read_len=L, insert_mean=300, insert_sd=30)
    gaps, flanks = GapFill.synth_layout(cfg)
    gf = GapFill(local)
    gf.set_gaps(gaps, int(cfg["n_scaffolds"][0]), flanks)
    # second context = second HIP stream on the same device: the alignment tagger + second hop are independent of the k-mer
    # screen until the pools are built, so they run beside the screen's verify pass
    if os.environ.get("GF_VERIFY_BATCH"):
        gf.set_option("screen_verify_batch", int(os.environ["GF_VERIFY_BATCH"]))
    if os.environ.get("GF_BENCH_SERIAL"):     # diagnostic: tagger on the same stream, so every phase time is stand-alone
        gf2 = gf
    else:
        gf2 = GapFill(local)
        gf2.set_gaps(gaps, int(cfg["n_scaffolds"][0]), None)
    lib = B.lib()
    rb = lib.gf_packed_read_bytes(L)

    # ---- inputs resident in HBM (torch = device-memory plumbing) ----
    d_reads = torch.empty(n_reads * rb + 64, dtype=torch.uint8, device=dev)
    d_recs = torch.empty(n_reads * 32, dtype=torch.uint8, device=dev)
    first_pair = rank * n_pairs
    gf.synth_pairs_dev(cfg, first_pair, n_pairs, d_reads.data_ptr(), d_recs.data_ptr())
    hit_cap = max(1 << 20, n_reads // 8)
    d_hits = torch.empty(hit_cap * 8, dtype=torch.uint8, device=dev)
    d_thits = torch.empty(hit_cap * 12, dtype=torch.uint8, device=dev)
    d_lhits = torch.empty(hit_cap * 12, dtype=torch.uint8, device=dev)
    low_cap = max(1 << 20, n_reads // 8)          # MAPQ==0 records compacted by the tagger pass (2 % of the records here)
    d_low = torch.empty(low_cap * 12, dtype=torch.uint8, device=dev)
    key_cap = 4 * hit_cap
    d_keys = torch.empty(key_cap, dtype=torch.int64, device=dev)
    pool_cap = max(1 << 20, n_reads // 32)    # pooled reads (also sizes the assembly workspace: ~6.5 KB per pooled read)
    d_pool = torch.empty(pool_cap * rb + 64, dtype=torch.uint8, device=dev)
    d_pool_off = torch.zeros(len(gaps) + 1, dtype=torch.int64, device=dev)
    d_pool_ids = torch.empty(pool_cap, dtype=torch.int32, device=dev)
    contig_cap, seq_cap = 256 * len(gaps) + 1024, 32768 * len(gaps) + (1 << 20)
    d_ctg = torch.empty(contig_cap * 32, dtype=torch.uint8, device=dev)
    d_seq = torch.empty(seq_cap, dtype=torch.uint8, device=dev)
    d_gap_err = torch.zeros(len(gaps), dtype=torch.int32, device=dev)
    # counters (device u32 unless noted): 0 screen hits, 4 tagger hits, 8 second-hop hits, 12 keys, 16 contigs,
    # 20 (u64) contig bases, 24 pool-sort overflow, 28 MAPQ==0 records, 29 second-hop table rows
    d_cnt = torch.zeros(32, dtype=torch.int32, device=dev)
    cp = d_cnt.data_ptr()
    gf.sync()
    kv = k - 2
    h = gf.handle

    h2 = gf2.handle

    def recruit():
        assert lib.gf_stream_wait(h2, h) == 0          # the previous step's consumers of the tagger buffers are done
        rc = lib.gf_screen_reads_dev(h, d_reads.data_ptr(), None, n_reads, L, k, 1, d_hits.data_ptr(), hit_cap, cp)
        assert rc == 0, rc
        rc = lib.gf_tag_alignments_low_dev(h2, d_recs.data_ptr(), n_reads, 300, 30, 250, 30, d_thits.data_ptr(), hit_cap, cp + 16,
                                           d_low.data_ptr(), low_cap, cp + 112)
        assert rc == 0, rc

    # second-hop table (run_multi_threads_discordant.py:19-122 inverts the discordant lines and runs sort(1) on the host): built
    # on the device from the tagger's hits INSIDE every step (gf_second_hop_table_dev); this untimed pass only sizes its buffers
    recruit()
    gf.sync()
    gf2.sync()
    n_th = int(d_cnt[4])
    th = np.frombuffer(d_thits[:n_th * 12].cpu().numpy().tobytes(), dtype=B.TAGHIT)
    row_cap = 2 * int((th["kind"] == B.KIND_DISCORDANT).sum()) + 4096
    d_rows = torch.empty(row_cap * 16, dtype=torch.uint8, device=dev)
    d_row_gap = torch.empty(row_cap, dtype=torch.int32, device=dev)

    exch_rows = [0]

    def step():
        recruit()
        rc = lib.gf_second_hop_table_dev(h2, d_recs.data_ptr(), d_thits.data_ptr(), cp + 16, hit_cap, d_rows.data_ptr(), d_row_gap.data_ptr(),
                                         row_cap, cp + 116)
        assert rc == 0, rc
        rc = lib.gf_tag_low_mapq_table_dev(h2, d_low.data_ptr(), cp + 112, low_cap, d_rows.data_ptr(), cp + 116, row_cap, d_lhits.data_ptr(),
                                           hit_cap, cp + 32)
        assert rc == 0, rc
        assert lib.gf_stream_wait(h, h2) == 0          # pools need the tagger's and the second hop's hits
        assert lib.gf_pool_keys_all_dev(h, d_hits.data_ptr(), cp, hit_cap, 1, d_recs.data_ptr(), d_thits.data_ptr(), cp + 16, hit_cap,
                                        d_lhits.data_ptr(), cp + 32, hit_cap, d_row_gap.data_ptr(), d_keys.data_ptr(), key_cap, cp + 48) == 0
        assert lib.gf_build_pools_dev(h, d_reads.data_ptr(), n_reads, L, d_keys.data_ptr(), cp + 48, key_cap, d_pool.data_ptr(),
                                      pool_cap, d_pool_off.data_ptr(), d_pool_ids.data_ptr(), cp + 96) == 0
        pool_ptr, off_ptr, pool_rows = d_pool.data_ptr(), d_pool_off.data_ptr(), pool_cap
        if args.exchange and world > 1:
            # optional exchange step: every gap gets ONE owner that holds the recruits of all ranks, so the
            # assembled gaps equal a single-process run over all reads; costs a host sync (row counts) + one all-to-all-v
            gf.sync()
            n_rows = int(d_pool_off[-1])
            merged, moff = SH.exchange_pools(d_pool[:n_rows * rb].view(n_rows, rb), d_pool_off, coll_device=coll_dev)
            torch.cuda.synchronize()
            step.keep = (merged, moff)     # alive until the assembly kernel has run
            exch_rows[0] = int(merged.shape[0])
            pool_ptr, off_ptr, pool_rows = merged.data_ptr(), moff.data_ptr(), max(1, int(merged.shape[0]))
        rc = lib.gf_assemble_dev(h, pool_ptr, None, off_ptr, len(gaps), pool_rows, L, k, kv, 2, 40,
                                 d_ctg.data_ptr(), contig_cap, cp + 64, d_seq.data_ptr(), seq_cap, cp + 80, d_gap_err.data_ptr())
        assert rc == 0, rc

    def barrier():
        gf.sync()
        gf2.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    gf.timing(True)
    gf2.timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    kt = {name: g_.kernel_time(idx) for name, idx, g_ in (("screen_filter", B.KERNEL_SCREEN, gf), ("screen_verify", B.KERNEL_VERIFY, gf),
                                                          ("tag_alignments", B.KERNEL_TAG, gf2), ("tag_low_mapq", B.KERNEL_LOWMAPQ, gf2),
                                                          ("pools", B.KERNEL_POOL, gf), ("assemble", B.KERNEL_ASSEMBLE, gf))}
    t_filter, n_filter = kt["screen_filter"]
    gf.timing(False)
    gf2.timing(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    cnt = d_cnt.cpu().numpy()
    n_hits, n_thits, n_lhits, n_keys, n_ctg = int(cnt[0]), int(cnt[4]), int(cnt[8]), int(cnt[12]), int(cnt[16])
    n_seq = int(cnt[20:22].view(np.uint64)[0])
    pool_off = d_pool_off.cpu().numpy()
    assert int(cnt[28]) <= low_cap and int(cnt[29]) <= row_cap
    assert int(cnt[24]) == 0 and int(d_gap_err.sum()) == 0 and pool_off[-1] <= pool_cap and n_ctg <= contig_cap and n_seq <= seq_cap
    ctg = np.frombuffer(d_ctg[:n_ctg * 32].cpu().numpy().tobytes(), dtype=B.CONTIG)
    if world > 1:
        # the only collective of the path: gather the assembled sequences of every rank's shard on rank 0 (RCCL)
        seq_local = d_seq[:n_seq].cpu().numpy().tobytes()
        payload = SH.encode_contigs([(int(c["gap"]), int(c["k"]), int(c["kv"]), int(c["n_nodes"]), int(c["cov_sum"]),
                                      seq_local[int(c["seq_off"]):int(c["seq_off"]) + int(c["length"])].decode()) for c in ctg])
        gathered = SH.gather_bytes(payload, dst=0, device=coll_dev)
        if rank == 0:
            assert len(gathered) == world and all(len(g) > 0 for g in gathered)

    out = None
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        filt_ms = t_filter / max(1, n_filter)
        achieved = n_reads * rb / (filt_ms * 1e-3) / 1e9
        phases = {name: t / max(1, n_filter) for name, (t, _) in kt.items()}
        gaps_with_contig = int(len(np.unique(ctg["gap"])))
        # "closed" = a contig anchored by both flanks (gappadder_amd/pick_contigs.py, anchor 30 then 15 like the reference's two
        # bwa scores); host-side, outside the timed region.  With one 300-bp library the recruited reads reach ~450 bp into a
        # 2-kb gap from each side, so the reference's first round cannot close these gaps either.
        from gappadder_amd.pick_contigs import pick_gap_sequence
        seq_all = d_seq[:n_seq].cpu().numpy().tobytes().decode()
        by_gap = {}
        for c in ctg:
            by_gap.setdefault(int(c["gap"]), []).append(("c", seq_all[int(c["seq_off"]):int(c["seq_off"]) + int(c["length"])]))
        gaps_closed = sum(1 for g, cs in by_gap.items()
                          if pick_gap_sequence(cs, flanks[g][0], flanks[g][1], 30) or pick_gap_sequence(cs, flanks[g][0], flanks[g][1], 15))
        out = {
            "metric": "reads_screened_per_s", "value": world * n_reads / (dt / args.steps), "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s: %d gaps x %d bp in %d x %.1f Mb scaffolds, %d x %d-bp read records (+ as many 32-B alignment "
                                   "records) per GPU, k=%d kv=%d, IS 300/30; step = k-mer screen + alignment tagger + second hop + "
                                   "per-gap pools + per-gap assembly" % (args.config, len(gaps), glen, nscf, slen / 1e6, n_reads, L, k, kv),
                       "reads_per_gpu": n_reads, "gaps": int(len(gaps)), "k": k, "kv": kv,
                       "sharding": ("reads sharded over ranks, gaps replicated; per-gap pools exchanged to one owner rank per gap (all-to-all-v), "
                                    "then the assembled sequences gathered" if (args.exchange and world > 1) else
                                    "reads sharded over ranks, gaps replicated; RCCL only gathers the assembled sequences")},
            "gaps_per_s": world * len(gaps) / (dt / args.steps),
            "gaps_closed_per_s": world * gaps_closed / (dt / args.steps),
            "roofline": {"bound": "hbm", "kernel": "screen_filter (software-pipelined wave kernel with LDS pre-filter; the plain kernel when the key set is too large for it, as at C4)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(n_reads, L, k),
                         "algorithmic_bytes_per_launch": n_reads * rb, "avg_launch_ms": filt_ms,
                         "frac_of_measured_copy_6290": achieved / 6290.0},
            "phases_ms": phases,
            "phases_note": "HIP-event spans per kernel group; tagger + second hop run on a second stream beside the screen, so "
                           "their spans include queueing behind the filter kernel (stand-alone: tagger 0.41 ms, verify 0.16 ms; GF_BENCH_SERIAL=1 runs everything on one stream)",
            "tagger_gbs": n_reads * 32 / (phases["tag_alignments"] * 1e-3) / 1e9,
            "counts": {"screen_hits": n_hits, "tagger_hits": n_thits, "second_hop_hits": n_lhits, "pool_keys": n_keys,
                       "pooled_reads": int(pool_off[-1]), "contigs": n_ctg, "contig_bases": n_seq,
                       "gaps_with_contig": gaps_with_contig, "gaps_closed": gaps_closed},
        }
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args, cfg, flanks, gaps, first_pair, L, k, kv, d_hits, n_hits, d_pool, pool_off,
                                               ctg, d_seq, n_seq, dt / args.steps, n_reads, B)
        if world == 1 and args.config == "C2" and not args.no_cpu and not args.no_human_scale and not (args.reads_given or args.k):
            out["human_scale_shard"] = human_scale_shard()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def human_scale_shard():
    """BASELINE.json's metric names the 30x human-scale synthetic (configs[3], 8 GPUs).  The bench line is quoted on the largest
    single-GPU configuration (C2); this adds one GPU's shard of the human-scale run (C4: 19 840 gaps, k=51, 112.5 M of the 900 M
    read records — what every rank of the 8-GPU job processes) measured by the same code in a child process."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", "C4", "--steps", "5", "--warmup", "2", "--no-cpu"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
        d = json.loads(r.stdout.decode().strip().splitlines()[-1])
        return {"workload": d["config"]["workload"], "ms_per_step": d["ms_per_step"], "reads_per_s": d["value"], "gaps_per_s": d["gaps_per_s"],
                "steps": d["steps"], "warmup": d["warmup"], "roofline_frac": d["roofline"]["frac"], "filter_ms": d["roofline"]["avg_launch_ms"],
                "phases_ms": d["phases_ms"], "counts": d["counts"]}
    except Exception as e:      # the headline line must not depend on this extra
        return {"error": repr(e)[:200]}


def pmc_traffic(n_reads, L, k):
    """Bytes per launch of the dominant kernel from the committed PMC passes (profiles/r01_traffic.json: rocprofv3 --pmc
    FETCH_SIZE and WRITE_SIZE in separate runs of this same command, gfx950 x2 correction applied to FETCH_SIZE).  Counters
    cannot be collected from inside the timed run; null when no profile matches this configuration."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        if t["reads_per_launch"] == n_reads and t["read_len"] == L and t["k"] == k:
            return t["traffic_bytes_per_launch"]
    except Exception:
        pass
    return None


def cpu_baseline(args, cfg, flanks, gaps, first_pair, L, k, kv, d_hits, n_hits, d_pool, pool_off, ctg, d_seq, n_seq, gpu_step_s,
                 n_reads, B):
    """The oracle (oracle/gp_oracle.c, OpenMP over all host cores; kind "port") on a bounded sample of the same step:
    k-mer screen + alignment tagger on the first --cpu-sample-reads reads of rank 0's shard, and the assembly of the first
    gaps' pools.  Also the checker: the GPU's hits on that prefix and its contigs for those gaps must equal the oracle's."""
    from oracle import c_oracle as CO
    n_s = min(args.cpu_sample_reads, args.reads) // 2 * 2
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:   # a cgroup CPU quota (cpu.max "quota period") caps the usable cores below the visible ones
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(q) // int(per)))
    except Exception:
        pass
    ocfg = np.frombuffer(cfg.tobytes(), dtype=CO.SYNTH_CFG).copy()
    CO.set_threads(cores)
    packed, recs = CO.synth_pairs(ocfg, first_pair, n_s // 2)
    blob = CO.unpack_reads(packed, L)
    t0 = time.perf_counter()
    ohits = CO.screen_reads(blob, L, flanks, k, 1, 0, cores)
    t1 = time.perf_counter()
    CO.tag_alignments(recs, gaps, 300, 30)
    t2 = time.perf_counter()
    hits = np.frombuffer(d_hits[:n_hits * 8].cpu().numpy().tobytes(), dtype=B.HIT)
    sub = np.sort(hits[hits["read"] < n_s], order=["gap", "read"])
    ok = len(sub) == len(ohits) and sub.tobytes() == ohits.astype(B.HIT).tobytes()
    # assembly sample: the first gaps' pools exactly as the GPU built them
    n_g = min(len(gaps), 256)
    rb = (L + 3) // 4
    pool = d_pool[:int(pool_off[n_g]) * rb].cpu().numpy().reshape(-1, rb)
    pblob = CO.unpack_reads(pool, L)
    seq = d_seq[:n_seq].cpu().numpy().tobytes()
    from concurrent.futures import ThreadPoolExecutor
    t3 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:   # gaps are independent (assemble_gaps.py:296-299 uses a process pool)
        exp = list(ex.map(lambda g: CO.assemble_pool(pblob[int(pool_off[g]) * L:int(pool_off[g + 1]) * L], L, k, kv), range(n_g)))
    t4 = time.perf_counter()
    ok_asm = True
    for g in range(n_g):
        mine = sorted((seq[int(c["seq_off"]):int(c["seq_off"]) + int(c["length"])].decode(), int(c["n_nodes"]), int(c["cov_sum"]))
                      for c in ctg[ctg["gap"] == g])
        ok_asm = ok_asm and mine == sorted(exp[g])
    # whole-step CPU time extrapolated from the two samples (recruit scales with reads, assembly with gaps)
    cpu_step = (t2 - t0) * (n_reads / n_s) + (t4 - t3) * (len(gaps) / n_g)
    return {"value": n_reads / cpu_step, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "recruit: first %d reads of rank 0's shard (k-mer screen %.2f s + alignment tagger %.2f s, OpenMP %d threads); "
                      "assembly: pools of the first %d gaps (%.2f s, same thread count); value = reads / (sample times scaled to the whole "
                      "step); oracle/gp_oracle.c" % (n_s, t1 - t0, t2 - t1, cores, n_g, t4 - t3),
            "recruit_reads_per_s": n_s / (t2 - t0), "assembly_gaps_per_s": n_g / (t4 - t3),
            "parity_on_sample": bool(ok and ok_asm), "parity_recruit": bool(ok), "parity_assembly": bool(ok_asm),
            "sample_hits": int(len(ohits)), "sample_contigs": int(sum(len(e) for e in exp))}


if __name__ == "__main__":
    main()
