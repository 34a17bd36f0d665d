#!/usr/bin/env python3
"""bench.py — reads screened/s + gaps closed/s of the recruit + local-assembly hot path on MI355X.

A step = one pass of the hot path over the seeded synthetic workload (include/gf_synth.h), inputs resident in HBM when the
timed region starts:   k-mer screen + alignment tagger + second hop  ->  per-gap pools (per library, merged in library order)
                       ->  [N > 1: pools to the gap's owner rank, RCCL all-to-all]  ->  per-gap assembly, every (k, kv)
                       ->  flank anchoring (which gaps are closed).
Default workload = the configuration BASELINE.json quotes its metric on, configs[3] ("C4"): human-scale draft,
19 840 gaps x 2 kb in 620 x 5 Mb scaffolds, 900 M 150-bp read records (+ 900 M alignment records), k=51 — all of it on ONE
GPU at N=1 (63 GB resident).  At N > 1 the SAME reads are split over the ranks (rank r owns a contiguous range of pairs;
scaling = strong), gaps and the flank index are replicated, and every gap is assembled exactly once, by its owner rank, from
the recruits of all ranks (the reference maps each gap to one Pool task, assemble_gaps.py:296-299).

    python bench.py --gpus N --steps K --warmup W [--config C2|C3|C4|C5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel = the screen filter; algorithmic bytes = ceil(2L/8) per read,
SURVEY.md §8d) and `cpu_baseline` (the oracle's C restatement — kind "port" — on the host cores on a bounded sample of the
same workload — stripes over the WHOLE read and gap range, `parity_sample_ranges` — also checked bit-for-bit against the GPU's results on that sample).  At N=1 the default run appends `extras`:
the same step on C2 (configs[1]) and on C5 (configs[4]: + mate-pair library IS 5000, k in {31,41,51}), each in a child process.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

# seed, scaffold_len, n_scaffolds, gaps_per_scaffold, gap_len, read records (whole job), [(k, kv)]
PRESETS = {
    "C2": (20260002, 5_000_000, 50, 20, 2000, 50_000_000, [(31, 29)]),
    "C3": (20260003, 4_600_000, 1, 200, 1000, 5_000_000, [(41, 39)]),
    "C4": (20260004, 5_000_000, 620, 32, 2000, 900_000_000, [(51, 49)]),
    # C4's draft and short library + a mate-pair library (IS 5000 / sd 500 -> the tagger's long-IS branch,
    # collect_reads_for_gaps.py:275-278) + the multi-k sweep; per-gap pools = both libraries in library order (merge_reads.py:43-51)
    "C5": (20260004, 5_000_000, 620, 32, 2000, 900_000_000, [(31, 29), (41, 39), (51, 49)]),
    # stress workload (not a BASELINE configuration): C2's draft with planted repeats — every eighth gap at a copy of a 50-copy repeat
    # family (0.5-5 kb, either strand), two of every eight at the copies of a 2-copy repeat, one with a low-complexity run in its
    # flank (include/gf_synth.h `repeats`).  Reads hit up to 50 gaps at once, pools of the repeat gaps hold thousands of reads.
    "C2R": (20260002, 5_000_000, 50, 20, 2000, 50_000_000, [(31, 29)]),
    # ... and the human-scale draft with planted repeats (every 16th gap at a copy of a 50-copy family: 1 240 repeat gaps in 25 families)
    "C4R": (20260004, 5_000_000, 620, 32, 2000, 900_000_000, [(51, 49)]),
    # ... and C2's draft with planted repeats, the mate-pair library and the multi-k sweep of configs[4] at its scale (not a BASELINE
    # configuration): gaps DO close here, so the truth check, the open-gap census and the contig-merge round say what a repeat-bearing
    # draft does to the anchors and the picks (VERDICT r4 missing 4)
    "C2RM": (20260002, 5_000_000, 50, 20, 2000, 50_000_000, [(31, 29), (41, 39), (51, 49)]),
}
REPEATS = {"C2R": (8, 50), "C4R": (16, 50), "C2RM": (8, 50)}   # config -> (period, copies) of the planted repeats
# Mate-pair library of C5.  SURVEY.md §8d says "extra 100 M records" = 4.8x: at KMC's min-count 2 a k-mer of the gap interior (covered by
# this library only) is then missing with P = e^-3.9 (1 + 3.9) = 10 % per position, so no 2-kb gap can close (measured on the GPU:
# 0 of 19 840; tools/closure_experiment.py: 0/6 at 4.8x, 1/6 at 10x, 28/30 at 15x, 30/30 at 19x).  The bench therefore draws
# 400 M records (19.4x); --mp-reads 100000000 reproduces the survey's figure.
MATE_PAIRS = {"C5": 400_000_000, "C2RM": 32_000_000}   # configs with the IS 5000 / sd 500 library: its read records (19.4x of the draft)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C4", choices=sorted(PRESETS),
                    help="BASELINE.json workload: C4 (default: the metric's configuration, 900 M reads), C2, C3, C5 (C4 + mate pairs + multi-k); "
                         "C2R = C2's draft with planted repeats (stress workload)")
    ap.add_argument("--reads", type=int, default=0, help="read records of the first library, WHOLE JOB (default: the config's)")
    ap.add_argument("--mp-reads", type=int, default=-1, help="C5: read records of the mate-pair library, whole job (default 400 M = 19.4x; SURVEY.md §8d names 100 M)")
    ap.add_argument("--cpu-sample-reads", type=int, default=4_000_000, help="reads of the first library the oracle sees, in stripes over the whole library (-1: all of them)")
    ap.add_argument("--cpu-sample-gaps", type=int, default=256, help="gaps whose pools the oracle assembles and picks from, drawn over the whole gap list (-1: all)")
    ap.add_argument("--asm-tiebreak", default="counts", choices=["counts", "none"],
                    help="error removal between branches of equal coverage: counts (default: fewer weak k-mers win) or none (reference-shaped: "
                         "sequence order alone, nothing Velvet could not have known, assemble_gaps.py:56-79); the oracle follows")
    ap.add_argument("--merge-round", default="auto", choices=["auto", "on", "off", "host"],
                    help="the contig-merge round (assemble_gaps.py:301-306) for the gaps the first pick leaves open, on the device, INSIDE the timed step: "
                         "auto (default) = on where a library can span the gaps (the configurations with the mate-pair library: C5, C2RM) — with the "
                         "300-bp library alone every 2-kb gap stays open by construction, nothing can be closed by merging, and the round over all "
                         "gaps is measured once behind the timed region instead (`contig_merge_round_all_gaps`); host = the host twin, untimed")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the C2 / C5 child runs that the default N=1 run appends as `extras`")
    ap.add_argument("--e2e-only", default="", help="run only the file-based end-to-end extra on this configuration (C2 / C3) and print its object")
    ap.add_argument("--dump-contigs", default="", help="rank 0 writes the gathered contigs (sorted) of the last step to this JSON file")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` from a bare shell (N > 1, no torch.distributed.run around it): start the N ranks as a CHILD
    `python -m torch.distributed.run` before this process has touched the GPU (never an exec, never after a HIP call), relay rank 0's
    JSON line and exit with the child's code.  On a box with fewer than N GPUs the ranks share GPU 0 and the collectives go through
    gloo (functional mode: the same kernels and the same exchange, no RCCL); the line then says so."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.device_count() < args.gpus and "GF_BENCH_BACKEND" not in env:     # (the ranks are a spawned child either way: never an exec of this process)
        env.update(GF_BENCH_BACKEND="gloo", GF_BENCH_ONE_GPU="1")
        sys.stderr.write("bench.py: %d rank(s) on %d GPU(s): ranks share cuda:0, collectives over gloo\n" % (args.gpus, torch.cuda.device_count()))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    sys.stdout.write(r.stdout.decode())
    sys.stdout.flush()
    sys.exit(r.returncode)


def main():
    args = parse_args()
    if args.e2e_only:
        print(json.dumps(e2e_files(args.e2e_only)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    out, rank, world = run(args)
    # the extras run in child processes AFTER this process has released its device memory (run()'s tensors and contexts are gone)
    if rank == 0 and world == 1 and args.config == "C4" and not args.no_extras and not args.no_cpu and not args.reads and args.mp_reads < 0:
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        out["extras"] = {"C2": child_run(["--config", "C2", "--steps", "20", "--warmup", "2"]),
                         # stress workload: C2's draft with planted repeats (reads that hit 50 gaps, pools of thousands of reads)
                         "C2_with_planted_repeats": child_run(["--config", "C2R", "--steps", "5", "--warmup", "1"]),
                         "C5_mate_pair_multi_k": child_run(["--config", "C5", "--steps", "3", "--warmup", "1"]),
                         # SURVEY.md §8d's own figure for the mate-pair library (100 M records = 4.8x): recorded as it is — at KMC's
                         # min-count 2 that depth leaves holes in every 2-kb gap, so (nearly) nothing closes; see MATE_PAIRS
                         "C5_survey_sized_100M_mate_pairs": child_run(["--config", "C5", "--mp-reads", "100000000", "--steps", "2", "--warmup", "1"])}
        # the product path on files: the CLI on a C3-sized BAM + FASTQ pair, wall time split by stage
        out["extras"]["e2e_files_C3"] = e2e_files("C3")
        # the second half of BASELINE.json's metric: 2-kb gaps cannot close from a 300-bp library alone (C4: 0 by construction of the
        # workload); configs[4] adds the mate-pair library and the multi-k sweep, and its closed count is part of this line
        c5 = out["extras"]["C5_mate_pair_multi_k"]
        if "gaps_closed_per_s" in c5:
            out["gaps_closed_per_s_with_mate_pairs"] = {"value": c5["gaps_closed_per_s"], "config": "C5 (extras.C5_mate_pair_multi_k)",
                                                        "gaps_closed": c5.get("counts", {}).get("gaps_closed"),
                                                        "gaps_closed_correct": c5.get("counts", {}).get("gaps_closed_correct"),
                                                        "correct_per_s": c5.get("gaps_closed_correct_per_s"), "ms_per_step": c5["ms_per_step"]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def run(args):

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # GF_BENCH_BACKEND=gloo + GF_BENCH_ONE_GPU=1: the multi-rank code path with every rank on cuda:0 (single-GPU boxes, the
    # 2-rank GPU test); the real runs use nccl (= RCCL) with one GPU per rank
    backend = os.environ.get("GF_BENCH_BACKEND", "nccl")
    one_gpu = bool(os.environ.get("GF_BENCH_ONE_GPU"))
    if one_gpu:
        local = 0
    # GF_BENCH_FORCE_EXCHANGE=1: the multi-rank code path — process group, side stream, second-hop union, owner exchange, final gather —
    # at ANY world size, world 1 included: on a one-GPU box this is how the `nccl` (= RCCL) branch of the collectives gets executed
    multi = world > 1 or os.environ.get("GF_BENCH_FORCE_EXCHANGE") == "1"
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29655")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        dist.barrier()
        # RCCL prints a version banner through C stdio when its first communicator comes up; stdout being a pipe, it would sit in
        # the C buffer until exit and land BEHIND rank 0's JSON line: flush it out now
        C.CDLL(None).fflush(None)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    from gappadder_amd import _lib as B
    from gappadder_amd import sharding as SH
    from gappadder_amd.hip_api import GapFill
    from gappadder_amd.pipeline import DeviceLibrary, Pipeline

    seed, slen, nscf, gps, glen, dreads, kk = PRESETS[args.config]
    if os.environ.get("GF_BENCH_KPAIRS"):            # diagnostics: other (k, kv) pairs than the config's, "31,29;41,39" (the line then names them)
        kk = [tuple(int(x) for x in p.split(",")) for p in os.environ["GF_BENCH_KPAIRS"].split(";")]
    total_reads = (args.reads or dreads) // 2 * 2
    L = 150
    # (name, IS, sd, library number, records, pull the mates of k-mer-screen hits?)  A read that shares a k-mer with a flank lies AT
    # the gap; its mate lies one insert away: inside or next to the gap for the 300-bp library (pulled: north_star's "candidate read
    # pairs"), 5 kb away for the mate-pair library (not pulled: those mates are recruited by the tagger's rules when they belong to the
    # gap — mate unmapped / discordant / clipped inside the focal window — and are distant sequence otherwise; GF_BENCH_MP_PAIRS=1 pulls
    # them anyway: 819 instead of 620 reads per pool, the same gaps closed)
    lib_defs = [("short-insert", 300, 30, 0, total_reads, 1)]
    if args.config in MATE_PAIRS:
        mp = MATE_PAIRS[args.config] if args.mp_reads < 0 else args.mp_reads
        lib_defs.append(("mate-pair", 5000, 500, 1, mp // 2 * 2, int(os.environ.get("GF_BENCH_MP_PAIRS", "0"))))

    lib = B.lib()
    rb = lib.gf_packed_read_bytes(L)
    gf = GapFill(local)
    # One stream by default: every placement of the tagger beside the screen was measured (beside the filter, beside the verification
    # pass, as one-wave workgroups next to the filter's) and gave the SUM of the stand-alone times within 1-3 % (C4: 68.0 / 70.3 / 69.0
    # vs 68.3 ms in a row) — the kernels take turns on the memory system; in a row every kernel's HIP-event span is its own duration.
    # GF_BENCH_TWO_STREAMS=1: tagger + second hop on a second context / stream beside the filter
    # (one per library: the tagger caches its coarse bin map per insert-size window)
    # GF_BENCH_TAG_AHEAD=1: the tagger of the NEXT step runs on a second stream beside this step's assembly (Pipeline.tag_ahead)
    tag_ahead = os.environ.get("GF_BENCH_TAG_AHEAD", "0") == "1"
    serial = os.environ.get("GF_BENCH_TWO_STREAMS", "0") != "1" and not tag_ahead
    gf2s = [gf if serial else GapFill(local) for _ in lib_defs]

    rep_p, rep_c = REPEATS.get(args.config, (0, 50))
    cfg0 = GapFill.synth_cfg(seed=seed, scaffold_len=slen, n_scaffolds=nscf, gaps_per_scaffold=gps, gap_len=glen, read_len=L,
                             insert_mean=300, insert_sd=30, repeat_period=rep_p, repeat_copies=rep_c)
    gaps, flanks = GapFill.synth_layout(cfg0)
    n_gaps = len(gaps)
    if os.environ.get("GF_BENCH_MAX_GAPS_PER_KMER"):     # the repeat mask: flank k-mers shared by more gaps than this leave the index (experiments on the repeat drafts)
        gf.set_option("max_gaps_per_kmer", int(os.environ["GF_BENCH_MAX_GAPS_PER_KMER"]))
    gf.set_gaps(gaps, int(cfg0["n_scaffolds"][0]), flanks)
    for g2 in gf2s:
        if g2 is not gf:
            g2.set_gaps(gaps, int(cfg0["n_scaffolds"][0]), None)
            # (GF_BENCH_TAG_LIGHT=1: the one-wave tagger variant that fits on the CUs whose LDS the filter owns — measured: no gain, the two
            # kernels contend for the memory system, C4 78.6 vs 77.4 ms)
            g2.set_option("tag_light", int(os.environ.get("GF_BENCH_TAG_LIGHT", "1" if tag_ahead else "0")))

    # The whole step lives in the package (gappadder_amd/pipeline.py: residency, sizing pass, capacities, recruit -> hop -> keys -> pools ->
    # merge / owner exchange -> assembly -> pick): this file generates the inputs, calls it and times it
    # GF_BENCH_TAG_KEYS=0: the tagger streams the 32-byte records themselves instead of their 8-byte key column (ablation)
    key_column = os.environ.get("GF_BENCH_TAG_KEYS", "1") != "0"
    merge_mode = args.merge_round
    merge_on = merge_mode == "on" or (merge_mode == "auto" and args.config in MATE_PAIRS and (args.mp_reads != 0))
    pipe = Pipeline(gf, n_gaps, L, kk, device=dev, world=world, rank=rank, backend=backend, force_exchange=multi and world == 1, key_column=key_column,
                    merge_in_step=merge_on)
    pipe.tag_after_filter = not serial and os.environ.get("GF_BENCH_TAG_AFTER_FILTER", "0") == "1"
    pipe.tag_ahead = tag_ahead
    h = gf.handle

    # ---- inputs resident in HBM (torch = device-memory plumbing) ----
    libs = []
    for name, is_mean, is_sd, lib_no, n_total, pull_mates in lib_defs:
        cfg = GapFill.synth_cfg(seed=seed, scaffold_len=slen, n_scaffolds=nscf, gaps_per_scaffold=gps, gap_len=glen, read_len=L,
                                insert_mean=is_mean, insert_sd=is_sd, library=lib_no, repeat_period=rep_p, repeat_copies=rep_c)
        p0, p1 = SH.shard_range(n_total // 2, rank, world)          # strong scaling: the same pairs, split
        n_reads = 2 * (p1 - p0)
        d_reads = torch.empty(n_reads * rb + 64, dtype=torch.uint8, device=dev)
        d_recs = torch.empty(max(1, n_reads) * 32, dtype=torch.uint8, device=dev)
        gf.synth_pairs_dev(cfg, p0, p1 - p0, d_reads.data_ptr(), d_recs.data_ptr())
        lb = DeviceLibrary(name, is_mean, is_sd, n_reads, d_reads, d_recs, pull_mates=pull_mates, n_total=n_total, first_pair=p0,
                           tag_ctx=None if gf2s[len(libs)] is gf else gf2s[len(libs)])
        lb.cfg = cfg
        libs.append(pipe.add_library(lb))
    gf.sync()
    k_screen = pipe.k_screen

    # ---- sizing pass (untimed): second-hop table rows, pooled reads, exchange slots ----
    pipe.prepare()
    screen_dropped = pipe.screen_dropped
    if args.config in ("C2", "C3", "C4", "C5"):
        assert screen_dropped == 0, screen_dropped
    max_pool_rows, asm_bound, per_gap = pipe.max_pool_rows, pipe.asm_bound, pipe.per_gap
    if os.environ.get("GF_BENCH_SCREEN_VARIANT"):    # filter kernel (experiments: 17 = pass A with unaligned runs)
        gf.set_option("screen_variant", int(os.environ["GF_BENCH_SCREEN_VARIANT"]))
    gf.set_option("asm_tiebreak", 1 if args.asm_tiebreak == "counts" else 0)
    if os.environ.get("GF_BENCH_ASM_SIMPLIFY"):      # rounds of tip clipping + bubble popping (experiments; the parity sample then disagrees unless it is 2)
        gf.set_option("asm_simplify", int(os.environ["GF_BENCH_ASM_SIMPLIFY"]))
    if os.environ.get("GF_BENCH_ASM_PRE_FRAC8"):     # count phase: share of the LDS region the pre-count bit arrays may take (eighths; experiments)
        gf.set_option("asm_pre_frac8", int(os.environ["GF_BENCH_ASM_PRE_FRAC8"]))
    if os.environ.get("GF_BENCH_ASM_SWEEP"):         # 0: one assembly launch per (k, kv) pair instead of the fused sweep (experiments)
        gf.set_option("asm_sweep", int(os.environ["GF_BENCH_ASM_SWEEP"]))
    if os.environ.get("GF_BENCH_ASM_THREADS"):       # threads per gap in the assembly kernel (1024 / 512 / 256; default: by the pool bound)
        gf.set_option("asm_threads", int(os.environ["GF_BENCH_ASM_THREADS"]))
    n_lib = len(libs)
    need_merge = pipe.need_merge

    d_astat = torch.zeros(4, dtype=torch.int64, device=dev)      # windows, k-mers counted exactly, surviving k-mers, nodes (all steps, all k)
    gf.set_option("asm_stats_ptr", d_astat.data_ptr())
    d_dbg = None
    if os.environ.get("GF_BENCH_ASM_PROBE"):      # diagnostic (needs GF_DIAGNOSTICS=1): per-gap phase stamps of the LAST assembly launch
        d_dbg = torch.zeros(n_gaps * 16, dtype=torch.int64, device=dev)
        gf.set_option("asm_dbg_ptr", d_dbg.data_ptr())
    pipe.step(args.warmup)
    pipe.barrier()
    d_astat.zero_()
    torch.cuda.synchronize()
    ctxs = pipe.contexts()
    [g_.timing(True) for g_ in ctxs]
    # Multi-rank runs: HIP-event spans of the parts of a step that do not shrink with the number of ranks (the union of the ranks'
    # second-hop rows, merged on every rank) or that exist only there (the owner exchange: pack, all-gather of counts, all-to-all,
    # merge) — `fixed_ms` of the line, for the scaling prediction of DESIGN.md §6
    pipe.fixed_on = multi
    t0 = time.perf_counter()
    pipe.step(args.steps)
    pipe.barrier()
    dt = time.perf_counter() - t0
    pipe.fixed_on = False
    fixed_ms = pipe.fixed_ms(args.steps)

    def ktime(idx):     # (total ms, launches) of one kernel group over all contexts
        tt = [g_.kernel_time(idx) for g_ in ctxs]
        return sum(t for t, _ in tt), sum(n for _, n in tt)
    kt = {name: ktime(idx) for name, idx in (("screen_filter", B.KERNEL_SCREEN), ("screen_verify", B.KERNEL_VERIFY),
                                             ("tag_alignments", B.KERNEL_TAG), ("tag_low_mapq", B.KERNEL_LOWMAPQ),
                                             ("pools", B.KERNEL_POOL), ("assemble", B.KERNEL_ASSEMBLE), ("pick_anchored", B.KERNEL_PICK),
                                             ("merge_round", B.KERNEL_MERGE))}
    [g_.timing(False) for g_ in ctxs]
    if multi:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    step_s = dt / args.steps

    if d_dbg is not None:
        d = d_dbg.cpu().numpy().reshape(-1, 16)
        d = d[d[:, 6] > 0]
        ph = np.diff(d[:, [0, 1, 2, 3, 4, 7, 5, 6]], axis=1) / 100.0
        names = ["P1 count", "P2 survivors", "P3 graph+index", "P4 links", "error removal", "ranking", "emission"]
        sys.stderr.write("assembly phases of the last (k, kv) launch, us per gap (%d gaps with reads): " % len(d) +
                         ", ".join("%s %.1f (max %.1f)" % (nm, ph[:, i].mean(), ph[:, i].max()) for i, nm in enumerate(names)) +
                         "; total %.1f\n" % ph.sum(1).mean())
        tot = ph.sum(1)
        sys.stderr.write("  per-gap total, percentiles 50 / 90 / 99 / 99.9 / max: %s us; the slowest 1 %% of the gaps take %.1f %% of the time\n"
                         % (" / ".join("%.0f" % np.percentile(tot, q) for q in (50, 90, 99, 99.9, 100)),
                            100.0 * np.sort(tot)[-max(1, len(tot) // 100):].sum() / tot.sum()))
        sub = d[(d[:, 8] > d[:, 0]) & (d[:, 9] >= d[:, 8])]
        if len(sub):   # the count phase's parts (gaps whose pre-count ran): bit arrays, prefixes + table init, exact pass
            sys.stderr.write("  count phase: pre-count pass %.1f, prefixes + table init %.1f, exact pass %.1f us (%d gaps)\n"
                             % ((sub[:, 8] - sub[:, 0]).mean() / 100.0, (sub[:, 9] - sub[:, 8]).mean() / 100.0, (sub[:, 1] - sub[:, 9]).mean() / 100.0, len(sub)))
        sys.stderr.write("  windows %.0f, counted %.0f, survivors %.0f, nodes %.0f (max %d); count table global in %.0f %% of the gaps; graph plan LDS / LDS + global pairs / global: %s\n"
                         % (d[:, 13].mean(), d[:, 10].mean(), d[:, 12].mean(), d[:, 14].mean(), d[:, 14].max(), 100.0 * d[:, 11].mean(),
                            " / ".join("%.0f %%" % (100.0 * (d[:, 15] == v).mean()) for v in (0, 1, 2))))
    # ---- results of the last step ----
    astat = (d_astat.cpu().numpy().astype(np.float64) / max(1, args.steps))      # per step (this rank's gaps)
    gf.set_option("asm_stats_ptr", 0)
    res = pipe.fetch()          # synchronises; raises on any capacity / overflow flag of the step
    n_ctg, n_seq, n_closed_local = res.n_contigs, res.n_seq, res.n_closed
    asm_off_t, asm_pool_t, asm_rows_total = res.asm_off_t, res.asm_pool_t, res.asm_rows_total
    d_seq, d_best = pipe.d_seq, pipe.d_best
    _t, _m, _l = C.c_int(0), C.c_uint32(0), C.c_uint32(0)     # the last (k, kv) pair's launches: threads per gap, gaps handed to the 1 024-thread launch, pools beyond the bound
    assert lib.gf_assemble_last_launch(gf.handle, C.byref(_t), C.byref(_m), C.byref(_l)) == 0
    asm_launch = {"threads_per_gap": _t.value, "gaps_to_the_whole_cu_launch": _m.value, "pools_to_the_last_launch": _l.value}
    ctg = res.contigs
    n_closed, n_ctg_all, gaps_with_contig = n_closed_local, n_ctg, int(len(np.unique(ctg["gap"])))
    gather_ms = None
    # ground truth (untimed): the sequence picked for every closed gap of this rank against the true bases behind the planted gap
    seq_host = res.seq
    truth = truth_check(cfg0, gaps, flanks, ctg, seq_host, res.best, GapFill)
    assert truth["closed"] == n_closed_local, (truth["closed"], n_closed_local)
    n_correct = truth["correct"]
    census = None
    if not multi and n_closed_local:      # (runs that close gaps at all: with the 300-bp library alone every 2-kb gap is a coverage hole)
        census = open_gap_census(cfg0, gaps, ctg, seq_host, res.best, GapFill)
    merge_round = None
    if res.merge is not None:
        # The reference merges a gap's contigs before it picks (assemble_gaps.py:301-306, 335-339).  The step picks, sends the contigs of the
        # gaps that pick left open through the contig merger ON THE DEVICE (gf_merge_open_gaps_dev: dedup, prefilter, overlap evaluation,
        # path search, merged strings; no host synchronisation) and picks again over the merged contigs — inside the timed region: the
        # closed counts above include it
        merge_round = dict(res.merge, inside_the_timed_step=True)
        idx = 0x7FFFFFFF - ((res.best >> np.uint64(1)) & np.uint64(0x7FFFFFFF)).astype(np.int64)
        by_merge = (res.best != 0) & (idx >= res.merge["contigs_before"])
        wrong_by_merge = sum(1 for g_ in truth["wrong_gaps"] if by_merge[g_])
        merge_round["closed_correct"] = int(by_merge.sum()) - wrong_by_merge        # (of the gaps a MERGED contig closes: equal to the true sequence)
        merge_round["gaps_closed_without_merging"] = int((res.best != 0).sum() - by_merge.sum())
    elif merge_mode == "host" and not multi and n_closed_local and n_gaps - n_closed_local <= max(1000, n_gaps // 4):
        # the host twin of the round (MergeContigs.merge_sets: two batched GPU calls + host path search), AFTER the timed region: comparison runs
        tm0 = time.perf_counter()
        mg = pipe.merge_open_gaps(res)
        tm = time.perf_counter() - tm0
        merge_round = {"gaps_tried": mg["gaps_tried"], "gaps_skipped_large": mg["gaps_skipped_large"], "gaps_with_new_contigs": mg["gaps_with_new_contigs"], "new_contigs": mg["new_contigs"],
                       "gaps_closed_by_merging": len(mg["closed"]), "seconds_untimed": tm, "inside_the_timed_step": False}
        if mg["closed"]:
            c2, s2, b2 = mg["arrays"]
            t2 = truth_check(cfg0, gaps, flanks, c2, s2, b2, GapFill)
            merge_round.update(closed_correct=t2["correct"], gaps_closed_total=n_closed_local + len(mg["closed"]),
                               gaps_closed_correct_total=n_correct + t2["correct"])
    if multi:
        red = torch.tensor([n_closed_local, n_ctg, gaps_with_contig, asm_rows_total, n_correct], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(red, op=dist.ReduceOp.SUM)
        n_closed, n_ctg_all, gaps_with_contig, asm_rows_total, n_correct = (int(x) for x in red)
        # final gather on rank 0 (north_star: "RCCL over xGMI only for the final gather of closed sequences"): the picked contig
        # of every closed gap — or every contig when --dump-contigs asks for the full comparison
        tg = time.perf_counter()
        seq_local = seq_host
        best = res.best
        if args.dump_contigs:
            sel = range(n_ctg)
        else:
            sel = [0x7FFFFFFF - int((int(b) >> 1) & 0x7FFFFFFF) for b in best if b]
        payload = SH.encode_contigs([(int(ctg[i]["gap"]), int(ctg[i]["k"]), int(ctg[i]["kv"]), int(ctg[i]["n_nodes"]), int(ctg[i]["cov_sum"]),
                                      seq_local[int(ctg[i]["seq_off"]):int(ctg[i]["seq_off"]) + int(ctg[i]["length"])].decode()) for i in sel])
        gathered = SH.gather_bytes(payload, dst=0, device=coll_dev)
        gather_ms = (time.perf_counter() - tg) * 1e3
        all_records = [r for part in (gathered or []) for r in SH.decode_contigs(part)]
    else:
        all_records = None
    if args.dump_contigs and rank == 0:
        if all_records is None:
            seq_local = seq_host
            all_records = [(int(c["gap"]), int(c["k"]), int(c["kv"]), int(c["n_nodes"]), int(c["cov_sum"]),
                            seq_local[int(c["seq_off"]):int(c["seq_off"]) + int(c["length"])].decode()) for c in ctg]
        with open(args.dump_contigs, "w") as f:
            json.dump({"contigs": sorted(all_records), "gaps_closed": n_closed}, f)

    if rank == 0:
        n_screened = sum(lb.n_total for lb in libs)
        t_filter, n_filter = kt["screen_filter"]
        filt_ms = t_filter / max(1, n_filter)                          # average launch of the filter (one launch per library and step)
        reads_per_launch = sum(lb.n_reads for lb in libs) / len(libs)
        achieved = reads_per_launch * rb / (filt_ms * 1e-3) / 1e9
        phases = {name: t / args.steps for name, (t, _) in kt.items()}
        k_s = k_screen
        wl = ("%s: %d gaps x %d bp in %d x %.1f Mb scaffolds; %s; k/kv %s; step = k-mer screen (k=%d) + alignment tagger + second hop + per-gap "
              "pools%s + per-gap assembly + flank anchoring" %
              (args.config, n_gaps, glen, nscf, slen / 1e6,
               " + ".join(("%s library IS %d/%d: %d x %d-bp read records (+ as many 32-B alignment records" + (" and their 8-B key column" if key_column else "") + ")%s") % (lb.name, lb.is_mean, lb.is_sd, lb.n_total, L, "" if lb.pull_mates else ", screen hits without their mates")
                          for lb in libs),
               ",".join("%d/%d" % p for p in kk), k_s,
               " (libraries merged in library order)" if n_lib > 1 else ""))
        out = {
            "metric": "reads_screened_per_s", "value": n_screened / step_s, "unit": "reads/s",
            # (ranks that share one GPU — the functional mode of a box with fewer GPUs than ranks — are not a scaling point: the line
            #  reports the PHYSICAL GPUs, the ranks beside them, and no scaling claim)
            "n_gpus": 1 if one_gpu else world, "ranks": world, "functional_mode": bool(one_gpu and world > 1),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_s * 1e3,
            "higher_is_better": True, "scaling": None if (one_gpu and world > 1) else "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": wl, "reads_total": n_screened, "reads_per_gpu": sum(lb.n_reads for lb in libs), "gaps": n_gaps,
                       "k_pairs": [list(p) for p in kk], "asm_tiebreak": args.asm_tiebreak,
                       "collectives": ("none (one rank)" if not multi else "RCCL (nccl backend), one GPU per rank" if backend == "nccl" else
                                       "%s through host memory%s" % (backend, ", all ranks on cuda:0 (functional mode)" if os.environ.get("GF_BENCH_ONE_GPU") else "")),
                       "sharding": ("single GPU: all reads and all gaps on one device" if not multi else
                                    "the same reads split over the ranks (contiguous pair ranges), gaps + flank index replicated; per-gap pools "
                                    "sent to one owner rank per gap (batches of %d gaps round-robin; device pack + ONE all-to-all with exact split "
                                    "sizes that also carries the per-gap counts + device merge, no host sync), every gap assembled once by its owner from all "
                                    "ranks' recruits; final gather of the closed gaps' contigs on rank 0" % pipe.batch)},
            "gaps_per_s": n_gaps / step_s,
            "gaps_closed_per_s": n_closed / step_s,
            "gaps_closed_correct_per_s": n_correct / step_s,
            "roofline": {"bound": "hbm",
                         "kernel": "screen_filter (one launch group per library and step: pf4_scatter_lines_kernel (pf4_scatter_kernel where a read has more than four probes) + pf4_probe_kernel + pf4_resolve_kernel + "
                                   "pf4_list_kernel — probes sorted into 256 slices of the level-1 bitmap as 4-byte pairs, each slice tested from LDS, "
                                   "read ids recovered from positions — on key sets beyond an L2 as at C4/C5; the software-pipelined "
                                   "screen_filter_pipe_kernel otherwise, as at C2)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(args.config, int(reads_per_launch), L, k_s, filt_ms, lib.gf_screen_kernels(h).decode()),
                         "algorithmic_bytes_per_launch": int(reads_per_launch * rb), "avg_launch_ms": filt_ms,
                         "frac_of_measured_copy_6290": achieved / 6290.0},
            "phases_ms": phases,
            "phases_note": "HIP-event spans per kernel group, summed over the libraries, per step; one stream: the step is their sum "
                           "(GF_BENCH_TWO_STREAMS=1 runs tagger + second hop on a second stream beside the filter: same step time within 1-3 %)",
            "counts": {"libraries": {lb.name: lb.counts for lb in libs}, "assembled_pool_reads": asm_rows_total, "contigs": n_ctg_all,
                       "gaps_with_contig": gaps_with_contig, "gaps_closed": n_closed, "gaps_closed_correct": n_correct,
                       "largest_pool_reads": max_pool_rows, "assembly_slice_rows": asm_bound, "screen_reads_not_verified_in_full": screen_dropped,
                       "pools_beyond_the_slice": int((per_gap > asm_bound).sum()), "assembly_last_launch": asm_launch},
            "closed_truth_check": {"what": "the picked sequence of EVERY closed gap (pick_contigs.py:341-349 slice of the winning contig) compared with the "
                                           "true bases behind the planted N-run, regenerated from include/gf_synth.h: genome[start-5 : end+6] on the forward "
                                           "strand, genome[start-6 : end+5] when the contig is reverse-complemented (the reference's slice keeps one anchor base)",
                                   "closed": n_closed, "correct": n_correct, "wrong_on_rank0": truth["wrong"][:8], "wrong_causes_rank0": truth["causes"]},
        }
        # the assembly's algorithmic bytes ("report bytes anyway"): 38 B x pool reads in + 2 x 20 B x distinct k-mers
        # (16-B key + 4-B count, written once, read once) + contig bases out, per (k, kv) pass; LDS/latency-bound, so no roofline claim
        asm_ms = phases["assemble"]
        asm_bytes = len(kk) * asm_rows_total * rb + 2 * 20 * float(astat[1]) * (world if world > 1 else 1) + n_seq
        out["assembly"] = {"algorithmic_bytes": int(asm_bytes), "gaps_per_s": n_gaps / (asm_ms * 1e-3) if asm_ms else None,
                           "us_per_gap_and_k": 1e3 * asm_ms / max(1, n_gaps * len(kk)), "ms": asm_ms, "k_passes": len(kk),
                           "pool_reads": asm_rows_total, "read_windows": int(astat[0]), "kmers_counted_exactly": int(astat[1]),
                           "surviving_kmers": int(astat[2]), "graph_nodes": int(astat[3]), "contig_bases": n_seq,
                           "achieved_GBps": asm_bytes / (asm_ms * 1e-3) / 1e9 if asm_ms else None,
                           "note": "per step; k-mers seen fewer than min_count times are stopped by the bit-array pre-count and never counted exactly"
                                   + ("; k-mer figures are rank 0's gaps x world" if world > 1 else "")}
        if census is not None:
            out["open_gap_census"] = census
        if merge_round is not None:
            out["contig_merge_round"] = merge_round
        if gather_ms is not None:
            out["final_gather_ms"] = gather_ms
        if multi:
            out["fixed_ms"] = dict(fixed_ms, note="per step and rank, HIP-event spans on the step's stream: second_hop_union = all-gather of the "
                                                  "ranks' second-hop rows (one packed slot per rank) + their merge, the same on every rank whatever their number; owner_exchange = pack + "
                                                  "the one all-to-all (rows + per-gap counts, exact split sizes) + merge of the per-gap pools (exists only in multi-rank runs)")
            out["config"]["forced_exchange_at_world_1"] = world == 1
            n_lib_ = len(libs)
            if getattr(pipe, "exact_exchange", False):
                out["exchange"] = {"form": "exact split sizes (rows per source, owner and library from the sizing pass), per-gap counts inside the one all-to-all",
                                   "bytes_sent_per_rank_and_step": int(pipe.xchg.bytes_sent), "header_bytes_per_peer": int(pipe.xchg.header_bytes),
                                   "collectives_per_step": 1 + n_lib_,
                                   "collectives": "1 all-to-all (pools + counts) + 1 all-gather per library (second-hop rows, their gaps and count in one packed slot)"}
            else:
                out["exchange"] = {"form": "equal slots padded to 1.25 x the largest block + all-gather of the counts (GF_XCHG=slots)",
                                   "bytes_sent_per_rank_and_step": int((world - 1) * (n_lib_ * pipe.slot_cap * rb + n_lib_ * n_gaps * 4)),
                                   "collectives_per_step": 2 + n_lib_,
                                   "collectives": "1 all-gather (counts) + 1 all-to-all (slots) + 1 all-gather per library (second-hop rows)"}
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, libs, flanks, gaps, L, kk, asm_pool_t, asm_off_t, ctg, d_seq, n_seq, d_best, step_s,
                                               n_screened, B, rb, merge_n0=res.merge["contigs_before"] if res.merge is not None else None)
    if rank == 0 and merge_mode == "auto" and not merge_on and not multi and kk:
        # no library spans the gaps (C2 / C3 / C4): the step above ran without the merge round; ONE more step with it, behind the timed region
        # (and behind every check of the timed steps' results), says what merging every open gap's contigs costs and yields — the reference
        # merges every gap (assemble_gaps.py:301-306)
        pipe.merge_in_step = True
        try:      # (an extra behind the timed region: the headline line must not depend on it)
            torch.cuda.synchronize()
            tm0 = time.perf_counter()
            pipe.step(1)
            pipe.barrier()
            tm = time.perf_counter() - tm0
            r2 = pipe.fetch()
            parity = None
            if not args.no_cpu and world == 1:      # a seeded sample of the gaps through the oracle's merger (tests/sample_check.py::merged_equal)
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import sample_check as SC
                from concurrent.futures import ThreadPoolExecutor
                n0_ = r2.merge["contigs_before"]
                rng = np.random.RandomState(20260630)
                with_m = np.unique(r2.contigs["gap"][n0_:]) if len(r2.contigs) > n0_ else np.zeros(0, dtype=np.int64)
                pick_ = sorted(int(x) for x in (with_m if len(with_m) <= 24 else rng.choice(with_m, 24, replace=False)))
                without = [int(g_) for g_ in rng.choice(n_gaps, min(n_gaps, 64), replace=False) if g_ not in set(with_m.tolist())][:8]
                with ThreadPoolExecutor(max_workers=max(1, (os.cpu_count() or 2))) as ex:
                    oks = list(ex.map(lambda g_: SC.merged_equal(r2.contigs, r2.seq, [g_], n0_, kk), pick_ + without))
                parity = {"gaps_checked": len(oks), "merged_contigs_equal_the_oracles": bool(all(oks))}
            out["contig_merge_round_all_gaps"] = dict(r2.merge, inside_the_timed_step=False, ms_of_one_step_with_the_round=tm * 1e3, parity=parity,
                                                      ms_of_the_round=tm * 1e3 - step_s * 1e3, gaps_closed_with_it=r2.n_closed,
                                                      note="measured once behind the timed region: no library of this configuration spans a gap, "
                                                           "so merging the open gaps' contigs closes (next to) nothing here")
        except Exception as e:
            out["contig_merge_round_all_gaps"] = {"error": repr(e)[:300]}
        pipe.merge_in_step = False
    for g_ in ctxs:
        g_.close()
    return (out if rank == 0 else None), rank, world


def truth_check(cfg, gaps, flanks, ctg, seq, best, GapFill):
    """Every closed gap of this rank: the sequence the picker writes for the winning contig (gappadder_amd/pick_contigs.py on that one
    contig: same contig, strand and span as the device word, asserted) against the TRUE bases behind the planted gap
    (gf_synth_truth; the reference evaluates its fills against the true sequences too, validate_gap_seqs.py:5-75)."""
    from gappadder_amd.pick_contigs import pick_gap_sequence
    closed = correct = 0
    wrong, causes, wrong_gaps = [], {}, []
    for g in np.nonzero(best)[0]:
        b = int(best[g])
        a_len, span1, ci, rev = b >> 56, (b >> 32) & 0xFFFFFF, 0x7FFFFFFF - ((b >> 1) & 0x7FFFFFFF), b & 1
        c = ctg[ci]
        assert int(c["gap"]) == g
        contig = seq[int(c["seq_off"]):int(c["seq_off"]) + int(c["length"])].decode()
        r = pick_gap_sequence([("c", contig)], flanks[g][0], flanks[g][1], a_len)
        assert r is not None and len(r[1]) == span1 and (r[2] != contig) == bool(rev), (g, b)
        st, en, sc = int(gaps[g]["start"]), int(gaps[g]["end"]), int(gaps[g]["scaffold"])
        lo, hi = (st - 6, en + 5) if rev else (st - 5, en + 6)
        true = GapFill.synth_truth(cfg, sc, lo, hi - lo)
        closed += 1
        if r[1] == true:
            correct += 1
        else:
            cause = ("length %+d" % (len(r[1]) - len(true))) if len(r[1]) != len(true) else "substitutions"
            causes[cause] = causes.get(cause, 0) + 1
            wrong_gaps.append(int(g))
            if len(wrong) < 64:
                nd = sum(1 for x, y in zip(r[1], true) if x != y) if len(r[1]) == len(true) else None
                wrong.append({"gap": int(g), "anchor": a_len, "k": int(c["k"]), "picked_len": len(r[1]), "true_len": len(true), "mismatches": nd})
    return {"closed": closed, "correct": correct, "wrong": wrong, "causes": causes, "wrong_gaps": wrong_gaps}


def open_gap_census(cfg, gaps, ctg, seq, best, GapFill, max_gaps=256, W=25):
    """Why is a gap still open?  For (a sample of) the gaps without a pick: the true sequence of the gap and its flank ends
    (gf_synth_truth) is cut into W-mers, and every W-mer is looked up in the gap's contigs (either strand):
      no_contigs                 the assembly emitted nothing for the gap
      spanning_contig_unpicked   one contig holds every W-mer from the left anchor to the right anchor — the picker should have closed it
      anchor_differs             one contig holds every W-mer of the gap proper but not of a 30-base anchor: the contig's copy of the
                                 flank end differs from the draft's (what `bwa mem -T 30` tolerates and the exact anchors do not)
      coverage_hole              some true W-mers are in no contig at all (a stretch no k-mer survived min-count / error removal for)
      fragmented                 every true W-mer is in some contig, but no single contig carries them all (an unresolved branch)
    The reference has no such tool (its evaluation compares picked sequences, validate_gap_seqs.py:5-75); this is the census
    VERDICT r3 (next 7) asks for before an alignment-grade anchor is worth building."""
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    open_gaps = np.nonzero(best == 0)[0]
    n_open = len(open_gaps)
    if n_open > max_gaps:
        open_gaps = open_gaps[np.linspace(0, n_open - 1, max_gaps).astype(np.int64)]
    order = np.argsort(ctg["gap"], kind="stable")
    gsorted = ctg["gap"][order]
    out = {"no_contigs": 0, "spanning_contig_unpicked": 0, "anchor_differs": 0, "coverage_hole": 0, "fragmented": 0}
    hole_sizes = []
    fl = int(cfg["flank_len"][0])
    for g in open_gaps:
        st, en, sc = int(gaps[g]["start"]), int(gaps[g]["end"]), int(gaps[g]["scaffold"])
        T = GapFill.synth_truth(cfg, sc, st - fl, en - st + 2 * fl).encode()
        lo, hi = np.searchsorted(gsorted, g), np.searchsorted(gsorted, g, side="right")
        sets = []
        for i in order[lo:hi]:
            c = seq[int(ctg[i]["seq_off"]):int(ctg[i]["seq_off"]) + int(ctg[i]["length"])]
            for s_ in (c, c.translate(comp)[::-1]):
                sets.append({s_[j:j + W] for j in range(len(s_) - W + 1)})
        if not sets:
            out["no_contigs"] += 1
            continue
        a0, a1 = fl - 5 - 30, fl + (en - st) + 5 + 30            # [left anchor start, right anchor end)
        i0, i1 = fl - 5, fl + (en - st) + 5                      # the gap proper + the 5 bases either side the flanks leave out
        ps = range(a0, a1 - W + 1)
        cover = [[T[p:p + W] in s_ for s_ in sets] for p in ps]
        if any(all(cover[j][c_] for j in range(len(ps))) for c_ in range(len(sets))):
            out["spanning_contig_unpicked"] += 1
        elif any(all(cover[j][c_] for j, p in enumerate(ps) if i0 <= p and p + W <= i1) for c_ in range(len(sets))):
            out["anchor_differs"] += 1
        elif not all(any(row) for row in cover):
            out["coverage_hole"] += 1
            hole_sizes.append(sum(1 for row in cover if not any(row)))
        else:
            out["fragmented"] += 1
    out.update(open_gaps=int(n_open), classified=int(len(open_gaps)), word=W,
               median_uncovered_words_in_a_hole=(int(np.median(hole_sizes)) if hole_sizes else None))
    return out


def e2e_files(config="C3"):
    """The product's own path, end to end, on FILES: tools/synth_files writes the configuration's draft FASTA, coordinate-sorted BAM
    and FASTQ pair (the same reads and records the step above takes from HBM), then the reference's CLI surface
    (`python -m gappadder_amd.main -c All -g cfg.json`, software_path.samtools = "builtin") runs on them in a child process: BAM and
    FASTQ are read once into HBM, gappadder_amd/pipeline.py recruits / pools / assembles / picks, the reference's working folder is
    written from the results, and the reference's later rounds (contig merging, both-unmapped recruitment, second assembly round,
    picks at 30 and 15, extended fills: assemble_gaps.py:328-368) follow.  Wall time of the whole CLI run, with its split."""
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth_files_util as SF
    seed, slen, nscf, gps, glen, dreads, kk = PRESETS[config]
    root = tempfile.mkdtemp(prefix="gf_e2e_", dir=os.environ.get("GF_E2E_DIR") or None)
    try:
        t0 = time.perf_counter()
        # (parameters.kmer_screen = the smallest k: the CLI recruits like the step above — alignment tagger + second hop + flank-k-mer screen)
        cfgp, wf = SF.write_case(root, seed, slen, nscf, gps, glen, [(300, 30, dreads // 2)], kk, kmer_screen=min(a for a, _ in kk),
                                 nthreads=max(1, (os.cpu_count() or 2) // 2))
        t_gen = time.perf_counter() - t0
        sizes = {fn: os.path.getsize(os.path.join(root, "data", fn)) for fn in sorted(os.listdir(os.path.join(root, "data")))}
        tfile = os.path.join(root, "timings.json")
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, "-m", "gappadder_amd.main", "-c", "All", "-g", cfgp], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=1500, env=dict(os.environ, GF_TIMINGS=tfile))
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": "CLI exit code %d" % r.returncode, "stderr_tail": r.stderr.decode()[-600:]}
        t = json.load(open(tfile))
        n_picked = open(wf + "picked_seqs.fa").read().count(">") if os.path.exists(wf + "picked_seqs.fa") else 0
        device_s = sum(t.get("seconds", {}).values())
        # the same configuration synthesised straight into HBM (a child run of this file): the CLI on files must have recruited exactly that
        dev = child_run(["--config", config, "--steps", "1", "--warmup", "0"])
        same = None
        if "counts" in dev and t.get("libraries"):
            a, b = list(dev["counts"]["libraries"].values())[0], t["libraries"][0]
            same = all(int(a[key]) == int(b[key]) for key in ("screen_hits", "tagger_hits", "second_hop_hits", "pool_keys", "pooled_reads"))
        return {"same_recruits_as_the_device_resident_run": same, "what": "python -m gappadder_amd.main -c All on the %s files (draft FASTA %.1f MB, BAM %.1f MB, FASTQ 2 x %.1f MB), "
                        "software_path.samtools = builtin; wall time of the child process, interpreter start-up and imports included"
                        % (config, sizes.get("draft.fa", 0) / 1e6, sizes.get("lib0.bam", 0) / 1e6, sizes.get("lib0_1.fq", 0) / 1e6),
                "reads": dreads, "wall_s": wall, "reads_per_s_end_to_end": dreads / wall,
                "stages_s": t.get("stages_s"), "device_collect_s": t.get("seconds"),
                "reads_per_s_collect_and_first_assembly": dreads / device_s if device_s else None,
                "libraries": t.get("libraries"), "gaps": t.get("gaps"), "gaps_closed_on_device": t.get("gaps_closed_on_device"),
                "picked_seqs": n_picked, "assembly_rounds": t.get("assembly"),
                "file_generation_s": t_gen,
                "reference_container_only": "BASELINE.md §2: the reference's own `main.py -c Collect` (2to3-converted, one scaffold) measured "
                                            "1.3e5 reads/s in the build container; informational, not measured on this box"}
    except Exception as e:      # the headline line must not depend on an extra
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def child_run(argv):
    """The same step on another BASELINE.json configuration, measured by the same code in a child process (never an exec of a
    process that has touched the GPU)."""
    r = None
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-extras", "--no-cpu"] + argv, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=900, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
        d = json.loads(r.stdout.decode().strip().splitlines()[-1])
        return {key: d[key] for key in ("value", "ms_per_step", "steps", "warmup", "gaps_per_s", "gaps_closed_per_s", "gaps_closed_correct_per_s",
                                        "phases_ms", "counts", "closed_truth_check", "assembly")} | \
               ({"open_gap_census": d["open_gap_census"]} if "open_gap_census" in d else {}) | \
               ({"contig_merge_round": d["contig_merge_round"]} if "contig_merge_round" in d else {}) | \
               ({"contig_merge_round_all_gaps": d["contig_merge_round_all_gaps"]} if "contig_merge_round_all_gaps" in d else {}) | \
               {"workload": d["config"]["workload"], "roofline_frac": d["roofline"]["frac"], "filter_ms": d["roofline"]["avg_launch_ms"]}
    except Exception as e:      # the headline line must not depend on an extra
        return {"error": repr(e)[:300], "stderr_tail": (r.stderr.decode()[-400:] if r is not None else "")}


def pmc_traffic(config, reads_per_launch, L, k, launch_ms, launched):
    """Bytes per launch of the dominant kernel group from the committed PMC passes (profiles/rNN_traffic_<config>.json, newest round
    first: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate runs of this same command, gfx950 x2 correction applied to FETCH_SIZE).
    Counters cannot be collected from inside the timed run, so the file must be shown to describe THIS build: same workload, the filter's
    kernels under exactly the names this build launched (gf_screen_kernels), and their rocprof launch times within 15 % of the launch
    time measured in this run —
    otherwise null (a kernel change without a re-profile must not leave a stale ratio in the line)."""
    import glob
    want = [w for w in launched.split(",") if w]      # gf_screen_kernels: what this build launched for the filter, template arguments included
    if not want:
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_%s.json" % config.lower())), reverse=True):
        try:
            t = json.load(open(path))
            names = list(t["kernels"])
            if t["reads_per_launch"] != reads_per_launch or t["read_len"] != L or t["k"] != k:
                continue
            if not all(any(w in n for n in names) for w in want) or not all(any(w in n for w in want) for n in names):
                continue
            if abs(t["rocprof_avg_launch_ns_sum"] * 1e-6 - launch_ms) > 0.15 * launch_ms:
                continue
            return t["traffic_bytes_per_launch"]
        except Exception:
            pass
    return None


def cpu_baseline(args, libs, flanks, gaps, L, kk, pool_t, off_t, ctg, d_seq, n_seq, d_best, gpu_step_s, n_screened, B, rb, merge_n0=None):
    """The oracle (oracle/gp_oracle.c, OpenMP over all host cores; kind "port") on a bounded sample of the same step:
    k-mer screen + alignment tagger on --cpu-sample-reads reads of the first library (a quarter of that of every further one) taken in
    stripes over the whole library, and the assembly of --cpu-sample-gaps gaps' pools, drawn over the whole gap list, at every (k, kv).
    Also the checker: the GPU's hits in those stripes, its contigs for those gaps and its closed flags must equal the oracle's / the
    host picker's (`parity_sample`: "striped", or "complete" when the sample is everything)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle as CO
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sample_check as SC      # the comparisons themselves live with the tests (tests/test_sample_check.py plants wrong hits, bases and picks)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:   # a cgroup CPU quota (cpu.max "quota period") caps the usable cores below the visible ones
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(q) // int(per)))
    except Exception:
        pass
    CO.set_threads(cores)
    k_s = min(a for a, _ in kk)
    ok, ok_tag, t_rec, n_rec, n_ohits, notes, where = True, True, 0.0, 0, 0, [], {}
    t_build = 0.0      # the oracle's flank k-mer table (once per run, not per read): timed INSIDE the screen call that builds it
    for li, lb in enumerate(libs):
        # The sample lies in STRIPES over the whole library (tests/sample_check.py::stripes): its first and last pairs, a stripe across
        # every place where a byte offset of the packed reads (rb per read), the 32-byte records or their 8-byte keys passes 4 GiB, one
        # across every shard boundary of a 2 / 4 / 8-rank run, and seeded places between — or ALL reads (--cpu-sample-reads -1).
        # First library: --cpu-sample-reads; further libraries: a quarter of that
        want = lb.n_reads if args.cpu_sample_reads < 0 else min(args.cpu_sample_reads if li == 0 else args.cpu_sample_reads // 4, lb.n_reads)
        pair_ranges = SC.stripes(lb.n_reads // 2, want // 2, strides=(2 * rb, 64, 16), seed=20260600 + li)
        read_ranges = [(2 * a, 2 * n) for a, n in pair_ranges]
        ocfg = np.frombuffer(lb.cfg.tobytes(), dtype=CO.SYNTH_CFG).copy()
        n_hits, n_th = int(lb.d_cnt[0]), int(lb.d_cnt[4])
        hits = np.frombuffer(lb.d_hits[:n_hits * 8].cpu().numpy().tobytes(), dtype=B.HIT)
        th = np.frombuffer(lb.d_thits[:n_th * 12].cpu().numpy().tobytes(), dtype=B.TAGHIT)
        n_s = t_scr = t_tag = 0
        n_oh = n_ot = 0
        for group in SC.chunks_of(pair_ranges, 8_000_000):       # one oracle call per group of stripes (one flank-table build each)
            parts = [CO.synth_pairs(ocfg, lb.first_pair + a, n) for a, n in group]
            packed, recs = np.concatenate([p_ for p_, _ in parts]), np.concatenate([r_ for _, r_ in parts])
            del parts
            blob = CO.unpack_reads(packed, L)
            t0 = time.perf_counter()
            ohits = CO.screen_reads(blob, L, flanks, k_s, 1, 0, cores)
            t1 = time.perf_counter()
            otags = CO.tag_alignments(recs, gaps, lb.is_mean, lb.is_sd)
            t2 = time.perf_counter()
            grp_reads = [(2 * a, 2 * n) for a, n in group]
            ok = SC.hits_equal(hits, ohits, grp_reads) and ok
            ok_tag = SC.taghits_equal(th, otags, grp_reads) and ok_tag
            t_build = CO.screen_last_build_s()          # (of this very call: no subtraction of a separately measured build)
            t_scr += max(1e-6, (t1 - t0) - t_build)
            t_tag += t2 - t1
            n_s += len(recs)
            n_oh += len(ohits)
            n_ot += len(otags)
            del blob, packed, recs
        t_rec += t_scr + t_tag
        n_rec += n_s
        n_ohits += n_oh
        complete = n_s == lb.n_reads
        where[lb.name] = {"reads": n_s, "of": lb.n_reads, "complete": complete, "stripes": len(read_ranges),
                          "read_ranges": [[a, a + n] for a, n in read_ranges]}
        notes.append("%s: %s (k-mer screen %.2f s + alignment tagger %.2f s; %d + %d hits)" %
                     (lb.name, "ALL %d reads" % n_s if complete else "%d reads in %d stripes over the whole library" % (n_s, len(read_ranges)),
                      t_scr, t_tag, n_oh, n_ot))
    ok = ok and ok_tag
    # assembly + pick sample: gaps drawn over the WHOLE gap list (first, last, one seeded gap in every part between; or all of them),
    # their pools exactly as the GPU assembled them (all libraries merged)
    gsel = SC.sample_gaps(len(gaps), len(gaps) if args.cpu_sample_gaps < 0 else args.cpu_sample_gaps, seed=20260610)
    n_g = len(gsel)
    pool_off = off_t.cpu().numpy()
    pblobs = [CO.unpack_reads(pool_t[int(pool_off[g]) * rb:int(pool_off[g + 1]) * rb].cpu().numpy().reshape(-1, rb), L) for g in gsel]
    seq = d_seq[:n_seq].cpu().numpy().tobytes()
    t3 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:   # gaps are independent (assemble_gaps.py:296-299 uses a process pool)
        exp = list(ex.map(lambda pb: [CO.assemble_pool(pb, L, k, kv, tiebreak=args.asm_tiebreak) for k, kv in kk], pblobs))
    t4 = time.perf_counter()
    best = d_best.cpu().numpy().view(np.uint64)
    ok_asm = SC.contigs_equal(ctg, seq, exp, kk, gsel)
    ok_pick = SC.picks_equal(ctg, seq, best, flanks, kk, gsel)
    # the merge round inside the step: the merged contigs of a seeded sample of the gaps it produced some for (and of sampled gaps it left
    # without any) against the oracle's merger on those gaps' own contigs; the picks over them are part of ok_pick above only for the
    # sampled gaps, so the gaps checked here go through the pick check too
    ok_merge, n_merge_checked = None, 0
    if merge_n0 is not None:
        has_merged = np.unique(ctg["gap"][merge_n0:]) if len(ctg) > merge_n0 else np.zeros(0, dtype=np.int64)
        rng = np.random.RandomState(20260620)
        pick_m = sorted(int(x) for x in (has_merged if len(has_merged) <= 32 else rng.choice(has_merged, 32, replace=False)))
        open_first = [g for g in gsel if int(best[g]) == 0][:8]          # sampled gaps that stayed open: the round must have appended exactly what the oracle says (often nothing)
        tmg = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:
            oks = list(ex.map(lambda g: SC.merged_equal(ctg, seq, [g], merge_n0, kk) and SC.picks_equal(ctg, seq, best, flanks, kk, [g]), pick_m + open_first))
        ok_merge, n_merge_checked = all(oks), len(oks)
        notes.append("merge round: %d gaps (%.2f s)" % (n_merge_checked, time.perf_counter() - tmg))
    striped = not all(w["complete"] for w in where.values()) or n_g < len(gaps)
    # whole-step CPU time extrapolated from the two samples (recruit scales with reads, assembly with gaps).  The flank k-mer table is
    # built once per run on either side (the GPU's index build is outside the timed step too): reported beside, not charged per step
    cpu_step = t_rec * (n_screened / n_rec) + (t4 - t3) * (len(gaps) / n_g)
    return {"value": n_screened / cpu_step, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "flank k-mer table %.2f s (once per run, not part of a step: `table_build_s`); recruit, OpenMP %d threads — %s; assembly + pick: pools of %s at %s "
                      "(%.2f s, same thread count); value = reads / (sample times scaled to the whole step); oracle/gp_oracle.c"
                      % (t_build, cores, "; ".join(notes), "ALL %d gaps" % n_g if n_g == len(gaps) else "%d gaps drawn over all %d (first, last, one seeded gap per part)" % (n_g, len(gaps)),
                         ",".join("%d/%d" % p for p in kk), t4 - t3),
            "parity_sample": "striped" if striped else "complete",
            "parity_sample_ranges": dict(where, gaps={"n": n_g, "of": len(gaps), "first": gsel[0], "last": gsel[-1],
                                                      "scaffolds_touched": int(len(np.unique(gaps["scaffold"][gsel])))}),
            "recruit_reads_per_s": n_rec / t_rec, "table_build_s": t_build, "assembly_gaps_per_s": n_g / (t4 - t3),
            "parity_on_sample": bool(ok and ok_asm and ok_pick and ok_merge is not False), "parity_recruit": bool(ok), "parity_assembly": bool(ok_asm),
            "parity_pick": bool(ok_pick), "parity_merge_round": ok_merge, "merge_round_gaps_checked": n_merge_checked, "sample_hits": int(n_ohits), "sample_contigs": int(sum(len(e) for ee in exp for e in ee))}


if __name__ == "__main__":
    main()
