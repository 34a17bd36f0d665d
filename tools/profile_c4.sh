#!/bin/bash
# Scratch profile of one GPU's shard of the human-scale run (bench.py --config C4) on the GPU box: kernel stats with every phase
# on one stream (stand-alone kernel times), then two PMC groups for the filter kernels.  Usage: gpurun -- 'bash tools/profile_c4.sh'
# Every rocprofv3 call is bounded (a counter group the hardware cannot collect aborts and then hangs in the finaliser).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_c4
rm -rf $O; mkdir -p $O
export GF_BENCH_SERIAL=1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --config C4 --steps 3 --warmup 1 --no-cpu > $O/trace.log 2>&1
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/pmc_tcc -- python3 $R/bench.py --config C4 --steps 2 --warmup 1 --no-cpu > $O/pmc_tcc.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --config C4 --steps 2 --warmup 1 --no-cpu > $O/pmc_sq.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_c4"
f = glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pf_" in r["Kernel_Name"] or "screen_filter" in r["Kernel_Name"] or "assemble" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(k, c, len(v), "%.4g" % (sum(v) / len(v)))
PY
