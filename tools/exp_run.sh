#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/exp; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "variants or partitioned or synthetic or human_scale" < /dev/null 2>&1 | tail -3
for k in 51; do
    timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/exp/r$k -o r -- python3 tools/exp_filter.py - 112500000 $k 0 < /dev/null 2>&1 | grep "filter"
    timeout 60 python3 tools/rocpd_top.py gpurun_out/exp/r$k pf4 < /dev/null 2>&1 | head -4
done
pm() {
timeout 300 rocprofv3 --pmc "$@" -d gpurun_out/exp/p -o r -- python3 tools/exp_filter.py - 112500000 51 0 < /dev/null > /dev/null 2>&1
python3 - <<PY
import glob, sqlite3, collections
db = sqlite3.connect(glob.glob("gpurun_out/exp/p/**/*_results.db", recursive=True)[0])
st = collections.defaultdict(list)
for kn, cn, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
    if "pf4_scatter" in kn: st[cn].append(v)
for k, v in sorted(st.items()): print("%-28s %.4g" % (k, sum(v) / len(v)))
PY
rm -rf gpurun_out/exp/p
}
pm SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
rm -rf gpurun_out/exp
