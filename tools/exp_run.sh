#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/exp; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "variants or partitioned or synthetic or human_scale" < /dev/null 2>&1 | tail -3
for lib in - exp/libgf_e1.so; do
    n=$(basename $lib .so)
    timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/exp/$n -o r -- python3 tools/exp_filter.py $lib 112500000 51 0 < /dev/null 2>&1 | grep "filter"
    timeout 60 python3 tools/rocpd_top.py gpurun_out/exp/$n pf4 < /dev/null 2>&1 | head -5
done
for v in 0; do
timeout 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/exp/w$v -o r -- python3 tools/exp_filter.py - 112500000 51 $v < /dev/null 2>&1 | grep "filter"
python3 - <<PY
import glob, sqlite3, collections
db = sqlite3.connect(glob.glob("gpurun_out/exp/w$v/**/*_results.db", recursive=True)[0])
st = collections.defaultdict(list)
for kn, cn, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
    if "pf4" in kn: st[(kn[:60], cn)].append(v)
for k, v in st.items(): print(k, sum(v) / len(v) * 1024 / 1e9, "GB")
PY
done
rm -rf gpurun_out/exp
