#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/exp; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "variants or partitioned or synthetic or human_scale" < /dev/null 2>&1 | tail -3
for k in 51 31; do
    timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/exp/r$k -o r -- python3 tools/exp_filter.py - 112500000 $k 0 < /dev/null 2>&1 | grep "filter"
    timeout 60 python3 tools/rocpd_top.py gpurun_out/exp/r$k pf4 < /dev/null 2>&1 | head -4
done
rm -rf gpurun_out/exp
