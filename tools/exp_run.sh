#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_assembly.py tests/test_gpu_scale.py tests/test_gpu_parity.py -x -q < /dev/null 2>&1 | tail -3
GF_DIAGNOSTICS=1 GF_BENCH_ASM_PROBE=1 timeout 600 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu < /dev/null 2>/tmp/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['phases_ms'])"; grep "assembly phases" /tmp/err.txt
