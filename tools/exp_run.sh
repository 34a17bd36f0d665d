#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q < /dev/null 2>&1 | tail -3
GF_DIAGNOSTICS=1 GF_BENCH_ASM_PROBE=1 timeout 600 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu < /dev/null 2>/tmp/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['phases_ms'])"; grep "assembly phases" /tmp/err.txt
timeout 900 python bench.py --steps 2 --warmup 1 --no-extras --no-cpu --config C5 < /dev/null 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['phases_ms'], d['counts']['gaps_closed'], d['closed_truth_check']['correct'])"
