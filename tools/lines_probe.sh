#!/bin/bash
# scratch: parity of the filter variants, then the default bench with pass A as whole lines / as unaligned runs
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "variants or partitioned or synthetic or human_scale" < /dev/null > gpurun_out/lines_tests.log 2>&1
tail -5 gpurun_out/lines_tests.log
for v in 0 17 0; do
  GF_BENCH_SCREEN_VARIANT=$v timeout 600 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu < /dev/null > gpurun_out/lines_bench_$v.json 2> gpurun_out/lines_bench_$v.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/lines_bench_$v.json").read().strip().splitlines()[-1])
    print("variant $v", d["ms_per_step"], d["roofline"]["frac"], d.get("phases_ms"))
except Exception as e:
    print("variant $v failed", e); print(open("gpurun_out/lines_bench_$v.err").read()[-1500:])
PY
done
