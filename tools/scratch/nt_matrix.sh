#!/bin/bash
# assembly time for several pool depths (C4 layout, k = 51; the survey-sized C5): automatic thread choice / 512 / 1024 per gap
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out/nt
for t in 0 ${NT_LIST:-}; do
  for reads in 900000000 1200000000 1500000000; do
    GF_BENCH_ASM_THREADS=$t timeout -k 5 300 python3 bench.py --config C4 --reads $reads --steps 3 --warmup 1 --no-cpu --no-extras < /dev/null 2>/dev/null | tail -1 > gpurun_out/nt/c4_${reads}_$t.json
  done
  GF_BENCH_ASM_THREADS=$t timeout -k 5 300 python3 bench.py --config C5 --mp-reads 100000000 --steps 3 --warmup 1 --no-cpu --no-extras < /dev/null 2>/dev/null | tail -1 > gpurun_out/nt/c5s_$t.json
  GF_BENCH_ASM_THREADS=$t timeout -k 5 300 python3 bench.py --config C5 --steps 3 --warmup 1 --no-cpu --no-extras < /dev/null 2>/dev/null | tail -1 > gpurun_out/nt/c5_$t.json
done
ls gpurun_out/nt | wc -l
