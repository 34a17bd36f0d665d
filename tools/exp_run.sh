#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q < /dev/null 2>&1 | tail -4
timeout 900 python bench.py < /dev/null > gpurun_out/bench_lines.json 2> gpurun_out/bench_lines.err; tail -c 3000 gpurun_out/bench_lines.json
