#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q < /dev/null 2>&1 | tail -3
bash tools/collect_profiles_r03.sh > /tmp/c.log 2>&1; tail -2 /tmp/c.log
