#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + the two HBM-traffic PMC passes of the default bench command.
# Outputs under gpurun_out/prof_final/; tools/summarise_profiles.py turns them into profiles/r01_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_final
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $O/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/pmc_tcc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $O/pmc_tcc.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $O/pmc_sq.log 2>&1
tail -1 $O/trace.log | head -c 600
ls $O
