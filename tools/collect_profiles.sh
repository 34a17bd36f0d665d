#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + the HBM-traffic PMC passes of the default bench command (C2), the
# same for one GPU's shard of the human-scale run (C4), and kernel stats of the BAM ingest microbenchmark.
# Outputs under gpurun_out/prof_final/; tools/summarise_profiles.py turns them into profiles/r01_*.
# Every rocprofv3 call is bounded: a counter group the hardware cannot collect makes it abort and then hang in its finaliser.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_final
rm -rf $O; mkdir -p $O
B="$R/bench.py --no-cpu"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $B --steps 10 --warmup 2 > $O/trace.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $B --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $B --steps 3 --warmup 1 > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/pmc_tcc -- python3 $B --steps 3 --warmup 1 > $O/pmc_tcc.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $B --steps 3 --warmup 1 > $O/pmc_sq.log 2>&1
# human-scale shard (C4)
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_trace -- python3 $B --config C4 --steps 5 --warmup 2 > $O/c4_trace.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c4_pmc_fetch -- python3 $B --config C4 --steps 2 --warmup 1 > $O/c4_pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c4_pmc_write -- python3 $B --config C4 --steps 2 --warmup 1 > $O/c4_pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/c4_pmc_tcc -- python3 $B --config C4 --steps 2 --warmup 1 > $O/c4_pmc_tcc.log 2>&1
# BAM ingest
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bam_trace -- python3 $R/tools/quick_bam_bench.py 200000 24 > $O/bam_trace.log 2>&1
tail -1 $O/trace.log | head -c 400; echo
tail -1 $O/c4_trace.log | head -c 400; echo
tail -3 $O/bam_trace.log
ls $O
