#!/bin/bash
# usage (on the GPU box, through gpurun): tools/collect_profiles.sh <round, e.g. r04> [configs, default "c4 c2 c5"]
# rocprofv3 kernel trace + the HBM-traffic / SQ / TCC PMC passes (one counter group per pass, no sys-trace domains) of bench.py at
# C4 (default), C2 and C5, summarised ON THE BOX (the rocpd databases are beyond the 64 MiB that gpurun copies back) into
# gpurun_out/prof_<round>_summ/ — copy those files into profiles/.
# Every rocprofv3 call is bounded: a counter group the hardware cannot collect makes it abort and then hang in its finaliser.
RND=${1:?round}; CFGS=${2:-"c4 c2 c5"}
R=${GRAFT_REPO_ROOT:-/root/repo}
P=/tmp/prof_$RND; S=$R/gpurun_out/prof_${RND}_summ
mkdir -p $P $S; cd /tmp; export TMPDIR=/tmp
run() {  # cfg reads k steps  extra-bench-args...
  cfg=$1; reads=$2; k=$3; steps=$4; shift 4
  cmd="python3 bench.py --steps $steps --warmup 1 --no-extras --no-cpu $*"
  cd $R
  timeout 600 rocprofv3 --kernel-trace --stats -d $P/$cfg/trace -- python3 bench.py --steps $steps --warmup 1 --no-extras --no-cpu "$@" 2> $P/$cfg.trace.err < /dev/null | tail -1 > $S/${RND}_bench_${cfg}_under_rocprof.json
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace -d $P/$cfg/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu "$@" > $P/$cfg.$c.log 2>&1 < /dev/null
  done
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $P/$cfg/pmc_SQ -- python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu "$@" > $P/$cfg.SQ.log 2>&1 < /dev/null
  timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d $P/$cfg/pmc_TCC -- python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu "$@" > $P/$cfg.TCC.log 2>&1 < /dev/null
  python3 tools/summarise_profiles.py $RND $P/$cfg $cfg $reads $k "$cmd" $S > $S/$cfg.summary.log 2>&1
  tail -3 $S/$cfg.summary.log
}
for c in $CFGS; do
  case $c in
    c4) run c4 900000000 51 5 ;;
    c2) run c2 50000000 31 20 --config C2 ;;
    c5) run c5 650000000 31 2 --config C5 ;;
  esac
done
if [[ " $CFGS " == *" c4 "* ]]; then
  cd $R && timeout 900 python3 bench.py --steps 20 --warmup 5 > $S/${RND}_bench_default_n1.json 2> $S/${RND}_bench_default_n1.err < /dev/null
fi
ls -la $S
