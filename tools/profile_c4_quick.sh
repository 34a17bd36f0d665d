cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_c4q
rm -rf $O; mkdir -p $O
GF_BENCH_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --config C4 --steps 3 --warmup 1 --no-cpu > $O/trace.log 2>&1
python3 - <<'PY'
import csv,glob,os
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/prof_c4q"
f=glob.glob(O+"/trace/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]: print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
PY
