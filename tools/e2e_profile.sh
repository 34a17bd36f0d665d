#!/bin/bash
# usage (on the GPU box, through gpurun): tools/e2e_profile.sh <round>   — rocprofv3 kernel trace of the CLI (`-c All`, builtin BAM, k-mer screen on)
# on the C3-sized files of tools/synth_files; the per-kernel summary goes to gpurun_out/<round>_kernel_stats_e2e_c3.csv (copy it into profiles/).
RND=${1:?round}
R=${GRAFT_REPO_ROOT:-/root/repo}
W=/tmp/e2e_prof_$RND; rm -rf $W; mkdir -p $W
cd $R
python3 - <<PY
import sys
sys.path.insert(0, "tools")
import synth_files_util as SF
SF.write_case("$W", 20260003, 4600000, 1, 200, 1000, [(300, 30, 2500000)], [(41, 39)], kmer_screen=41, nthreads=8)
PY
cd /tmp; export TMPDIR=/tmp
cd $R
timeout 600 rocprofv3 --kernel-trace --stats -d $W/trace -- python3 -m gappadder_amd.main -c All -g $W/cfg.json > $W/run.log 2>&1 < /dev/null
python3 - <<PY
import collections, csv, glob, os, sqlite3
db = sqlite3.connect(max(glob.glob("$W/trace/**/*_results.db", recursive=True), key=os.path.getmtime))
stat = collections.defaultdict(list)
for name, dur in db.execute("select name, duration from kernels"):
    stat[name].append(dur)
total = sum(sum(v) for v in stat.values())
rows = sorted(((n, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / total) for n, v in stat.items()), key=lambda r: -r[2])
with open("$R/gpurun_out/${RND}_kernel_stats_e2e_c3.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 -m gappadder_amd.main -c All -g cfg.json   (C3-sized files: BAM 161 MB + FASTQ 2 x 789 MB, 5 M reads, builtin BAM, kmer_screen 41; MI355X; ns; all kernels of the run = %.1f ms)\n" % (total / 1e6))
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for r in rows[:60]:
        w.writerow([r[0] if len(r[0]) < 200 else r[0][:120] + " ... " + r[0][-60:], r[1], r[2], "%.1f" % r[3], "%.2f" % r[4]])
print("kernels total ms", total / 1e6)
PY
tail -2 $W/run.log
