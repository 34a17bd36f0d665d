"""rocprofv3 results (rocpd SQLite: <src>/trace, <src>/pmc_<COUNTER>) -> profiles/<round>_kernel_stats_<cfg>.csv,
<round>_pmc_summary_<cfg>.csv, <round>_traffic_<cfg>.json (the file bench.py's roofline.traffic reads; it carries the kernels' names
and their rocprof launch times, which bench.py checks against its own run before quoting the traffic).
usage: summarise_profiles.py ROUND SRC CFG READS_PER_LAUNCH K "CMD" [DST=profiles]      (ROUND = r04, ...) """
import collections, csv, glob, json, os, sqlite3, sys

rnd, src, cfg, reads, k, cmd = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
dst = sys.argv[7] if len(sys.argv) > 7 else "profiles"
os.makedirs(dst, exist_ok=True)


def newest_db(d):
    return max(glob.glob(d + "/**/*_results.db", recursive=True), key=os.path.getmtime)


def short(n):      # rocPRIM's template names run to kilobytes
    return n if len(n) < 300 else n[:140] + " ... " + n[-100:]


db = sqlite3.connect(newest_db(src + "/trace"))
stat = collections.defaultdict(list)
for name, dur in db.execute("select name, duration from kernels"):
    stat[name].append(dur)
total = sum(sum(v) for v in stat.values())
rows = sorted(((n, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / total, min(v), max(v)) for n, v in stat.items()), key=lambda r: -r[2])
with open("%s/%s_kernel_stats_%s.csv" % (dst, rnd, cfg), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- %s   (MI355X; durations in ns, every dispatch of the run incl. the untimed sizing pass)\n" % cmd)
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([short(r[0]), r[1], r[2], "%.1f" % r[3], "%.2f" % r[4], r[5], r[6]])
avg = {r[0]: r[3] for r in rows}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(src + "/pmc_*/")):
    c = sqlite3.connect(newest_db(d))
    for kn, cn, v in c.execute("select kernel_name, counter_name, value from counters_collection"):
        if "gf::" in kn:
            acc[kn][cn].append(float(v))
with open("%s/%s_pmc_summary_%s.csv" % (dst, rnd, cfg), "w") as f:
    f.write("# rocprofv3 --pmc <one counter group per pass> -- %s; mean counter value per dispatch\n" % cmd)
    w = csv.writer(f)
    w.writerow(["Kernel", "Counter", "Dispatches", "MeanPerDispatch"])
    for kn in sorted(acc):
        for cn in sorted(acc[kn]):
            v = acc[kn][cn]
            w.writerow([short(kn), cn, len(v), "%.6g" % (sum(v) / len(v))])
# the dominant kernel group = the screen filter: pf4_scatter + pf4_probe + pf4_resolve + pf4_list (256 buckets, 4-byte pairs) or screen_filter_* (one kernel)
kern, tot_traffic, tot_ns = {}, 0.0, 0.0
for n in acc:
    if not any(t in n for t in ("pf4_scatter", "pf4_probe", "pf4_resolve", "pf4_list", "screen_filter")) or "FETCH_SIZE" not in acc[n] or "WRITE_SIZE" not in acc[n]:
        continue
    fch, wr = (sum(acc[n][c]) / len(acc[n][c]) for c in ("FETCH_SIZE", "WRITE_SIZE"))
    kern[n] = {"FETCH_SIZE_kb": fch, "WRITE_SIZE_kb": wr, "traffic_bytes_per_launch": (2 * fch + wr) * 1024.0, "rocprof_avg_launch_ns": avg.get(n)}
    tot_traffic += (2 * fch + wr) * 1024.0
    tot_ns += avg.get(n) or 0.0
out = {"workload": cfg, "reads_per_launch": reads, "read_len": 150, "k": k, "algorithmic_bytes_per_launch": reads * 38,
       "kernels": kern, "traffic_bytes_per_launch": tot_traffic, "rocprof_avg_launch_ns_sum": tot_ns,
       "traffic_over_algorithmic": tot_traffic / (reads * 38.0),
       "correction": "gfx950: FETCH_SIZE counts 128-B fabric requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; Infinity-Cache "
                     "hits are included, so this is fabric traffic, an upper bound of HBM traffic",
       "cmd": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- " + cmd}
json.dump(out, open("%s/%s_traffic_%s.json" % (dst, rnd, cfg), "w"), indent=1)
print(json.dumps(out, indent=1))
