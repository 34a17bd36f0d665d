"""gpurun_out/prof_final -> profiles/r01_kernel_stats.csv, profiles/r01_pmc_summary.csv, profiles/r01_traffic.json"""
import collections, csv, glob, json, os, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_final"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"
os.makedirs(dst, exist_ok=True)
newest = lambda pat: max(glob.glob(pat, recursive=True), key=os.path.getmtime)   # gpurun merges runs: keep the latest
stats = newest(src + "/trace/**/*kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
with open(dst + "/r01_kernel_stats.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu   (MI355X, C2 workload)\n")
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        if r["Name"].startswith(("gf::", "void gf::")):
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(src + "/pmc_*/")):
    f = newest(d + "**/*counter_collection.csv")
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith(("gf::", "void gf::")):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(dst + "/r01_pmc_summary.csv", "w") as f:
    f.write("# rocprofv3 --pmc <one group per pass> -- python3 bench.py --steps 3 --warmup 1 --no-cpu; mean counter value per dispatch\n")
    w = csv.writer(f)
    w.writerow(["Kernel", "Counter", "Dispatches", "MeanPerDispatch"])
    for k in sorted(acc):
        for c in sorted(acc[k]):
            v = acc[k][c]
            w.writerow([k, c, len(v), "%.6g" % (sum(v) / len(v))])
name = [r["Name"] for r in rows if "screen_filter" in r["Name"]][0]
fetch = sum(acc[name]["FETCH_SIZE"]) / len(acc[name]["FETCH_SIZE"])
write = sum(acc[name]["WRITE_SIZE"]) / len(acc[name]["WRITE_SIZE"])
avg_ns = [float(r["AverageNs"]) for r in rows if r["Name"] == name][0]
out = {"kernel": name, "reads_per_launch": 50000000, "read_len": 150, "k": 31,
       "FETCH_SIZE_kb_per_launch": fetch, "WRITE_SIZE_kb_per_launch": write,
       "traffic_bytes_per_launch": (2 * fetch + write) * 1024.0,
       "correction": "gfx950: FETCH_SIZE counts 128-B fabric requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; "
                     "Infinity-Cache hits are included, so this is fabric traffic, an upper bound of HBM traffic",
       "rocprof_avg_launch_ns": avg_ns,
       "cmd": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu"}
json.dump(out, open(dst + "/r01_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))


# ---- human-scale shard (C4) and BAM ingest, when collected
def kernel_stats(sub, out_name, header):
    fs = glob.glob(src + "/" + sub + "/**/*kernel_stats.csv", recursive=True)
    if not fs:
        return None
    rows = list(csv.DictReader(open(max(fs, key=os.path.getmtime))))
    with open(dst + "/" + out_name, "w") as f:
        f.write(header)
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            if r["Name"].startswith(("gf::", "void gf::")):
                w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    return rows


rows4 = kernel_stats("c4_trace", "r01_kernel_stats_c4.csv",
                     "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu --config C4 --steps 5 --warmup 2   (MI355X, one GPU's shard of the human-scale run)\n")
if rows4:
    acc4 = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(src + "/c4_pmc_*/")):
        for r in csv.DictReader(open(newest(d + "**/*counter_collection.csv"))):
            if r["Kernel_Name"].startswith(("gf::", "void gf::")):
                acc4[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(dst + "/r01_pmc_summary_c4.csv", "w") as f:
        f.write("# rocprofv3 --pmc <one group per pass> -- python3 bench.py --no-cpu --config C4 --steps 2 --warmup 1; mean counter value per dispatch\n")
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "Dispatches", "MeanPerDispatch"])
        for k in sorted(acc4):
            for c in sorted(acc4[k]):
                v = acc4[k][c]
                w.writerow([k, c, len(v), "%.6g" % (sum(v) / len(v))])
    pf = {}
    for k in acc4:
        if "pf_" in k and "FETCH_SIZE" in acc4[k] and "WRITE_SIZE" in acc4[k]:
            fch, wr = (sum(acc4[k][c]) / len(acc4[k][c]) for c in ("FETCH_SIZE", "WRITE_SIZE"))
            avg = [float(r["AverageNs"]) for r in rows4 if r["Name"] == k]
            pf[k] = {"FETCH_SIZE_kb": fch, "WRITE_SIZE_kb": wr, "traffic_bytes_per_launch": (2 * fch + wr) * 1024.0, "rocprof_avg_launch_ns": avg[0] if avg else None}
    json.dump({"workload": "C4 shard: 19 840 gaps, 112.5 M reads, k=51", "algorithmic_bytes_per_launch": 112500000 * 38, "kernels": pf,
               "correction": "FETCH_SIZE x2 on gfx950 (see r01_traffic.json)"}, open(dst + "/r01_traffic_c4.json", "w"), indent=1)
    print(json.dumps(pf, indent=1))
rowsb = kernel_stats("bam_trace", "r01_kernel_stats_bam_ingest.csv",
                     "# rocprofv3 --kernel-trace --stats -- python3 tools/quick_bam_bench.py 200000 24   (MI355X; 200 k records, the BGZF bytes x24 for the inflate timing)\n")
if rowsb and os.path.exists(src + "/bam_trace.log"):
    with open(dst + "/r01_kernel_stats_bam_ingest.csv", "a") as f:
        for line in open(src + "/bam_trace.log"):
            if line.startswith(("made", "gf_", "ingest", "zlib", "x")):
                f.write("# " + line)
