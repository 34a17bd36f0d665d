"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if not any(x in k for x in ("screen", "tag", "low_mapq", "assemble", "pool")):
        continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
