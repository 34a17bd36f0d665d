"""mean counter values per dispatch from the newest rocpd database under DIR: rocpd_pmc.py DIR [kernel substring ...]"""
import collections, glob, os, sqlite3, sys
db = sqlite3.connect(max(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True), key=os.path.getmtime))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for kn, cn, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
    if len(sys.argv) > 2 and not any(s in kn for s in sys.argv[2:]):
        continue
    acc[kn][cn].append(float(v))
for kn in acc:
    print(kn[:100])
    print("   ", {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(acc[kn].items())})
