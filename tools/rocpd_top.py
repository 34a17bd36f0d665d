"""print the per-kernel averages of the newest rocprofv3 rocpd database under a directory: rocpd_top.py DIR [substring ...]"""
import collections, glob, os, sqlite3, sys
db = sqlite3.connect(max(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True), key=os.path.getmtime))
st = collections.defaultdict(list)
for n, d in db.execute("select name, duration from kernels"):
    st[n].append(d)
for n, v in sorted(st.items(), key=lambda kv: -sum(kv[1])):
    if len(sys.argv) > 2 and not any(s in n for s in sys.argv[2:]):
        continue
    print("%10.1f us avg  %10.1f min  %4d calls  %s" % (sum(v) / len(v) / 1e3, min(v) / 1e3, len(v), n[:110]))
