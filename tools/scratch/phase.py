"""print ms_per_step and one phase of a bench.py JSON line read from stdin: phase.py <label> <phase>"""
import json, sys
def find(d, key):
    if isinstance(d, dict):
        if key in d: return d[key]
        for v in d.values():
            r = find(v, key)
            if r is not None: return r
d = json.loads([l for l in sys.stdin if l.startswith("{")][-1])
print(sys.argv[1], round(d["ms_per_step"], 2), sys.argv[2], round(find(d, "phases_ms")[sys.argv[2]], 3))
