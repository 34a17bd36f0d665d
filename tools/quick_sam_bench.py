"""Scratch measurement: gf_sam_pack (host text -> device parse -> records back) vs the host decoder on synthetic SAM lines."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gappadder_amd.hip_api import GapFill
from gappadder_amd import sam_io
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.RandomState(1)
names = ["scf%d" % i for i in range(50)]
seq, qual = "ACGT" * 37 + "AC", "I" * 150
lines = ["r%d\t%d\t%s\t%d\t%d\t150M\t=\t%d\t%d\t%s\t%s\tNM:i:0\tAS:i:150" % (i, 99 if i & 1 else 147, names[i % 50], 1000 + i, 60 if i % 50 else 0,
                                                                        1300 + i, 450, seq, qual) for i in range(n)]
text = ("\n".join(lines) + "\n").encode()
gf = GapFill(0)
gf.sam_pack(text[:100000].rsplit(b"\n", 1)[0] + b"\n", names)
t = time.time(); recs, lb = gf.sam_pack(text, names); dt = time.time() - t
print("gf_sam_pack: %d lines, %.1f MB: %.3f s = %.2e lines/s (%.2f GB/s incl. H2D and the Python wrapper)" % (n, len(text) / 1e6, dt, n / dt, len(text) / dt / 1e9))
m = min(n, 200000)
t = time.time(); exp, _ = sam_io.decode(lines[:m], {nm: i for i, nm in enumerate(names)}); dh = time.time() - t
print("host decoder (CPython): %.2e lines/s" % (m / dh))
assert recs[:m].tobytes() == exp.tobytes()
ms, nl = gf.kernel_time(8) if False else (0, 0)
