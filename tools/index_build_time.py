import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from gappadder_amd.hip_api import GapFill
for name, args in (("C2", dict(seed=20260002, scaffold_len=5_000_000, n_scaffolds=50, gaps_per_scaffold=20, gap_len=2000)),
                   ("C4", dict(seed=20260004, scaffold_len=5_000_000, n_scaffolds=620, gaps_per_scaffold=32, gap_len=2000))):
    cfg = GapFill.synth_cfg(**args)
    t = time.time(); gaps, flanks = GapFill.synth_layout(cfg); t_l = time.time() - t
    gf = GapFill(0)
    t = time.time(); gf.set_gaps(gaps, int(cfg["n_scaffolds"][0]), flanks); gf.sync(); t_s = time.time() - t
    packed = np.zeros((64, 38), dtype=np.uint8)
    for k in (31, 51):
        t = time.time(); gf.screen_reads(packed, 150, k); t_i = time.time() - t
        t = time.time(); gf.screen_reads(packed, 150, k); t_2 = time.time() - t
        print("%s: %d gaps; layout %.2f s; set_gaps %.3f s; first screen k=%d (index build) %.3f s; second %.4f s" % (name, len(gaps), t_l, t_s, k, t_i, t_2))
    gf.close()
