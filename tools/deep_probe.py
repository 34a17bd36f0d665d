import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gappadder_amd import _lib as B
from gappadder_amd.hip_api import GapFill
from oracle import c_oracle as CO
n_pairs = int(sys.argv[1])
cfg = GapFill.synth_cfg(scaffold_len=400_000, n_scaffolds=50, gaps_per_scaffold=2, gap_len=2000)
gaps, flanks = GapFill.synth_layout(cfg)
gf = GapFill(0)
gf.set_gaps(gaps, 50, flanks)
ocfg = np.frombuffer(cfg.tobytes(), dtype=CO.SYNTH_CFG).copy()
packed, recs = CO.synth_pairs(ocfg, 0, n_pairs)
hits = gf.screen_reads(packed, 150, 31)
ids = {}
for h in hits:
    ids.setdefault(int(h["gap"]), set()).update((int(h["read"]), int(h["read"]) ^ 1))
order = [sorted(ids.get(g, [])) for g in range(len(gaps))]
gf.set_option("asm_keyslot", int(os.environ.get("KS", "1")))
g0 = int(os.environ.get("G0", "0"))
order = order[g0:]
for ng in [int(x) for x in os.environ.get("NG", "1,5,100").split(",")]:
    o2 = order[:ng]
    off = np.cumsum([0] + [len(o) for o in o2]).astype(np.uint64)
    pool = packed[np.concatenate([np.array(o, dtype=np.int64) for o in o2 if o])]
    for mc in [int(x) for x in os.environ.get("MC", "2,4").split(",")]:
        try:
            ctg, seq = gf.assemble(pool, off, 150, [(31, 29)], min_count=mc)
            print("gaps", ng, "reads", len(pool), "min_count", mc, "contigs", len(ctg))
        except Exception as e:
            print("gaps", ng, "reads", len(pool), "min_count", mc, "ERR", e)
