"""Diagnostic: per-gap phase times of the assembly kernel (wall_clock64 stamps, 100 MHz) on the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gappadder_amd import _lib as B
from gappadder_amd.hip_api import GapFill
from oracle import c_oracle as CO

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
cfg = GapFill.synth_cfg(scaffold_len=400_000, n_scaffolds=50, gaps_per_scaffold=20 // 10, gap_len=2000)
gaps, flanks = GapFill.synth_layout(cfg)
gf = GapFill(0)
gf.set_gaps(gaps, 50, flanks)
ocfg = np.frombuffer(cfg.tobytes(), dtype=CO.SYNTH_CFG).copy()
packed, recs = CO.synth_pairs(ocfg, 0, n_pairs)
hits = gf.screen_reads(packed, 150, 31)
ids = {}
for h in hits:
    ids.setdefault(int(h["gap"]), set()).update((int(h["read"]), int(h["read"]) ^ 1))
for h in gf.tag_alignments(recs, 300, 30):
    ids.setdefault(int(h["gap"]), set()).add(int(recs[h["rec"]]["read"]) ^ int(h["to_mate"]))
order = [sorted(ids.get(g, [])) for g in range(len(gaps))]
off = np.cumsum([0] + [len(o) for o in order]).astype(np.uint64)
pool = packed[np.concatenate([np.array(o, dtype=np.int64) for o in order if o])]
print("gaps", len(gaps), "pooled reads", len(pool), "mean/gap", len(pool) / len(gaps))
dev = torch.device("cuda:0")
dbg = torch.zeros(len(gaps) * 16, dtype=torch.int64, device=dev)
gf.set_option("asm_dbg_ptr", dbg.data_ptr())
gf.timing(True)
if os.environ.get("PRECOUNT"):
    gf.set_option("asm_precount", int(os.environ["PRECOUNT"]))
if os.environ.get("SIMPLIFY"):
    gf.set_option("asm_simplify", int(os.environ["SIMPLIFY"]))
for _ in range(3):
    K = int(os.environ.get("K", "31")); ctg, seq = gf.assemble(pool, off, 150, [(K, K - 2)])
ms, n = gf.kernel_time(B.KERNEL_ASSEMBLE)
print("assemble kernel %.3f ms avg, contigs %d" % (ms / n, len(ctg)))
d = dbg.cpu().numpy().reshape(-1, 16)
order = [0, 1, 2, 3, 4, 7, 5, 6]        # stamp 7 sits between 4 (links) and 5 (emission start)
ph = np.diff(d[:, order], axis=1) / 100.0  # us
names = ["P1 count", "P2 survivors", "P3 graph+index", "P4 links", "error removal", "ranking", "emission"]
for i, nm in enumerate(names):
    print("%-14s mean %8.1f us   max %8.1f us" % (nm, ph[:, i].mean(), ph[:, i].max()))
if d[:, 8].any():
    print("P1 split: pre-count %.1f us, table init %.1f us, exact count %.1f us" % (((d[:, 8] - d[:, 0]) / 100.0).mean(), ((d[:, 9] - d[:, 8]) / 100.0).mean(), ((d[:, 1] - d[:, 9]) / 100.0).mean()))
print("windows %.0f, distinct counted %.0f, survivors %.0f, global table in %.0f %% of the gaps" % (d[:, 13].mean(), d[:, 10].mean(), d[:, 12].mean(), 100.0 * d[:, 11].mean()))
print("nodes %.0f, graph plan (0 LDS, 1 LDS + global pairs, 2 global) mean %.2f" % (d[:, 14].mean(), d[:, 15].mean()))
print("total/gap mean %.1f us max %.1f us; kernel span %.1f us" % (ph.sum(1).mean(), ph.sum(1).max(), (d[:, 6].max() - d[:, 0].min()) / 100.0))
