"""Phase times of the deepest pools of the repeat-bearing stress workload (C2R): the assembly kernel's wall-clock stamps (option asm_dbg_ptr,
100 MHz) for the gaps with the most reads.  usage (GPU box): python3 tools/microbench/deep_pool_phases.py [n_deepest]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gappadder_amd.hip_api import GapFill
from gappadder_amd.pipeline import DeviceLibrary, Pipeline

seed, slen, nscf, gps, glen, dreads, kk = 20260002, 5_000_000, 50, 20, 2000, 50_000_000, [(31, 29)]
L, rb = 150, 38
gf = GapFill(0)
cfg0 = GapFill.synth_cfg(seed=seed, scaffold_len=slen, n_scaffolds=nscf, gaps_per_scaffold=gps, gap_len=glen, read_len=L, insert_mean=300, insert_sd=30,
                         repeat_period=8, repeat_copies=50)
gaps, flanks = GapFill.synth_layout(cfg0)
gf.set_gaps(gaps, nscf, flanks)
pipe = Pipeline(gf, len(gaps), L, kk)
cfg = GapFill.synth_cfg(seed=seed, scaffold_len=slen, n_scaffolds=nscf, gaps_per_scaffold=gps, gap_len=glen, read_len=L, insert_mean=300, insert_sd=30,
                        library=0, repeat_period=8, repeat_copies=50)
n_pairs = dreads // 2
d_reads = torch.empty(2 * n_pairs * rb + 64, dtype=torch.uint8, device="cuda")
d_recs = torch.empty(2 * n_pairs * 32, dtype=torch.uint8, device="cuda")
gf.synth_pairs_dev(cfg, 0, n_pairs, d_reads.data_ptr(), d_recs.data_ptr())
pipe.add_library(DeviceLibrary("short", 300, 30, 2 * n_pairs, d_reads, d_recs))
gf.sync()
pipe.prepare()
pipe.step(2)
d_dbg = torch.zeros(len(gaps) * 16, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
gf.set_option("asm_dbg_ptr", d_dbg.data_ptr())
pipe.step(1)
pipe.barrier()
gf.set_option("asm_dbg_ptr", 0)
st = d_dbg.cpu().numpy().reshape(-1, 16)
off = pipe.d_moff.cpu().numpy() if pipe.need_merge else pipe.libs[0].d_pool_off.cpu().numpy()
n = np.diff(off)
names = {0: "start", 8: "pre-count bits", 9: "exact count", 1: "count phase end", 2: "survivors", 3: "graph", 4: "links", 7: "error removal", 5: "ranking", 6: "emission"}
order = [0, 8, 9, 1, 2, 3, 4, 7, 5, 6]
for g in np.argsort(-n)[:int(sys.argv[1]) if len(sys.argv) > 1 else 4]:
    row = st[g]
    t0, prev, parts = row[0], row[0], []
    for s in order[1:]:
        if row[s]:
            parts.append("%s %.0f us" % (names[s], (row[s] - prev) / 100.0))
            prev = row[s]
    print("gap %d: %d reads, %.0f us in all: %s" % (g, n[g], (prev - t0) / 100.0, ", ".join(parts)))
t_all = (st[:, order].max() - st[:, 0][st[:, 0] > 0].min()) / 100.0
print("first start to last stamp: %.0f us" % t_all)
