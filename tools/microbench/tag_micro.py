"""micro-benchmark: tagger kernel time with / without the MAPQ-0 compaction, short and long inserts (C4 layout, 200 M records)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gappadder_amd.hip_api import GapFill
from gappadder_amd import _lib as B
lib = B.lib()
gf = GapFill(0)
cfg = GapFill.synth_cfg(seed=20260004, scaffold_len=5_000_000, n_scaffolds=620, gaps_per_scaffold=32, gap_len=2000)
gaps, flanks = GapFill.synth_layout(cfg)
gf.set_gaps(gaps, 620, flanks)
n_pairs = 100_000_000
dev = torch.device("cuda:0")
rb = lib.gf_packed_read_bytes(150)
d_reads = torch.empty(2 * n_pairs * rb + 64, dtype=torch.uint8, device=dev)
d_recs = torch.empty(2 * n_pairs * 32, dtype=torch.uint8, device=dev)
gf.synth_pairs_dev(cfg, 0, n_pairs, d_reads.data_ptr(), d_recs.data_ptr())
gf.sync()
del d_reads
n = 2 * n_pairs
cap = 1 << 24
d_out = torch.empty(cap * 16, dtype=torch.uint8, device=dev)
d_low = torch.empty(cap * 12, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(8, dtype=torch.int32, device=dev)
gf.timing(True)
for ins, sd in ((300, 30), (5000, 500)):
    for low in (0, 1):
        for it in range(4):
            if low:
                rc = lib.gf_tag_alignments_low_dev(gf.handle, d_recs.data_ptr(), n, ins, sd, 250, 30, d_out.data_ptr(), cap, d_cnt.data_ptr(),
                                                   d_low.data_ptr(), cap, d_cnt.data_ptr() + 4)
            else:
                rc = lib.gf_tag_alignments_dev(gf.handle, d_recs.data_ptr(), n, ins, sd, 250, 30, d_out.data_ptr(), cap, d_cnt.data_ptr())
            assert rc == 0, rc
            gf.sync()
            if it == 0:
                gf.timing(True)
        ms, k = gf.kernel_time(B.GF_KERNEL_TAG if hasattr(B, "GF_KERNEL_TAG") else 1)
        c = d_cnt.cpu().numpy()
        print(f"IS {ins}: low={low}  {ms / max(k, 1):.3f} ms per {n / 1e6:.0f} M records = {n * 32 / (ms / max(k, 1)) / 1e9:.2f} TB/s   hits {c[0]} low {c[1]}", flush=True)
