"""Scratch measurement: tagger kernel alone on the C2 records."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gappadder_amd import _lib as B
from gappadder_amd.hip_api import GapFill
c4 = len(sys.argv) > 1 and sys.argv[1] == "C4"
n_pairs = 100_000_000 if c4 else 25_000_000
cfg = GapFill.synth_cfg(seed=20260004, n_scaffolds=620, gaps_per_scaffold=32) if c4 else GapFill.synth_cfg()
gaps, flanks = GapFill.synth_layout(cfg)
gf = GapFill(0)
for kv in sys.argv[2:]:
    k, v = kv.split("="); gf.set_option(k, int(v))
gf.set_gaps(gaps, 620 if c4 else 50, None)
dev = torch.device("cuda:0")
d_reads = torch.empty(2 * n_pairs * 38 + 64, dtype=torch.uint8, device=dev)
d_recs = torch.empty(2 * n_pairs * 32, dtype=torch.uint8, device=dev)
gf.synth_pairs_dev(cfg, 0, n_pairs, d_reads.data_ptr(), d_recs.data_ptr())
cap = 1 << 23
d_t = torch.empty(cap * 12, dtype=torch.uint8, device=dev)
d_low = torch.empty(cap * 4 * 12, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(8, dtype=torch.int32, device=dev)
L, h, cp = B.lib(), gf.handle, d_cnt.data_ptr()
for mode in ("plain", "low"):
    def run():
        if mode == "plain":
            rc = L.gf_tag_alignments_dev(h, d_recs.data_ptr(), 2 * n_pairs, 300, 30, 250, 30, d_t.data_ptr(), cap, cp)
        else:
            rc = L.gf_tag_alignments_low_dev(h, d_recs.data_ptr(), 2 * n_pairs, 300, 30, 250, 30, d_t.data_ptr(), cap, cp, d_low.data_ptr(), cap * 4, cp + 4)
        assert rc == 0
    run(); gf.sync()
    gf.timing(True)
    for _ in range(10): run()
    ms, n = gf.kernel_time(B.KERNEL_TAG)
    print("%s: %.3f ms (%.0f GB/s) hits=%d low=%d" % (mode, ms / n, 2 * n_pairs * 32 / (ms / n * 1e-3) / 1e9, int(d_cnt[0]), int(d_cnt[1])))
    gf.timing(False)
