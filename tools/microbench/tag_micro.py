"""micro-benchmark: tagger kernel time on the C4 layout (200 M records): the 32-byte record stream against the 8-byte key column, with /
without the MAPQ-0 by-product, short and long inserts; GF_DIAGNOSTICS=1 adds the key-column kernel with parts switched off (tag_dbg)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gappadder_amd.hip_api import GapFill
from gappadder_amd import _lib as B
lib = B.lib()
gf = GapFill(0)
cfg = GapFill.synth_cfg(seed=20260004, scaffold_len=5_000_000, n_scaffolds=620, gaps_per_scaffold=32, gap_len=2000)
gaps, flanks = GapFill.synth_layout(cfg)
gf.set_gaps(gaps, 620, None)
n_pairs = 100_000_000
dev = torch.device("cuda:0")
d_reads = torch.empty(2 * n_pairs * 38 + 64, dtype=torch.uint8, device=dev)
d_recs = torch.empty(2 * n_pairs * 32, dtype=torch.uint8, device=dev)
gf.synth_pairs_dev(cfg, 0, n_pairs, d_reads.data_ptr(), d_recs.data_ptr())
gf.sync()
del d_reads
n = 2 * n_pairs
cap = 1 << 24
d_out = torch.empty(cap * 16, dtype=torch.uint8, device=dev)
d_low = torch.empty(cap * 12, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(8, dtype=torch.int32, device=dev)
d_keys = torch.empty(n + 1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
assert lib.gf_alnrec_keys_dev(gf.handle, d_recs.data_ptr(), n, d_keys.data_ptr()) == 0
gf.sync()
diag = bool(os.environ.get("GF_DIAGNOSTICS"))
for ins, sd in ((300, 30), (5000, 500)):
    for name, keyed, low, dbg in [("records", 0, 1, 0), ("keys", 1, 1, 0), ("keys, no by-product", 1, 0, 0)] + \
                                 ([("keys dbg: by-product off in the kernel", 1, 1, 1), ("keys dbg: nothing passes the bin map", 1, 1, 2),
                                   ("keys dbg: stream only", 1, 1, 3), ("keys dbg: candidates dropped", 1, 1, 4)] if diag else []):
        if diag:
            gf.set_option("tag_dbg", dbg)
        for it in range(4):
            lo = (d_low.data_ptr(), cap, d_cnt.data_ptr() + 4) if low else (None, 0, None)
            if keyed:
                rc = lib.gf_tag_alignments_keys_dev(gf.handle, d_recs.data_ptr(), d_keys.data_ptr(), n, ins, sd, 250, 30, d_out.data_ptr(), cap, d_cnt.data_ptr(), *lo)
            else:
                rc = lib.gf_tag_alignments_low_dev(gf.handle, d_recs.data_ptr(), n, ins, sd, 250, 30, d_out.data_ptr(), cap, d_cnt.data_ptr(), *lo)
            assert rc == 0, rc
            gf.sync()
            if it == 0:
                gf.timing(True)
        ms, k = gf.kernel_time(B.KERNEL_TAG)
        c = d_cnt.cpu().numpy()
        print(f"IS {ins}: {name:45s} {ms / max(k, 1):.3f} ms per {n / 1e6:.0f} M records   hits {c[0]} low {c[1]}", flush=True)
