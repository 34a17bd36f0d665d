"""Scratch measurement (not the bench): the screen filter alone on synthetic C4-shaped reads resident in HBM, with a chosen build of
the library (experiments on pass A).  usage: exp_filter.py <lib.so or -> [n_pairs] [k] [variant]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gappadder_amd import _lib as B
if len(sys.argv) > 1 and sys.argv[1] != "-":
    B.LIB_PATH = os.path.abspath(sys.argv[1])
from gappadder_amd.hip_api import GapFill
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 112_500_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 51
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
L = 150
cfg = GapFill.synth_cfg(seed=20260004, scaffold_len=5_000_000, n_scaffolds=620, gaps_per_scaffold=32, gap_len=2000)
gaps, flanks = GapFill.synth_layout(cfg)
gf = GapFill(0)
gf.set_gaps(gaps, 620, flanks)
gf.set_option("screen_variant", variant)
dev = torch.device("cuda:0")
lib = B.lib()
rb = lib.gf_packed_read_bytes(L)
d_reads = torch.empty(2 * n_pairs * rb + 64, dtype=torch.uint8, device=dev)
gf.synth_pairs_dev(cfg, 0, n_pairs, d_reads.data_ptr())
out = torch.zeros(1 << 22, 2, dtype=torch.int32, device=dev)
nout = torch.zeros(4, dtype=torch.int32, device=dev)
def run():
    rc = lib.gf_screen_reads_dev(gf.handle, d_reads.data_ptr(), None, 2 * n_pairs, L, k, 1, out.data_ptr(), out.shape[0], nout.data_ptr())
    assert rc == 0, rc
run(); gf.sync()
gf.timing(True)
for _ in range(5):
    run()
gf.sync()
ms, nl = gf.kernel_time(B.KERNEL_SCREEN)
ms2, nl2 = gf.kernel_time(B.KERNEL_VERIFY)
print("%s reads=%d k=%d variant=%d: filter %.3f ms (%.2f TB/s algorithmic) verify %.3f ms hits=%d" %
      (os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "-", 2 * n_pairs, k, variant, ms / nl, 2 * n_pairs * rb / (ms / nl * 1e-3) / 1e12, ms2 / nl2, int(nout[0])))
