"""FETCH_SIZE calibration on known byte counts (run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`):
 (a) torch copy of 1.9 GB (reads 1.9 GB), (b) the screen filter (its 16-B/lane tile stream reads n_reads * 38 B, the probes add
 the bitmap / exact-set lines).  (Rounds 1-3 also ran the filter with its probes disabled through a timing-only option that has
 since been removed; the result — FETCH_SIZE reports exactly half of the bytes on gfx950 — is recorded in DESIGN.md.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gappadder_amd import _lib as B
from gappadder_amd.hip_api import GapFill
n_reads, L, k = 50_000_000, 150, 31
dev = torch.device("cuda:0")
reads = torch.randint(0, 256, (n_reads, 38), dtype=torch.uint8, device=dev)
dst = torch.empty_like(reads)
for _ in range(3):
    dst.copy_(reads)
torch.cuda.synchronize()
rng = np.random.RandomState(1)
n_gaps = 1000
gaps = np.zeros(n_gaps, dtype=B.GAP)
gaps["scaffold"] = np.arange(n_gaps) // 20
gaps["start"] = (np.arange(n_gaps) % 20 + 1) * 200000
gaps["end"] = gaps["start"] + 2000
gaps["idx_in_scaffold"] = np.arange(n_gaps) % 20 + 1
lut = np.frombuffer(b"ACGT", np.uint8)
flanks = [(lut[rng.randint(0, 4, 295)].tobytes().decode(), lut[rng.randint(0, 4, 295)].tobytes().decode()) for _ in range(n_gaps)]
gf = GapFill(0)
out = torch.zeros(1 << 20, 2, dtype=torch.int32, device=dev)
nout = torch.zeros(4, dtype=torch.int32, device=dev)
for _rep in (0,):
    gf.set_gaps(gaps, int(gaps["scaffold"].max()) + 1, flanks)
    for _ in range(3):
        rc = B.lib().gf_screen_reads_dev(gf.handle, reads.data_ptr(), None, n_reads, L, k, 1, out.data_ptr(), out.shape[0], nout.data_ptr())
        assert rc == 0
    gf.sync()
print("bytes per launch: copy reads %d, filter reads %d" % (reads.numel(), reads.numel()))
