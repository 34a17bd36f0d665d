"""Scratch measurement: gf_fastq_pack_dev on FASTQ text resident in HBM (2 M x 150-bp records by default)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gappadder_amd import _lib as B
from gappadder_amd.hip_api import GapFill
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000, 150
rng = np.random.RandomState(1)
rec = 12 + L + 3 + L + 1       # "@r%09d\n" = 12 bytes incl. newline? -> '@' 'r' 9 digits '\n' = 12
text = np.empty((n, rec), dtype=np.uint8)
ids = np.char.zfill(np.arange(n).astype("S9"), 9)
text[:, 0] = ord("@"); text[:, 1] = ord("r")
text[:, 2:11] = np.frombuffer(ids.tobytes(), np.uint8).reshape(n, 9)
text[:, 11] = 10
text[:, 12:12 + L] = np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, (n, L))]
text[:, 12 + L] = 10; text[:, 13 + L] = ord("+"); text[:, 14 + L] = 10
text[:, 15 + L:15 + 2 * L] = ord("I"); text[:, 15 + 2 * L] = 10
dev = torch.device("cuda:0")
d_text = torch.from_numpy(text.reshape(-1)).to(dev)
gf = GapFill(0)
rb = 38
d_packed = torch.empty(n * rb + 64, dtype=torch.uint8, device=dev)
d_nm = torch.empty(n * 5, dtype=torch.int32, device=dev)
d_hdr = torch.empty(n + 1, dtype=torch.int64, device=dev)
d_cnt = torch.zeros(4, dtype=torch.int64, device=dev)
L_ = B.lib()
def run():
    rc = L_.gf_fastq_pack_dev(gf.handle, d_text.data_ptr(), d_text.numel(), L, d_packed.data_ptr(), n, d_nm.data_ptr(), d_hdr.data_ptr(),
                              d_cnt.data_ptr(), d_cnt.data_ptr() + 8)
    assert rc == 0
run(); gf.sync()
assert int(d_cnt[0]) == n
t = time.time()
for _ in range(5): run()
gf.sync()
dt = (time.time() - t) / 5
print("ingest: %d records, %.1f MB text: %.3f ms = %.1f GB/s of text, %.2e reads/s (PCIe at 63 GB/s would need %.1f ms)"
      % (n, d_text.numel() / 1e6, dt * 1e3, d_text.numel() / dt / 1e9, n / dt, d_text.numel() / 63e9 * 1e3))
exp, _ = GapFill.pack_reads(text[:1000, 12:12 + L].tobytes(), L)
assert np.array_equal(d_packed[:1000 * rb].cpu().numpy().reshape(1000, rb), exp)
