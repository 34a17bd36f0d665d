"""Scratch measurement: screen filter kernel on uniform-random packed reads resident in HBM (not the bench)."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from gappadder_amd import _lib as B
from gappadder_amd.hip_api import GapFill

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
n_gaps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 31
L = 150
rng = np.random.RandomState(1)
gaps = np.zeros(n_gaps, dtype=B.GAP)
gaps["scaffold"] = np.arange(n_gaps) // 20
gaps["start"] = (np.arange(n_gaps) % 20 + 1) * 200000
gaps["end"] = gaps["start"] + 2000
gaps["idx_in_scaffold"] = np.arange(n_gaps) % 20 + 1
lut = np.frombuffer(b"ACGT", np.uint8)
flanks = [(lut[rng.randint(0, 4, 295)].tobytes().decode(), lut[rng.randint(0, 4, 295)].tobytes().decode()) for _ in range(n_gaps)]
gf = GapFill(0)
variants = [0] if len(sys.argv) <= 5 else [int(x) for x in sys.argv[5].split(",")]
import itertools
for bl, var in itertools.product(([0] if len(sys.argv) <= 4 else [int(x) for x in sys.argv[4].split(",")]), variants):
    fuse = 0
    gf.set_option("screen_stream_policy", int(os.environ.get("POL", "1")))
    gf.set_option("bitmap_log2", bl)
    gf.set_option("screen_lds_log2_max", int(os.environ.get("LDSMAX", "20")))
    gf.set_option("screen_variant", var % 100)
    gf.set_option("screen_ext", int(os.environ.get("EXT", "1")))
    t = time.time()
    gf.set_gaps(gaps, int(gaps["scaffold"].max()) + 1, flanks)
    dev = torch.device("cuda:0")
    reads = torch.randint(0, 256, (n_reads, 38), dtype=torch.uint8, device=dev)
    out = torch.zeros(1 << 20, 2, dtype=torch.int32, device=dev)
    nout = torch.zeros(4, dtype=torch.int32, device=dev)
    L_ = B.lib()
    def run():
        rc = L_.gf_screen_reads_dev(gf.handle, reads.data_ptr(), None, n_reads, L, k, 1, out.data_ptr(), out.shape[0], nout.data_ptr())
        assert rc == 0, rc
    run(); gf.sync()
    print("index build+first run %.2fs" % (time.time() - t))
    gf.timing(True)
    t = time.time()
    for _ in range(5):
        run()
    gf.sync()
    dt = (time.time() - t) / 5
    ms, nl = gf.kernel_time(B.KERNEL_SCREEN)
    ms2, nl2 = gf.kernel_time(B.KERNEL_VERIFY)
    print("fuse=%d var=%d bitmap_log2=%d reads=%d gaps=%d k=%d: wall %.3f ms/iter; filter %.3f ms (%.1f GB/s, %.2e reads/s) verify %.3f ms; hits=%d"
          % (fuse, var, bl, n_reads, n_gaps, k, dt * 1e3, ms / nl, n_reads * 38 / (ms / nl * 1e-3) / 1e9, n_reads / (ms / nl * 1e-3), ms2 / nl2, int(nout[0])))
    gf.timing(False)
