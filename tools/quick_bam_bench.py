"""Scratch measurement: BAM ingest on the GPU (gf_bgzf_inflate + gf_bam_pack) vs zlib on one host core, on a synthetic BAM
with realistic record content (random read bases and qualities, so the DEFLATE stream is not trivially compressible)."""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import bam_util as U
from gappadder_amd.hip_api import GapFill
from gappadder_amd import bam_io, _lib as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
rng = np.random.RandomState(1)
names = ["scf%d" % i for i in range(50)]
acgt = np.frombuffer(b"ACGT", np.uint8)
qs = np.frombuffer(b"FFFFFFFF:FF,F#", np.uint8)
t = time.time()
lines = []
for i in range(n):
    seq = bytes(acgt[rng.randint(0, 4, 150)]).decode()
    qual = bytes(qs[rng.randint(0, len(qs), 150)]).decode()
    lines.append("r%d\t%d\t%s\t%d\t%d\t150M\t=\t%d\t%d\t%s\t%s\tNM:i:0\tAS:i:150" % (i, 99 if i & 1 else 147, names[i * 50 // n], 1000 + i, 60 if i % 50 else 0,
                                                                            1300 + i, 450, seq, qual))
stream = U.sam_to_bam_stream(lines, names, [10 ** 8] * 50)
bam = U.bgzf_compress(stream, levels=(6,))
print("made %d records: %.1f MB inflated, %.1f MB BGZF (%.0f s)" % (n, len(stream) / 1e6, len(bam) / 1e6, time.time() - t))
gf = GapFill(0)
gf.bgzf_inflate(bam[:U.struct.unpack_from("<H", bam, 16)[0] + 1])   # warm-up: one block
gf.timing(True)
t = time.time(); out, used = gf.bgzf_inflate(bam); t_inf = time.time() - t
assert out.tobytes() == stream
hdr_names, first = bam_io.parse_header(out)
t = time.time(); recs, rb, cons = gf.bam_pack(None, first, np.arange(50, dtype=np.uint32), n_bytes=len(out)); t_pack = time.time() - t
assert len(recs) == n and cons == len(out)
tm = gf.kernel_time(B.KERNEL_INGEST)
print("gf_bgzf_inflate: %.3f s wall (H2D + kernel + D2H of the inflated bytes) = %.2f GB/s inflated" % (t_inf, len(out) / t_inf / 1e9))
print("gf_bam_pack    : %.3f s wall = %.2e records/s" % (t_pack, n / t_pack))
print("ingest kernels : %.2f ms total over %d launches-groups" % (tm[0], tm[1]))
t = time.time(); ref = U.bgzf_decompress(bam); t_z = time.time() - t
print("zlib, one host core: %.3f s = %.2f GB/s inflated" % (t_z, len(ref) / t_z / 1e9))
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 16
big = bam[:-len(U.BGZF_EOF)] * rep          # concatenated BGZF members: enough blocks to fill the chip (inflate only)
gf.timing(True)
t = time.time(); out, used = gf.bgzf_inflate(big); t_inf = time.time() - t
tm = gf.kernel_time(B.KERNEL_INGEST)
print("x%d: %d blocks, %.1f MB BGZF -> %.1f MB: kernel %.2f ms = %.1f GB/s inflated (%.1f GB/s of file bytes); wall %.3f s" %
      (rep, rep * (len(stream) // 0xFF00 + 1), len(big) / 1e6, len(out) / 1e6, tm[0], len(out) / tm[0] / 1e6, len(big) / tm[0] / 1e6, t_inf))
