"""Per-gap FASTQ pools -> fixed-length 2-bit packed reads for the GPU (ingest side, host).  Reads of a pool may differ in
length: they are packed at the pool set's maximum length, the tail marked as N in the mask (k-mers touching it are skipped,
like KMC skips k-mers with non-ACGT symbols)."""
import numpy as np

from .hip_api import GapFill


def read_fastq_seqs(path):
    seqs = []
    with open(path) as f:
        while True:
            h = f.readline()
            if not h:
                break
            seqs.append(f.readline().strip())
            f.readline()
            f.readline()
    return seqs


def pack_pools(pools):
    """pools: list of lists of sequences.  Returns (packed [n, rb], n_mask [n, nmw], pool_off [len(pools)+1], read_len)."""
    L = max([len(s) for p in pools for s in p] + [16])
    blob = bytearray()
    off = [0]
    for p in pools:
        for s in p:
            blob += s.encode().ljust(L, b"N")
        off.append(off[-1] + len(p))
    packed, nm = GapFill.pack_reads(bytes(blob), L, with_mask=True)
    return packed, nm, np.asarray(off, dtype=np.uint64), L
