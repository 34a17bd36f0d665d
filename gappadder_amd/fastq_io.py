"""Per-gap FASTQ pools -> fixed-length 2-bit packed reads for the GPU (ingest side, host).  Reads of a pool may differ in
length: they are packed at the pool set's maximum length, the tail marked as N in the mask (k-mers touching it are skipped,
like KMC skips k-mers with non-ACGT symbols)."""
import numpy as np

from .hip_api import GapFill


def read_fastq_seqs(path):
    """The sequence line of every record: second of each four lines, stripped; a file that ends inside a record still counts it."""
    with open(path) as f:
        parts = f.read().split("\n")
    n = len(parts) - 1 if parts[-1] == "" else len(parts)       # lines the file holds (the piece behind the last line end is no line)
    return [(parts[i + 1] if i + 1 < n else "").strip() for i in range(0, n, 4)]


def pack_pools(pools):
    """pools: list of lists of sequences.  Returns (packed [n, rb], n_mask [n, nmw], pool_off [len(pools)+1], read_len)."""
    L = max([len(s) for p in pools for s in p] + [16])
    off = [0]
    for p in pools:
        off.append(off[-1] + len(p))
    blob = "".join([s if len(s) == L else s.ljust(L, "N") for p in pools for s in p]).encode()
    packed, nm = GapFill.pack_reads(blob, L, with_mask=True)
    return packed, nm, np.asarray(off, dtype=np.uint64), L
