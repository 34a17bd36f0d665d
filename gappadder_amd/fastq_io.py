"""Per-gap FASTQ pools -> fixed-length 2-bit packed reads for the GPU (ingest side, host).  Reads of a pool may differ in
length: they are packed at the pool set's maximum length, the tail marked as N in the mask (k-mers touching it are skipped,
like KMC skips k-mers with non-ACGT symbols)."""
import os
import zlib

import numpy as np

from .hip_api import GapFill


def is_gzip(path):
    with open(path, "rb") as f:
        return f.read(2) == b"\x1f\x8b"


def plain_fastq(path, tmp_dir, chunk=8 << 20):
    """SURVEY.md §8f-4 lists FASTQ(.gz) (the reference itself opens plain text, run_multi_threads_discordant.py:209): a gzip file —
    plain gzip, several concatenated members, or BGZF (bgzip), which is a sequence of gzip members — is inflated ONCE into
    `tmp_dir` (zlib on the host, streaming, every member in turn) and the plain copy's path returned; anything else comes back as it is.
    Every later step — the device ingest, the per-gap FASTQ files that are cut out of the input at recorded offsets, the host join of the
    per-scaffold path — then reads plain text.  An existing copy that is newer than its source is reused."""
    if not is_gzip(path):
        return path
    os.makedirs(tmp_dir, exist_ok=True)
    base = os.path.basename(path)
    out = os.path.join(tmp_dir, (base[:-3] if base.endswith(".gz") else base) + ".plain.fq")
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(path):
        return out
    tmp = out + ".part"
    with open(path, "rb") as f, open(tmp, "wb") as o:
        d = zlib.decompressobj(31)
        while True:
            buf = f.read(chunk)
            if not buf:
                break
            while buf:
                o.write(d.decompress(buf))
                if d.eof:                       # end of a member: the next one (BGZF blocks, `cat a.gz b.gz`) starts in what is left
                    buf = d.unused_data
                    d = zlib.decompressobj(31)
                else:
                    buf = b""
        o.write(d.flush())
    os.replace(tmp, out)
    return out


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
