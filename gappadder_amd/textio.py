"""Text of the reference's file contract for many records at once (csrc/textio.hip through the C ABI: plain host loops, no GPU work).
The callers hold the bytes already — BAM records fetched from the device, the input FASTQ files mapped — and would otherwise format
one record per CPython iteration."""
import ctypes as C

import numpy as np

from . import _lib as B


def _chk(rc, what, h):
    if rc:
        raise B.GapFillError(rc, what, B.lib().gf_last_error(h).decode() if h else "")


def bam_records_text(blob, rec_begin, ref_names, handle=None):
    """blob: uint8 array of raw BAM alignment records, record i at rec_begin[i] -> (SAM lines, both-unmapped FASTQ form), bytes each
    (gf_bam_records_text: `samtools view`'s eleven mandatory columns; `@{QNAME}_1|_2`, SEQ, `+`, QUAL)."""
    lib = B.lib()
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    rb = np.ascontiguousarray(rec_begin, dtype=np.uint64)
    names = b"".join(n.encode() + b"\0" for n in ref_names) or b"\0"
    n_sam, n_fq = C.c_size_t(0), C.c_size_t(0)
    args = (handle, B._p(blob), len(blob), B._p(rb), len(rb), names, len(ref_names))
    rc = lib.gf_bam_records_text(*args, None, 0, C.byref(n_sam), None, 0, C.byref(n_fq))
    if rc not in (B.GF_OK, B.GF_E_NOSPACE):
        _chk(rc, "gf_bam_records_text", handle)
    sam = np.empty(max(1, n_sam.value), dtype=np.uint8)
    fq = np.empty(max(1, n_fq.value), dtype=np.uint8)
    _chk(lib.gf_bam_records_text(*args, B._p(sam), len(sam), C.byref(n_sam), B._p(fq), len(fq), C.byref(n_fq)), "gf_bam_records_text", handle)
    return sam[:n_sam.value].tobytes(), fq[:n_fq.value].tobytes()


def fastq_records_text(files, begin, end, which, suffixes, want_ids=False, handle=None):
    """files: the FASTQ files as buffers (mmap / bytes / uint8 arrays) or as open binary file objects / descriptors (then every record is
    one positioned read); record i = files[which[i]][begin[i]:end[i]] -> (text, text_end[n]) or (text, text_end, ids, ids_end): the
    records as the reference re-writes them (`@{id}{suffixes[which[i]]}`, sequence, `+`, qualities), back to back in one uint8 array;
    record i ends at text_end[i]."""
    import os
    lib = B.lib()
    by_fd = all(isinstance(f, int) or hasattr(f, "fileno") and not hasattr(f, "find") for f in files) and len(files) > 0
    n = len(begin)
    begin = np.ascontiguousarray(begin, dtype=np.uint64)
    end = np.ascontiguousarray(end, dtype=np.uint64)
    which = np.ascontiguousarray(which, dtype=np.uint8)
    if by_fd:
        fds = np.array([f if isinstance(f, int) else f.fileno() for f in files], dtype=np.int32)
        lens = np.array([os.fstat(int(fd)).st_size for fd in fds], dtype=np.uint64)
        ptrs = None
    else:
        arrs = [np.frombuffer(f, dtype=np.uint8) if not isinstance(f, np.ndarray) else f for f in files]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        lens = np.array([len(a) for a in arrs], dtype=np.uint64)
        fds = None
    sfx = (C.c_char_p * len(files))(*[bytes(s) for s in suffixes])
    out_end = np.zeros(max(1, n), dtype=np.uint64)
    ids_end = np.zeros(max(1, n), dtype=np.uint64) if want_ids else None
    n_out, n_ids = C.c_size_t(0), C.c_size_t(0)
    # sized from the slices: the output of a record is at most its bytes + suffix + 6 (`@`, `+`, line ends of a record cut short)
    cap = int((end - begin).sum()) + n * (max(len(s) for s in suffixes) + 8) + 16 if n else 16
    out = np.empty(cap, dtype=np.uint8)
    ids = np.empty(int((end - begin).sum()) + 16 if want_ids else 1, dtype=np.uint8)
    _chk(lib.gf_fastq_records_text(handle, ptrs, B._p(fds), B._p(lens), len(files), B._p(begin), B._p(end), B._p(which), sfx, n, B._p(out), len(out),
                                   B._p(out_end), B._p(ids) if want_ids else None, len(ids) if want_ids else 0, B._p(ids_end) if want_ids else None,
                                   C.byref(n_out), C.byref(n_ids)), "gf_fastq_records_text", handle)
    if want_ids:
        return out[:n_out.value], out_end[:n], ids[:n_ids.value], ids_end[:n]
    return out[:n_out.value], out_end[:n]
