"""TEST INFRASTRUCTURE — ctypes loader of oracle/liboracle.so (gp_oracle.c).  tests/, smoke() and bench.py's
cpu_baseline leg only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

GAP = np.dtype([("scaffold", "<u4"), ("start", "<u4"), ("end", "<u4"), ("idx_in_scaffold", "<u4")])
ALNREC = np.dtype([("pos", "<u4"), ("mate_pos", "<u4"), ("tlen", "<i4"), ("ref", "<u4"), ("mate_ref", "<u4"),
                   ("flag", "<u2"), ("mapq", "u1"), ("clipflag", "u1"), ("read", "<u8")])
TAGHIT = np.dtype([("rec", "<u4"), ("gap", "<u4"), ("kind", "<u2"), ("to_mate", "<u2")])
DPOS = np.dtype([("mate_scaffold", "<u4"), ("mate_pos", "<u4"), ("src_scaffold", "<u4"), ("src_gap", "<u4")])
HIT = np.dtype([("gap", "<u4"), ("read", "<u4")])
SYNTH_CFG = np.dtype([("seed", "<u8"), ("scaffold_len", "<u8"), ("n_scaffolds", "<u4"), ("gaps_per_scaffold", "<u4"),
                      ("gap_len", "<u4"), ("read_len", "<u4"), ("insert_mean", "<u4"), ("insert_sd", "<u4"),
                      ("err_q16", "<u4"), ("mapq0_q16", "<u4"), ("chimeric_q16", "<u4"), ("flank_len", "<u4"),
                      ("library", "<u4"), ("repeats", "<u4")])


def lib():
    global _lib
    if _lib is None:
        src = [os.path.join(_HERE, f) for f in ("gp_oracle.c", "gp_oracle.h", "../include/gf_synth.h")]
        if not os.path.exists(_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_PATH) for s in src):
            subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
        L = C.CDLL(_PATH)
        vp, sz, i32, u32 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32
        L.or_tag_alignments.restype = sz
        L.or_tag_alignments.argtypes = [vp, sz, vp, sz, i32, i32, i32, i32, vp, sz]
        L.or_tag_low_mapq.restype = sz
        L.or_tag_low_mapq.argtypes = [vp, sz, vp, sz, vp, sz]
        L.or_screen_reads.restype = sz
        L.or_screen_reads.argtypes = [C.c_char_p, sz, i32, C.c_char_p, vp, sz, i32, i32, u32, vp, sz, i32]
        L.or_screen_last_build_s.restype = C.c_double
        L.or_screen_last_build_s.argtypes = []
        L.or_set_threads.restype = None
        L.or_set_threads.argtypes = [i32]
        L.or_pack_kmer64.restype = C.c_uint64
        L.or_pack_kmer64.argtypes = [C.c_char_p, i32]
        L.or_unpack_reads.restype = None
        L.or_unpack_reads.argtypes = [vp, sz, i32, vp]
        L.or_count_kmers.restype = sz
        L.or_count_kmers.argtypes = [C.c_char_p, sz, i32, i32, i32, vp, vp, vp, sz]
        L.or_assemble_pool.restype = sz
        L.or_assemble_pool.argtypes = [C.c_char_p, sz, i32, i32, i32, i32, i32, vp, vp, vp, sz, vp, sz, C.POINTER(sz)]
        L.or_assemble_pool2.restype = sz
        L.or_assemble_pool2.argtypes = [C.c_char_p, sz, i32, i32, i32, i32, i32, i32, vp, vp, vp, sz, vp, sz, C.POINTER(sz)]
        L.or_assemble_pool3.restype = sz
        L.or_assemble_pool3.argtypes = [C.c_char_p, sz, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, sz, vp, sz, C.POINTER(sz)]
        L.or_quick_check.restype = sz
        L.or_quick_check.argtypes = [C.c_char_p, vp, sz, i32, vp, vp, sz]
        L.or_overlap_evaluate.restype = None
        L.or_overlap_evaluate.argtypes = [C.c_char_p, i32, C.c_char_p, i32, vp, vp]
        L.or_synth_pairs.restype = None
        L.or_synth_pairs.argtypes = [vp, C.c_uint64, sz, vp, vp]
        L.or_synth_layout.restype = None
        L.or_synth_layout.argtypes = [vp, vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def screen_last_build_s():
    """Seconds the last screen_reads call spent building its flank k-mer table (the rest of the call is the per-read pass)."""
    return float(lib().or_screen_last_build_s())


def set_threads(n):
    lib().or_set_threads(int(n))


def tag_alignments(recs, gaps, insert_size, sd, clip_dist=250, anchor_mapq=30):
    recs = np.ascontiguousarray(recs, dtype=ALNREC)
    gaps = np.ascontiguousarray(gaps, dtype=GAP)
    cap = max(1024, 4 * len(recs))
    out = np.zeros(cap, dtype=TAGHIT)
    n = lib().or_tag_alignments(_p(recs), len(recs), _p(gaps), len(gaps), insert_size, sd, clip_dist, anchor_mapq, _p(out), cap)
    assert n <= cap
    return out[:n]


def tag_low_mapq(recs, table):
    recs = np.ascontiguousarray(recs, dtype=ALNREC)
    table = np.ascontiguousarray(table, dtype=DPOS)
    cap = max(1024, 8 * len(recs))
    while True:
        out = np.zeros(cap, dtype=TAGHIT)
        n = lib().or_tag_low_mapq(_p(recs), len(recs), _p(table), len(table), _p(out), cap)
        if n <= cap:
            return out[:n]
        cap = n


def screen_reads(reads_blob, read_len, flanks, k, min_hits=1, max_gaps_per_kmer=0, threads=0):
    """reads_blob: bytes of n*read_len ASCII bases; flanks: [(left, right)]."""
    n = len(reads_blob) // read_len
    parts, off = [], [0]
    for l, r in flanks:
        for s in (l, r):
            parts.append(s)
            off.append(off[-1] + len(s))
    fblob = "".join(parts).encode()
    offs = np.asarray(off, dtype=np.uint64)
    cap = max(1024, n)
    while True:
        out = np.zeros(cap, dtype=HIT)
        cnt = lib().or_screen_reads(bytes(reads_blob), n, read_len, fblob, _p(offs), len(flanks), k, min_hits,
                                    max_gaps_per_kmer, _p(out), cap, threads)
        if cnt <= cap:
            return out[:cnt]
        cap = cnt


def unpack_reads(packed, read_len):
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    n = packed.shape[0]
    out = np.zeros(n * read_len, dtype=np.uint8)
    lib().or_unpack_reads(_p(packed), n, read_len, _p(out))
    return out.tobytes()


def synth_cfg(seed=20260002, scaffold_len=5_000_000, n_scaffolds=50, gaps_per_scaffold=20, gap_len=2000, read_len=150,
              insert_mean=300, insert_sd=30, err=0.005, mapq0=0.02, chimeric=0.01, flank_len=300, library=0, repeat_period=0,
              repeat_copies=50):
    c = np.zeros(1, dtype=SYNTH_CFG)
    c[0] = (seed, scaffold_len, n_scaffolds, gaps_per_scaffold, gap_len, read_len, insert_mean, insert_sd,
            int(round(err * 65536)), int(round(mapq0 * 65536)), int(round(chimeric * 65536)), flank_len, library,
            (int(repeat_period) & 0xFF) | ((int(repeat_copies) & 0xFF) << 8) if repeat_period else 0)
    return c


def synth_pairs(cfg, first_pair, n_pairs, with_records=True):
    rb = (int(cfg["read_len"][0]) + 3) // 4
    packed = np.zeros((2 * n_pairs, rb), dtype=np.uint8)
    recs = np.zeros(2 * n_pairs, dtype=ALNREC) if with_records else None
    lib().or_synth_pairs(_p(cfg), first_pair, n_pairs, _p(packed), _p(recs) if with_records else None)
    return packed, recs


def synth_layout(cfg):
    n = int(cfg["n_scaffolds"][0]) * int(cfg["gaps_per_scaffold"][0])
    fl = int(cfg["flank_len"][0]) - 5
    gaps = np.zeros(n, dtype=GAP)
    blob = np.zeros(2 * n * fl, dtype=np.uint8)
    off = np.zeros(2 * n + 1, dtype=np.uint64)
    lib().or_synth_layout(_p(cfg), _p(gaps), _p(blob), _p(off))
    b = blob.tobytes().decode()
    flanks = [(b[int(off[2 * g]):int(off[2 * g + 1])], b[int(off[2 * g + 1]):int(off[2 * g + 2])]) for g in range(n)]
    return gaps, flanks


def count_kmers(reads_blob, read_len, k, min_count=2):
    """[(hi, lo, count)] ascending = the `kmc -k{k}` | `kmc_dump` listing."""
    n = len(reads_blob) // read_len
    cap = max(16, n * (read_len - k + 1))
    hi = np.zeros(cap, np.uint64); lo = np.zeros(cap, np.uint64); cnt = np.zeros(cap, np.uint32)
    m = lib().or_count_kmers(bytes(reads_blob), n, read_len, k, min_count, _p(hi), _p(lo), _p(cnt), cap)
    return hi[:m], lo[:m], cnt[:m]


def assemble_pool(reads_blob, read_len, k, kv, min_count=2, min_contig=40, simplify=8, tiebreak="counts"):
    """[(sequence, n_nodes, cov_sum)] sorted by (-length, sequence).  simplify = rounds of tip clipping + bubble popping, stopping
    when a round removes nothing (8: the product's default = to convergence in practice, Velvet's defaults on; 0: raw unitigs).
    tiebreak: "counts" (default: fewer weak nodes win between equal coverage) or "none" (the reference-shaped mode: sequence order
    alone, nothing Velvet could not have known — the product's option asm_tiebreak = 0)."""
    tb = {"counts": 1, "none": 0}[tiebreak]
    n = len(reads_blob) // read_len
    cap = max(16, n * (read_len - k + 1))
    nn = np.zeros(cap, np.uint32); ln = np.zeros(cap, np.uint32); cv = np.zeros(cap, np.uint32)
    scap = max(1024, 8 * n * read_len)
    seq = np.zeros(scap, np.uint8)
    need = C.c_size_t(0)
    m = lib().or_assemble_pool3(bytes(reads_blob), n, read_len, k, kv, min_count, min_contig, simplify, tb, _p(nn), _p(ln), _p(cv), cap,
                                _p(seq), scap, C.byref(need))
    assert m <= cap and need.value <= scap
    out, off = [], 0
    b = seq.tobytes()
    for i in range(m):
        out.append((b[off:off + int(ln[i])].decode(), int(nn[i]), int(cv[i])))
        off += int(ln[i])
    return out


def quick_check(contigs, k=10):
    """[(i, j)] feasible pairs of the node list [c0, revcomp(c0), c1, ...] (the contig merger's prefilter), in (i, j) order."""
    blob = "".join(contigs).encode()
    off = np.zeros(len(contigs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(c) for c in contigs])
    nn = 2 * len(contigs)
    cap = nn * (nn + 1) // 2 + 1
    oi = np.zeros(cap, np.uint32); oj = np.zeros(cap, np.uint32)
    m = lib().or_quick_check(blob, _p(off), len(contigs), k, _p(oi), _p(oj), cap)
    assert m <= cap
    return list(zip(oi[:m].tolist(), oj[:m].tolist()))


OVL_PARAMS = np.dtype([("mismatch", "<f8"), ("indel", "<f8"), ("max_clip", "<f8"), ("frac_min_overlap", "<f8"), ("frac_loss", "<f8"),
                       ("min_overlap", "<f8"), ("min_overlap_scaffold", "<f8"), ("relax", "<f8")])
OVL_RESULT = np.dtype([(n, "<i4") for n in ("res", "row_end", "col_end", "nclip", "score", "contained", "merged_len", "overlap",
                                            "containment", "first_goes_first")])
GAPPADDER_OVL = (-2.0, -2.0, 50.0, 0.005, 0.4, 12.0, 6.0)   # MergeContigs.py:75 (-i1 -i2 -y -s -x) + ContigsMerger's defaults (main.cpp:24-27)


def merger_nodes(contigs):
    """The contig merger's node list [c0, revcomp(c0), c1, ...] (CompactVer3, ContigsCompactor.cpp:782-800; upper-cased)."""
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    out = []
    for c in contigs:
        c = c.upper()
        out += [c, "".join(comp.get(ch, ch) for ch in reversed(c))]
    return out


def overlap_evaluate(s1, s2, params=GAPPADDER_OVL, relax=False):
    """ContigsCompactor::Evaluate on one ordered pair of node strings -> dict of OVL_RESULT fields.  relax: its fRelax mode
    (no significance test; FormMergedSeqFromPath, ContigsCompactor.cpp:1489)."""
    pr = np.zeros(1, OVL_PARAMS); pr[0] = tuple(params)[:7] + (1.0 if relax else 0.0,)
    out = np.zeros(1, OVL_RESULT)
    lib().or_overlap_evaluate(s1.encode(), len(s1), s2.encode(), len(s2), _p(pr), _p(out))
    return {n: int(out[0][n]) for n in OVL_RESULT.names}
