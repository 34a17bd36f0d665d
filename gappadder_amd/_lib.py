"""ctypes binding of libgapfill_hip.so (include/gapfill_hip.h).  There is NO fallback: if the HIP library is
missing or fails to load, importing any compute entry point raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgapfill_hip.so")

GAP = np.dtype([("scaffold", "<u4"), ("start", "<u4"), ("end", "<u4"), ("idx_in_scaffold", "<u4")])
ALNREC = np.dtype([("pos", "<u4"), ("mate_pos", "<u4"), ("tlen", "<i4"), ("ref", "<u4"), ("mate_ref", "<u4"),
                   ("flag", "<u2"), ("mapq", "u1"), ("clipflag", "u1"), ("read", "<u8")])
TAGHIT = np.dtype([("rec", "<u4"), ("gap", "<u4"), ("kind", "<u2"), ("to_mate", "<u2")])
DPOS = np.dtype([("mate_scaffold", "<u4"), ("mate_pos", "<u4"), ("src_scaffold", "<u4"), ("src_gap", "<u4")])
HIT = np.dtype([("gap", "<u4"), ("read", "<u4")])
CONTIG = np.dtype([("gap", "<u4"), ("k", "<u2"), ("kv", "<u2"), ("n_nodes", "<u4"), ("length", "<u4"), ("cov_sum", "<u4"),
                   ("reserved", "<u4"), ("seq_off", "<u8")])
SYNTH_CFG = np.dtype([("seed", "<u8"), ("scaffold_len", "<u8"), ("n_scaffolds", "<u4"), ("gaps_per_scaffold", "<u4"),
                      ("gap_len", "<u4"), ("read_len", "<u4"), ("insert_mean", "<u4"), ("insert_sd", "<u4"),
                      ("err_q16", "<u4"), ("mapq0_q16", "<u4"), ("chimeric_q16", "<u4"), ("flank_len", "<u4"),
                      ("library", "<u4"), ("repeats", "<u4")])
QCPAIR = np.dtype([("set", "<u4"), ("i", "<u4"), ("j", "<u4")])
OVL_PARAMS = np.dtype([("mismatch", "<f8"), ("indel", "<f8"), ("max_clip", "<f8"), ("frac_min_overlap", "<f8"), ("frac_loss", "<f8"),
                       ("min_overlap", "<f8"), ("min_overlap_scaffold", "<f8"), ("relax", "<f8")])
OVL_RESULT = np.dtype([(n, "<i4") for n in ("res", "row_end", "col_end", "nclip", "score", "contained", "merged_len", "overlap",
                                            "containment", "first_goes_first")])
assert CONTIG.itemsize == 32 and GAP.itemsize == 16 and ALNREC.itemsize == 32 and TAGHIT.itemsize == 12 and DPOS.itemsize == 16 and HIT.itemsize == 8

GF_OK, GF_E_INVAL, GF_E_NODEV, GF_E_NOMEM, GF_E_NOSPACE, GF_E_STATE, GF_E_UNSUPPORTED, GF_E_FORMAT = 0, -1, -2, -3, -4, -5, -6, -7
KIND_CLIP, KIND_DISCORDANT, KIND_UNMAP, KIND_LOWMAPQ = 0, 1, 2, 3
KIND_NAMES = {KIND_CLIP: "clip", KIND_DISCORDANT: "discordant", KIND_UNMAP: "unmap"}
KERNEL_SCREEN, KERNEL_TAG, KERNEL_LOWMAPQ, KERNEL_ASSEMBLE, KERNEL_POOL, KERNEL_SYNTH, KERNEL_COUNT, KERNEL_VERIFY, KERNEL_INGEST, KERNEL_PICK, KERNEL_MERGE = range(11)

# words of the merge round's statistics (gf_merge_open_gaps_dev, u32[32])
MG_N_PRE, MG_N_SETS, MG_SKIPPED, MG_N_PAIRS, MG_QC_FLAGS, MG_N_JOBS, MG_ERR, MG_N0, MG_N_EDGES, MG_SETS_WITH_JOBS = range(10)
MG_SKIPPED_GRAPH = 16
MG_WORDS = 32

_lib = None


class GapFillError(RuntimeError):
    def __init__(self, code, what, detail=""):
        self.code = code
        RuntimeError.__init__(self, "%s failed: %s (%d)%s" % (what, lib().gf_strerror(code).decode(), code,
                                                            (" — " + detail) if detail else ""))


def lib():
    """The loaded library; raises (loudly) when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libgapfill_hip.so not built (%s missing): run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C gappadder_amd/csrc`; there is no CPU fallback" % LIB_PATH)
    try:
        import torch  # noqa: F401  -- load torch's HIP runtime first: one libamdhip64 per process
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, i32, u32, u64p = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.POINTER(C.c_uint64)
    szp = C.POINTER(C.c_size_t)
    sig = {
        "gf_init": (i32, [i32, C.POINTER(vp)]),
        "gf_destroy": (None, [vp]),
        "gf_strerror": (C.c_char_p, [i32]),
        "gf_last_error": (C.c_char_p, [vp]),
        "gf_screen_kernels": (C.c_char_p, [vp]),
        "gf_screen_last_overflow": (i32, [vp, vp]),
        "gf_set_stream": (i32, [vp, vp]),
        "gf_sync": (i32, [vp]),
        "gf_stream_wait": (i32, [vp, vp]),
        "gf_stream_wait_after_filter": (i32, [vp, vp]),
        "gf_set_option": (i32, [vp, C.c_char_p, C.c_long]),
        "gf_set_gaps": (i32, [vp, vp, sz, u32, C.c_char_p, vp]),
        "gf_pack_reads": (i32, [C.c_char_p, sz, i32, vp, vp]),
        "gf_packed_read_bytes": (sz, [i32]),
        "gf_fastq_pack": (i32, [vp, C.c_char_p, sz, i32, vp, sz, vp, vp, szp, C.POINTER(C.c_uint32)]),
        "gf_sam_pack": (i32, [vp, C.c_char_p, sz, C.c_char_p, vp, sz, vp, sz, vp, szp]),
        "gf_bgzf_inflate": (i32, [vp, C.c_char_p, sz, C.c_char_p, sz, vp, sz, szp, szp]),
        "gf_tag_alignments_bam": (i32, [vp, i32, i32, i32, i32, vp, sz, szp]),
        "gf_tag_low_mapq_bam": (i32, [vp, vp, sz, vp, sz, szp]),
        "gf_bam_fetch": (i32, [vp, vp, vp, sz, vp, sz, szp]),
        "gf_bam_pack": (i32, [vp, vp, sz, sz, vp, sz, vp, sz, vp, szp, szp]),
        "gf_fastq_pack_dev": (i32, [vp, vp, sz, i32, vp, sz, vp, vp, vp, vp]),
        "gf_fastq_index_dev": (i32, [vp, vp, sz, vp, sz, vp, vp]),
        "gf_bam_append_dev": (i32, [vp, sz, sz, vp, sz, vp, sz, sz, vp, vp, sz, sz, vp, vp, sz, vp, szp, szp, szp]),
        "gf_read_join_dev": (i32, [vp, vp, sz, vp, vp, sz, vp]),
        "gf_fetch_slices": (i32, [vp, vp, sz, vp, vp, sz, vp, sz, szp]),
        "gf_gather_rows_dev": (i32, [vp, vp, sz, sz, vp, vp, sz, vp]),
        "gf_bam_records_text": (i32, [vp, vp, sz, vp, sz, vp, sz, vp, sz, szp, vp, sz, szp]),
        "gf_fastq_records_text": (i32, [vp, vp, vp, vp, sz, vp, vp, vp, vp, sz, vp, sz, vp, vp, sz, vp, szp, szp]),
        "gf_screen_reads": (i32, [vp, vp, vp, sz, i32, i32, i32, vp, sz, szp]),
        "gf_screen_reads_dev": (i32, [vp, vp, vp, sz, i32, i32, i32, vp, sz, vp]),
        "gf_tag_alignments": (i32, [vp, vp, sz, i32, i32, i32, i32, vp, sz, szp]),
        "gf_tag_alignments_dev": (i32, [vp, vp, sz, i32, i32, i32, i32, vp, sz, vp]),
        "gf_tag_low_mapq": (i32, [vp, vp, sz, vp, sz, vp, sz, szp]),
        "gf_tag_low_mapq_dev": (i32, [vp, vp, sz, vp, sz, vp, sz, vp]),
        "gf_second_hop_table_dev": (i32, [vp, vp, vp, vp, sz, vp, vp, sz, vp]),
        "gf_second_hop_table_merge_dev": (i32, [vp, vp, vp, vp, i32, sz, vp, vp, sz, vp]),
        "gf_tag_low_mapq_table_dev": (i32, [vp, vp, vp, sz, vp, vp, sz, vp, sz, vp]),
        "gf_pool_keys_all_dev": (i32, [vp, vp, vp, sz, i32, vp, vp, vp, sz, vp, vp, sz, vp, vp, sz, vp]),
        "gf_pool_keys_from_second_hop_dev": (i32, [vp, vp, vp, vp, sz, vp, vp, sz, vp]),
        "gf_tag_alignments_low_dev": (i32, [vp, vp, sz, i32, i32, i32, i32, vp, sz, vp, vp, sz, vp]),
        "gf_tag_low_mapq_compact_dev": (i32, [vp, vp, vp, sz, vp, sz, vp, sz, vp]),
        "gf_alnrec_keys_dev": (i32, [vp, vp, sz, vp]),
        "gf_tag_alignments_keys_dev": (i32, [vp, vp, vp, sz, i32, i32, i32, i32, vp, sz, vp, vp, sz, vp]),
        "gf_assemble": (i32, [vp, vp, vp, vp, sz, i32, vp, vp, i32, i32, i32, vp, sz, szp, vp, sz, szp]),
        "gf_assemble_dev": (i32, [vp, vp, vp, vp, sz, sz, i32, i32, i32, i32, i32, vp, sz, vp, vp, sz, vp, vp]),
        "gf_assemble_multi_dev": (i32, [vp, vp, vp, vp, sz, sz, i32, vp, vp, i32, i32, i32, vp, sz, vp, vp, sz, vp, vp]),
        "gf_assemble_last_launch": (i32, [vp, vp, vp, vp]),
        "gf_pool_counts_dev": (i32, [vp, vp, sz, vp]),
        "gf_pools_pack_for_owners_dev": (i32, [vp, vp, vp, sz, i32, i32, i32, i32, i32, vp, sz, vp, vp]),
        "gf_pools_merge_dev": (i32, [vp, vp, sz, vp, i32, i32, sz, i32, i32, i32, i32, vp, sz, vp, vp]),
        "gf_pools_pack_for_owners_v_dev": (i32, [vp, vp, vp, sz, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
        "gf_pools_merge_v_dev": (i32, [vp, vp, vp, vp, i32, i32, sz, i32, i32, i32, i32, vp, sz, vp, vp]),
        "gf_quick_check": (i32, [vp, C.c_char_p, vp, vp, sz, i32, vp, sz, szp]),
        "gf_quick_check_dev": (i32, [vp, vp, vp, vp, sz, sz, i32, vp, sz, vp]),
        "gf_overlap_evaluate": (i32, [vp, C.c_char_p, vp, vp, sz, vp, sz, vp, vp]),
        "gf_overlap_evaluate_dev": (i32, [vp, vp, vp, vp, vp, sz, vp, vp]),
        "gf_pick_anchored_dev": (i32, [vp, vp, vp, sz, vp, i32, vp, vp]),
        "gf_pick_anchored2_dev": (i32, [vp, vp, vp, sz, vp, i32, i32, vp, vp]),
        "gf_pick_anchored2_from_dev": (i32, [vp, vp, vp, sz, vp, i32, i32, vp, vp, vp]),
        "gf_bridging_reads": (i32, [vp, C.c_char_p, vp, vp, C.c_char_p, vp, vp, sz, i32, i32, vp]),
        "gf_merge_open_gaps_dev": (i32, [vp, vp, vp, sz, vp, vp, sz, vp, sz, vp, i32, i32, vp, vp, i32, vp]),
        "gf_count_kmers": (i32, [vp, vp, vp, sz, i32, i32, i32, vp, vp, sz, szp]),
        "gf_pool_keys_reset": (i32, [vp, vp]),
        "gf_pool_keys_from_screen_dev": (i32, [vp, vp, vp, sz, i32, vp, sz, vp]),
        "gf_pool_keys_from_tags_dev": (i32, [vp, vp, vp, vp, sz, vp, sz, vp, sz, vp]),
        "gf_build_pools_dev": (i32, [vp, vp, sz, i32, vp, vp, sz, vp, sz, vp, vp, vp]),
        "gf_dev_alloc": (i32, [vp, sz, C.POINTER(vp)]),
        "gf_dev_free": (i32, [vp, vp]),
        "gf_memcpy_h2d": (i32, [vp, vp, vp, sz]),
        "gf_memcpy_d2h": (i32, [vp, vp, vp, sz]),
        "gf_memset_dev": (i32, [vp, vp, i32, sz]),
        "gf_timing_enable": (i32, [vp, i32]),
        "gf_timing_read": (i32, [vp, i32, C.POINTER(C.c_double), u64p]),
        "gf_timing_reset": (i32, [vp]),
        "gf_synth_pairs_dev": (i32, [vp, vp, C.c_uint64, sz, vp, vp]),
        "gf_synth_layout": (i32, [vp, vp, vp, vp]),
        "gf_synth_truth": (i32, [vp, C.c_uint32, C.c_uint64, sz, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)   # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None
