/* TEST INFRASTRUCTURE — C restatement (oracle) of GAPPadder's recruit + local-assembly hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so; the product
 * never does.  Semantics and parity status: see the header of oracle/gp_oracle.py (a-1..a-5, a-7 pinned on
 * reference-generated fixtures; a-6 PARITY UNPINNED).  Plain C, straightforward algorithms (sorted arrays +
 * binary search, per-record loops over every gap) chosen to differ from the GPU implementation. */
#ifndef GP_ORACLE_H
#define GP_ORACLE_H
#include <stddef.h>
#include <stdint.h>

typedef struct { uint32_t scaffold, start, end, idx_in_scaffold; } or_gap;
typedef struct { uint32_t pos, mate_pos; int32_t tlen; uint32_t ref, mate_ref; uint16_t flag; uint8_t mapq, clipflag; uint64_t read; } or_alnrec;
typedef struct { uint32_t rec, gap; uint16_t kind, to_mate; } or_taghit;
typedef struct { uint32_t mate_scaffold, mate_pos, src_scaffold, src_gap; } or_dpos;
typedef struct { uint32_t gap, read; } or_hit;

/* collect_reads_for_gaps.py:68-263 on decoded records; hits in (rec, gap, kind) order. returns count (may exceed cap) */
size_t or_tag_alignments(const or_alnrec* recs, size_t n, const or_gap* gaps, size_t n_gaps, int insert_size, int sd,
                         int clip_dist, int anchor_mapq, or_taghit* out, size_t cap);
/* collect_discordant_low_mapq_reads.py:4-84; table sorted; hit.gap = row index; (rec,row) order */
size_t or_tag_low_mapq(const or_alnrec* recs, size_t n, const or_dpos* table, size_t n_rows, or_taghit* out, size_t cap);
/* north-star screen on ASCII reads (fixed length) and ASCII flanks; hits sorted (gap, read). threads: OpenMP */
size_t or_screen_reads(const char* reads_ascii, size_t n_reads, int read_len, const char* flank_ascii,
                       const uint64_t* flank_off, size_t n_gaps, int k, int min_hits, uint32_t max_gaps_per_kmer,
                       or_hit* out, size_t cap, int threads);
double or_screen_last_build_s(void); /* seconds the last or_screen_reads call spent on its flank k-mer table, before the per-read pass */
void or_set_threads(int n); /* OpenMP threads used by the parallel functions */
/* KmerUtils.cpp:61-69 */
uint64_t or_pack_kmer64(const char* seq, int k);
void or_unpack_reads(const uint8_t* packed, size_t n_reads, int read_len, char* ascii);
/* a-6 (PARITY UNPINNED, semantics in gp_oracle.c): counted canonical k-mers (kmc | kmc_dump), and the contigs of one pool */
size_t or_count_kmers(const char* reads, size_t n_reads, int L, int k, int min_count, uint64_t* hi, uint64_t* lo,
                      uint32_t* cnt, size_t cap);
size_t or_assemble_pool(const char* reads, size_t n_reads, int L, int k, int kv, int min_count, int min_contig,
                        uint32_t* n_nodes, uint32_t* length, uint32_t* cov_sum, size_t cap, char* seq_out, size_t seq_cap,
                        size_t* seq_need);
/* same with `simplify` rounds of tip clipping + bubble popping (Velvet's defaults, as defined in gp_oracle.c); 0 = raw unitigs */
size_t or_assemble_pool2(const char* reads, size_t n_reads, int L, int k, int kv, int min_count, int min_contig, int simplify,
                         uint32_t* n_nodes, uint32_t* length, uint32_t* cov_sum, size_t cap, char* seq_out, size_t seq_cap,
                         size_t* seq_need);
/* ... and with the tie-break mode of the error removal spelled out: 1 = "counts" (between equal coverage the side with fewer weak
 * nodes — k-mers seen at most min_count + 1 times — stays; or_assemble_pool2's mode), 0 = "none" (sequence order alone: nothing Velvet
 * could not have known, assemble_gaps.py:56-79) */
size_t or_assemble_pool3(const char* reads, size_t n_reads, int L, int k, int kv, int min_count, int min_contig, int simplify, int tiebreak,
                         uint32_t* n_nodes, uint32_t* length, uint32_t* cov_sum, size_t cap, char* seq_out, size_t seq_cap,
                         size_t* seq_need);
/* f-3: the contig merger's all-pairs k-mer prefilter (QuickCheckerContigsMatch, ContigsCompactor.cpp:1982-2095) over the node list
 * [c0, revcomp(c0), c1, ...]; pairs (i <= j) in order; returns their number (may exceed cap).  Contigs of >= 30 bases. */
typedef struct { double mismatch, indel, max_clip, frac_min_overlap, frac_loss, min_overlap, min_overlap_scaffold, relax; } or_ovl_params;
typedef struct { int32_t res, row_end, col_end, nclip, score, contained, merged_len, overlap, containment, first_goes_first; } or_ovl_result;
void or_overlap_evaluate(const char* s1, int n1, const char* s2, int n2, const or_ovl_params* pr, or_ovl_result* out);
size_t or_quick_check(const char* seqs, const uint64_t* off, size_t n, int k, uint32_t* out_i, uint32_t* out_j, size_t cap);
/* synthetic workload, definition in include/gf_synth.h (cfg = gf_synth_cfg) */
void or_synth_pairs(const void* cfg, uint64_t first_pair, size_t n_pairs, uint8_t* packed, or_alnrec* recs_or_null);
void or_synth_layout(const void* cfg, or_gap* gaps, char* flank_ascii, uint64_t* flank_off);
#endif
