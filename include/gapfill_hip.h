/* gapfill_hip.h — C ABI of libgapfill_hip.so: the MI355X (gfx950) implementation of GAPPadder's
 * read-recruitment + per-gap local-assembly hot path.
 *
 * The reference (simoncchu/GAPPadder) has no FFI for this path; its seams are process + file contracts
 * (SURVEY.md §8b).  Each entry point below names the reference interface whose arithmetic it replaces
 * (paths relative to the reference tree).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions: return 0 (GF_OK) or a negative GF_E_* code; never throws, never exits.  The caller owns
 * every buffer.  Plain entry points take HOST pointers and stage through HBM; the *_dev entry points take
 * DEVICE pointers, enqueue on the context's stream and do not synchronise (they are what bench.py times
 * with inputs already resident in HBM).  Output goes into caller-provided capacity; on overflow the call
 * returns GF_E_NOSPACE and *n_out holds the required element count.  One gf_ctx per device; a ctx is not
 * thread-safe; different ctxs may be driven from different threads/processes.
 */
#ifndef GAPFILL_HIP_H
#define GAPFILL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GF_OK 0
#define GF_E_INVAL (-1)    /* bad argument */
#define GF_E_NODEV (-2)    /* no usable HIP device / HIP runtime error (see gf_last_error) */
#define GF_E_NOMEM (-3)    /* host or device allocation failed */
#define GF_E_NOSPACE (-4)  /* output capacity too small; *n_out = required count */
#define GF_E_STATE (-5)    /* call order violated (e.g. screen before gf_set_gaps) */
#define GF_E_UNSUPPORTED (-6)
#define GF_E_FORMAT (-7)   /* malformed input file bytes (BGZF / BAM); see gf_last_error */

typedef struct gf_ctx gf_ctx;

/* One gap of the draft (one line of gap_positions.txt: gnrt_pos_true_seqs.py:54).  Gaps are passed grouped
 * by scaffold in file order; idx_in_scaffold is the reference's 1-based per-scaffold counter
 * (collect_reads_for_gaps.py:34-63, merge_reads.py:27-41). */
typedef struct {
    uint32_t scaffold;        /* index of the scaffold in the .fai (0-based) */
    uint32_t start;           /* 0-based position of the first N */
    uint32_t end;             /* position of the first upper-case ACGT after the run */
    uint32_t idx_in_scaffold; /* 1-based */
} gf_gap;

/* One alignment record = the 9 SAM columns the reference reads (collect_reads_for_gaps.py:76-91),
 * decoded: 32 bytes. */
typedef struct {
    uint32_t pos;      /* SAM POS (1-based, un-shifted as the reference uses it) */
    uint32_t mate_pos; /* SAM PNEXT */
    int32_t tlen;      /* SAM TLEN */
    uint32_t ref;      /* RNAME as .fai index; 0xFFFFFFFF for '*' */
    uint32_t mate_ref; /* RNEXT as .fai index ('=' -> same value as ref); 0xFFFFFFFF for '*' */
    uint16_t flag;     /* SAM FLAG */
    uint8_t mapq;      /* SAM MAPQ */
    uint8_t clipflag;  /* GapReadsCollector.is_clipped(CIGAR): +2 right S/H, +1 left S/H (:13-26) */
    uint64_t read;     /* caller's read id: 2*pair + (mate number - 1) */
} gf_alnrec;

/* kinds of a tagger hit = the 4th column of the reference's list lines */
#define GF_KIND_CLIP 0       /* collect_reads_for_gaps.py:119-123  -> the read's OWN left/right list */
#define GF_KIND_DISCORDANT 1 /* :126-150                           -> the MATE's list */
#define GF_KIND_UNMAP 2      /* :153-159                           -> the MATE's list */
#define GF_KIND_LOWMAPQ 3    /* collect_discordant_low_mapq_reads.py:74-79 -> the read's OWN list */

typedef struct {
    uint32_t rec;  /* index of the record in the batch */
    uint32_t gap;  /* gf_tag_alignments: index into the gf_set_gaps array; gf_tag_low_mapq: row of the table */
    uint16_t kind; /* GF_KIND_* */
    uint16_t to_mate; /* 1: the line goes to the list of the read's mate, 0: to its own list */
} gf_taghit;

/* One row of discordant_reads_pos.txt.sorted.txt (run_multi_threads_discordant.py:87-103). */
typedef struct {
    uint32_t mate_scaffold; /* scaffold index the discordant mate maps to */
    uint32_t mate_pos;
    uint32_t src_scaffold;
    uint32_t src_gap;       /* 1-based gap index inside src_scaffold */
} gf_dpos;

typedef struct {
    uint32_t gap;  /* index into the gf_set_gaps array */
    uint32_t read; /* index of the read in the batch */
} gf_hit;

/* ---- context ------------------------------------------------------------------------------------- */
int gf_init(int device_ordinal, gf_ctx** out);
void gf_destroy(gf_ctx* ctx);
const char* gf_strerror(int code);
const char* gf_last_error(gf_ctx* ctx);       /* text of the last HIP error seen by this ctx */
/* the filter kernels the last gf_screen_reads[_dev] call launched, comma separated, as a profiler names them without namespace and
 * argument list (e.g. "pf4_scatter_lines_kernel<3u, false>,pf4_probe_kernel,pf4_resolve_kernel,pf4_list_kernel"): lets a caller
 * tie a committed rocprofv3 summary to the build that is running */
const char* gf_screen_kernels(gf_ctx* ctx);
int gf_set_stream(gf_ctx* ctx, void* hip_stream); /* adopt a caller's hipStream_t (NULL: back to the ctx's own) */
int gf_sync(gf_ctx* ctx);
/* Two contexts on one device run their work on two streams (e.g. the k-mer screen on one, the alignment tagger on the
 * other: they are independent until the pools are built).  gf_stream_wait makes everything enqueued on `waiter` after
 * this call wait for everything enqueued on `producer` before it — no host synchronisation. */
int gf_stream_wait(gf_ctx* waiter, gf_ctx* producer);
/* One-shot: everything enqueued on `waiter` after the NEXT gf_screen_reads_dev call on `producer` waits until that call's FILTER pass
 * has finished — not for its verification pass.  The filter owns every CU's LDS and most of the memory system, the verification pass
 * is a latency-bound look-up kernel: a bandwidth-bound stream (the alignment tagger) overlaps the second, not the first. */
int gf_stream_wait_after_filter(gf_ctx* waiter, gf_ctx* producer);
/* options: "max_gaps_per_kmer" (0 = unlimited; flank k-mers shared by more gaps are dropped from the index),
 * "bitmap_log2" (size of the screen's level-1 16-mer bitmap, 0 = automatic), "index_host" (1: build the flank k-mer index with
 * the host comparator instead of the device kernels — same index up to slot order; a test aid, refused unless the environment
 * has GF_DIAGNOSTICS set).
 * Ablation / diagnostic switches (results never change): "screen_variant" (0 automatic, 9 plain, 13 pipelined,
 * 16 partitioned with 256 buckets and 4-byte pairs, 17 the same with unaligned pair runs in its first pass), "screen_ext" (1: the
 * partitioned filter drops seeds whose neighbouring bases are none of the flanks'; 0: 16-base seeds as they are),
 * "screen_verify_ext" (1: seed-and-extend verification when min_hits == 1), "screen_verify_gate",
 * "screen_verify_batch", "screen_stream_policy", "screen_lds_log2_max", "screen_pf4_cap8",
 * "asm_keyslot", "asm_precount", "asm_ranked", "asm_lds_pool_kb", "asm_threads" (threads per gap: 1024 / 512 / 256; 0 = automatic:
 * 512 — two gaps per CU — when there are >= 8 gaps per CU and asm_max_pool_reads says that most pools fit half a CU's LDS, the gaps
 * that do not are handed to a second launch with 1024), "asm_stats_ptr" (device u64[4] the assembly adds its window / k-mer / survivor / node counts to), "asm_dbg_ptr".
 * Tagger: "tag_light" (1: one-wave workgroups that read the coarse bin map through L1/L2 instead of staging it in LDS — same hits;
 * for a pipeline that runs the tagger on a second context beside the k-mer filter, whose workgroups own most of every CU's LDS).
 * Assembly: "asm_tiebreak" (error removal between branches of EQUAL coverage: 1 (default) = the branch with fewer weak nodes — k-mers seen
 * at most min_count + 1 times — wins, then sequence order; 0 = sequence order alone: the reference-shaped mode, nothing that Velvet
 * could not have known, since cvtFaToFq drops the counts before Velvet reads a k-mer, assemble_gaps.py:56-79),
 * "asm_simplify" (rounds of tip clipping + bubble popping — Velvet's defaults, which the reference runs with
 * (assemble_gaps.py:117); default 8 (where the rounds have converged), 0 = raw unitigs), "asm_max_pool_reads" (device variants: upper bound on the rows of one
 * pool; the assembly workspace is then one slice of that size per workgroup instead of one per pool row — a pool beyond the bound
 * sets its gap_error; 0 = no bound), "asm_sweep" (1: gf_assemble_multi_dev runs the k list 31/29, 41/39, 51/49 as ONE launch in which a
 * workgroup assembles a gap three times — same contigs, a gap's contigs in (k, kv) order; default 0 = one launch per pair, which measured
 * 1 % faster at C5). */
int gf_set_option(gf_ctx* ctx, const char* name, long value);

/* ---- gaps + flanks (gnrt_pos_true_seqs.py:12-100 defines them; host-side there and here) ----------- */
/* flank_ascii: concatenated flank sequences, left then right per gap; flank_off[2*g], [2*g+1], ... are the
 * start offsets, flank_off[2*n_gaps] the total length.  Bases other than upper-case ACGT break k-mers. */
int gf_set_gaps(gf_ctx* ctx, const gf_gap* gaps, size_t n_gaps, uint32_t n_scaffolds,
                const char* flank_ascii, const uint64_t* flank_off);

/* ---- a-7: 2-bit packing (KmerUtils.cpp:22-58 layout: A=00 C=01 G=10 T=11, base i MSB-first) --------- */
/* Packs n_reads fixed-length reads (ASCII, read_len bases each, contiguous) into ceil(read_len/4) bytes per
 * read, base i in byte i/4 at bits 7-2(i%4)..6-2(i%4); non-ACGT -> A (KmerUtils.cpp:25) and, when n_mask is
 * non-NULL, bit (i%32) of word n_mask[r*ceil(read_len/32) + i/32] is set. */
int gf_pack_reads(const char* ascii, size_t n_reads, int read_len, uint8_t* packed, uint32_t* n_mask);
size_t gf_packed_read_bytes(int read_len);

/* ---- ingest: FASTQ TEXT -> the packed layout above, on the device.  `text` is the content of a FASTQ file
 * (or a chunk that starts at a record boundary): strict 4-line records as the reference's line machine assumes
 * (run_multi_threads_discordant.py:205-232); the last line may lack its newline; CRLF tolerated.  Read r = sequence line of
 * record r, padded with masked A (N) up to read_len.  hdr_begin[r] = byte offset of record r's '@' line (ids are cut from the
 * host's copy of the text: first whitespace token, up to '/', :212-214); needs cap_reads + 1 entries.
 * *n_reads = records found (may exceed cap_reads: GF_E_NOSPACE from the host variant, status bit 4 from the device variant).
 * status bits: 1 = a sequence line longer than read_len was truncated, 2 = trailing partial record ignored, 4 = capacity,
 * 8 = the quality line of the last record ends without a newline (a whole file: fine; a piece of one: that record may be cut). */
int gf_fastq_pack(gf_ctx* ctx, const char* text, size_t n_bytes, int read_len, uint8_t* packed, size_t cap_reads,
                  uint32_t* n_mask_or_null, uint64_t* hdr_begin_or_null, size_t* n_reads, uint32_t* status);
/* device variant: d_n_reads = u64, d_status = u32 (both written by the call; it synchronises the stream once to size the
 * second pass). */
int gf_fastq_pack_dev(gf_ctx* ctx, const void* d_text, size_t n_bytes, int read_len, void* d_packed, size_t cap_reads,
                      void* d_n_mask_or_null, void* d_hdr_begin_or_null, void* d_n_reads, void* d_status);

/* SAM TEXT (alignment lines as `samtools view` prints them) -> gf_alnrec, parsed on the device exactly as the reference reads
 * a line (collect_reads_for_gaps.py:76-91: whitespace-split, columns 0-8; clipflag per is_clipped :13-26): '@' lines and
 * lines with fewer than nine fields yield no record; the others are numbered in line order (rec.read = record index).
 * names_blob/name_off: the scaffold names of the .fai in order (n_names + 1 offsets); RNAME/RNEXT decode to the name's index,
 * '=' to RNAME's index, an unknown name to 0xFFFFFFFF.  line_begin[r] = byte offset of record r's line (QNAME and the other
 * text columns are cut from the host's copy for the few records that produce a hit).  GF_E_NOSPACE: *n_recs = needed. */
int gf_sam_pack(gf_ctx* ctx, const char* text, size_t n_bytes, const char* names_blob, const uint32_t* name_off, size_t n_names,
                gf_alnrec* recs, size_t cap_recs, uint64_t* line_begin_or_null, size_t* n_recs);

/* BGZF (the blocked gzip container of BAM, SAMv1 §4.1) inflated on the device, one wavefront per block, every block's CRC-32
 * and ISIZE checked (GF_E_FORMAT + gf_last_error on a mismatch).  `bgzf` = file bytes starting at a block boundary; only whole
 * blocks are taken: *n_consumed = bytes used (bring the rest back in front of the next chunk).  The inflated stream is
 * `carry` (bytes the previous gf_bam_pack call left unconsumed: a partial record) followed by the blocks' contents; it STAYS ON
 * THE DEVICE for gf_bam_pack / gf_bam_fetch; out_or_null (capacity `cap`; GF_E_NOSPACE with *n_out = needed) additionally
 * receives a host copy of all of it (NULL: none — the usual case, see gf_bam_fetch).
 * Replaces the `samtools view` pipe of run_multi_threads_collect_reads.py:30-32 / run_multi_threads_discordant.py:131-133. */
int gf_bgzf_inflate(gf_ctx* ctx, const uint8_t* bgzf, size_t n_bytes, const uint8_t* carry, size_t n_carry, uint8_t* out_or_null,
                    size_t cap, size_t* n_out, size_t* n_consumed);
/* BAM alignment records (SAMv1 §4.2) -> gf_alnrec, the same values gf_sam_pack gives for the `samtools view` line of the
 * record (POS/PNEXT 1-based, clipflag from the first/last CIGAR operation, rec.read = record index in stream order).
 * stream_or_null: the inflated bytes (NULL: the n_bytes left on the device by the last gf_bgzf_inflate); `first` = offset of
 * the first alignment record (after the header in the first chunk, 0 afterwards); ref_map[refID] = .fai index of that BAM
 * reference or 0xFFFFFFFF.  Only complete records are taken: *n_consumed = offset of the first incomplete one (== n_bytes when
 * none).  rec_begin[r] = offset of record r's block_size field.  GF_E_NOSPACE: *n_recs = needed. */
int gf_bam_pack(gf_ctx* ctx, const uint8_t* stream_or_null, size_t n_bytes, size_t first, const uint32_t* ref_map, size_t n_ref,
                gf_alnrec* recs, size_t cap_recs, uint64_t* rec_begin_or_null, size_t* n_recs, size_t* n_consumed);

/* gf_tag_alignments / gf_tag_low_mapq on the records the last gf_bam_pack left on the device (no second trip of the 32-byte
 * records over PCIe); GF_E_STATE when there are none. */
int gf_tag_alignments_bam(gf_ctx* ctx, int insert_size, int sd, int clip_dist, int anchor_mapq, gf_taghit* out, size_t cap,
                          size_t* n_out);
int gf_tag_low_mapq_bam(gf_ctx* ctx, const gf_dpos* table, size_t n_rows, gf_taghit* out, size_t cap, size_t* n_out);
/* Slices [begin[i], end[i]) of the inflated stream that gf_bgzf_inflate / gf_bam_pack left on the device, packed back to back
 * into dst (one small gather + one copy): the host asks for the header and for the few records that produced a hit instead of
 * copying every inflated byte back.  *n_bytes = total size; GF_E_NOSPACE when cap is smaller. */
int gf_bam_fetch(gf_ctx* ctx, const uint64_t* begin, const uint64_t* end, size_t n, uint8_t* dst, size_t cap, size_t* n_bytes);

/* ---- files -> a library RESIDENT in HBM (the device pipeline of the CLI, gappadder_amd/device_collect.py; csrc/resident.hip).
 * The reference joins alignments and reads by NAME (list lines carry QNAME, collect_reads_for_gaps.py:119-159; the FASTQ line machine
 * looks the id of every record up in {readId -> set(gapKey)}, run_multi_threads_discordant.py:153-185, 209-241); the device pipeline
 * addresses reads by index (gf_alnrec.read = 2 * FASTQ record number + mate), which these calls establish once per library. */
/* 64-bit hash of the id of every record of a FASTQ chunk on the device — the id as the reference cuts it: first whitespace token of the
 * header line, up to the first '/', first character ('@') dropped (run_multi_threads_discordant.py:212-214) — and the length of the
 * longest sequence line (atomicMax into *d_max_len, u32: the caller zeroes it).  d_hdr_begin as written by gf_fastq_pack_dev. */
int gf_fastq_index_dev(gf_ctx* ctx, const void* d_text, size_t n_bytes, const void* d_hdr_begin, size_t n_reads, void* d_id_hash /* u64[n_reads] */,
                       void* d_max_len);
/* gf_bam_pack for a resident library: the records of the inflated stream gf_bgzf_inflate left on the device are APPENDED to the caller's
 * device array d_recs at index rec_base (capacity rec_cap records), with
 *   d_qhash[rec]     the same 64-bit hash of the record's QNAME (raw, as the reference compares it), or NULL;
 *   d_names / d_name_off[rec]  the QNAME bytes back to back in a caller's arena (appended at name_base, capacity name_cap bytes) and the
 *                    offset of each (d_name_off[rec_base + n] = end of the last), or NULL: the host later fetches the names of the few
 *                    records that produce a hit (gf_fetch_slices) for the list files of the reference's file contract;
 *   d_ref_seen[scaffold] |= 1 for a record on that scaffold, |= 2 for a MAPQ-0 record (u32 per .fai scaffold, caller zeroes), or NULL:
 *                    which per-scaffold list files the reference would have opened (collect_reads_for_gaps.py:93-102);
 *   d_rec_begin[n]   for the n-th record of THIS chunk (not offset by rec_base; room for n_bytes / 36 + 1 entries) the offset of its
 *                    block_size field in the inflated stream, or NULL: with the FLAGs in d_recs the caller selects records (e.g. both
 *                    mates unmapped) and fetches their bytes while the stream is still there (gf_bam_fetch).
 * gf_alnrec.read is left 0xFFFFFFFF ("no read") until gf_read_join_dev.  *n_recs / *n_name_bytes = what the chunk holds; GF_E_NOSPACE
 * (nothing written) when a capacity is too small: grow and call again, the stream is still there. */
int gf_bam_append_dev(gf_ctx* ctx, size_t n_bytes, size_t first, const uint32_t* ref_map, size_t n_ref, void* d_recs, size_t rec_base, size_t rec_cap,
                      void* d_qhash_or_null, void* d_names_or_null, size_t name_base, size_t name_cap, void* d_name_off_or_null,
                      void* d_ref_seen_or_null, size_t n_scaffolds, void* d_rec_begin_or_null, size_t* n_recs, size_t* n_name_bytes,
                      size_t* n_consumed);
/* the join: gf_alnrec.read = 2 * (number of the FASTQ record whose id hash equals the record's QNAME hash) + (FLAG & 0x40 ? 0 : 1), or
 * 0xFFFFFFFF when there is none (such a record recruits nothing).  d_id_hash = the ids of ONE mate file (both files of a pair carry the
 * same ids in the same order: the caller checks that).  d_stats (u32[4], written): [0] ids that occur more than once, [1] records
 * without a read.  By hash: the caller verifies the names of the records that matter (the hits) and fails loudly on a mismatch. */
int gf_read_join_dev(gf_ctx* ctx, const void* d_id_hash, size_t n_ids, void* d_recs, const void* d_qhash, size_t n_recs, void* d_stats);
/* gf_bam_fetch on any device buffer: slices [begin[i], end[i]) of d_src back to back into the host buffer dst */
int gf_fetch_slices(gf_ctx* ctx, const void* d_src, size_t src_len, const uint64_t* begin, const uint64_t* end, size_t n, uint8_t* dst, size_t cap,
                    size_t* n_bytes);
/* d_dst[i] = row d_ids[i] of d_src for i < min(*d_n, cap) (rows of row_bytes, a multiple of 4; *d_n a device u64 — e.g. d_pool_off[n_gaps];
 * an id beyond n_src_rows yields all-ones): the N masks of pooled reads from the read ids gf_build_pools_dev reports */
int gf_gather_rows_dev(gf_ctx* ctx, const void* d_src, size_t n_src_rows, size_t row_bytes, const void* d_ids, const void* d_n, size_t cap,
                       void* d_dst);

/* ---- host-side text of the reference's file contract, for the records that leave the device pipeline.  No GPU work: plain C loops
 * over bytes the caller already holds (a run formats one record per recruited read; the reference does it in its FASTQ line machine,
 * run_multi_threads_discordant.py:209-241, and through `samtools view`).  ctx may be NULL (it only carries the error text). */
/* BAM alignment records, raw as in the inflated stream (record i starts at blob[rec_begin[i]] with its block_size field) ->
 *   sam: the eleven mandatory SAM columns as `samtools view` prints them, one line per record: the builtin stand-in of
 *        `samtools view -f 12` (collect_both_unmapped_reads.py:14-22)
 *   fq:  the same records in that module's FASTQ form: `@{QNAME}_2` when FLAG > 128 (a comparison, :26) else `@{QNAME}_1`, SEQ, `+`, QUAL
 *        (:24-33); NULL: sized only.
 * ref_names = the header's n_ref reference names, NUL-terminated, back to back.  *sam_len / *fq_len = bytes needed; GF_E_NOSPACE when a
 * given buffer is smaller (what was written is then incomplete), GF_E_FORMAT for bytes that are no BAM record. */
int gf_bam_records_text(gf_ctx* ctx_or_null, const uint8_t* blob, size_t blob_len, const uint64_t* rec_begin, size_t n_recs, const char* ref_names,
                        size_t n_ref, char* sam, size_t sam_cap, size_t* sam_len, char* fq_or_null, size_t fq_cap, size_t* fq_len);
/* FASTQ records [begin[i], end[i]) of file which[i] — given as images in memory (files) or, when files is NULL, as open descriptors read
 * with one pread per record (fds) — as the reference re-writes them into the per-gap files (run_multi_threads_discordant.py:212-221):
 * `@{id}{suffix[which[i]]}` — id = the header's first word up to its first '/', without the '@' —, the sequence line, a bare `+`, the
 * quality line, trailing white space dropped.  out_end[i] = where record i ends in `out`; ids_or_null / ids_end_or_null: the bare ids
 * back to back.  *out_len / *ids_len = bytes needed, GF_E_NOSPACE when a buffer is smaller. */
int gf_fastq_records_text(gf_ctx* ctx_or_null, const uint8_t* const* files_or_null, const int* fds_or_null, const uint64_t* file_len, size_t n_files,
                          const uint64_t* begin, const uint64_t* end, const uint8_t* which, const char* const* suffix, size_t n, char* out, size_t cap,
                          uint64_t* out_end,
                          char* ids_or_null, size_t ids_cap, uint64_t* ids_end_or_null, size_t* out_len, size_t* ids_len);

/* ---- north-star flank-k-mer screen ("flank-k-mer lookup to tag reads") ------------------------------
 * Emits (gap, read) for every read that has >= min_hits k-mer positions whose canonical k-mer occurs in the
 * gap's flank k-mer set (predicate shape of IsReadContainingFreqKmers, KmerUtils.cpp:215-241, on canonical
 * k-mers, per gap).  16 <= k <= 64.  Host variant returns hits sorted by (gap, read). */
int gf_screen_reads(gf_ctx* ctx, const uint8_t* packed_reads, const uint32_t* n_mask_or_null, size_t n_reads,
                    int read_len, int k, int min_hits, gf_hit* out, size_t cap, size_t* n_out);
/* device variant: d_out order is unspecified; *d_n_out (device u32, zeroed by the call) = hits produced. */
int gf_screen_reads_dev(gf_ctx* ctx, const void* d_packed_reads, const void* d_n_mask_or_null, size_t n_reads,
                        int read_len, int k, int min_hits, void* d_out, size_t cap, void* d_n_out);
/* The device variant cannot return GF_E_UNSUPPORTED for the one thing the verification may not be able to finish — a read with more
 * than 15 000 (k-mer position, gap) matches, i.e. a low-complexity read against hundreds of flanks that share its k-mers: such a read
 * is counted and its hits are incomplete.  This reads the counter of the LAST gf_screen_reads_dev call on this context (it waits for
 * the context's stream); 0 = every read was verified in full.  The remedy on such drafts is the repeat mask, option
 * "max_gaps_per_kmer". */
int gf_screen_last_overflow(gf_ctx* ctx, size_t* n_reads_dropped);

/* ---- a-2: alignment-record tagger (GapReadsCollector.parse_reads_fall_in_gaps_one_scaffold[_short_is],
 * collect_reads_for_gaps.py:68-263; mode switch at IS >= 750, :275).  Host variant: sorted by
 * (rec, gap, kind). */
int gf_tag_alignments(gf_ctx* ctx, const gf_alnrec* recs, size_t n, int insert_size, int sd, int clip_dist,
                      int anchor_mapq, gf_taghit* out, size_t cap, size_t* n_out);
int gf_tag_alignments_dev(gf_ctx* ctx, const void* d_recs, size_t n, int insert_size, int sd, int clip_dist,
                          int anchor_mapq, void* d_out, size_t cap, void* d_n_out);

/* ---- a-3: second hop (collect_discordant_low_mapq_reads.py:4-84).  table = rows sorted by
 * (mate_scaffold, mate_pos, src_scaffold, src_gap), duplicates kept; a MAPQ==0 record at POS p on scaffold s
 * is linked to the LAST position q of s with q-199 <= p <= q+299 and emits one hit per row of q
 * (hit.gap = row index).  Host variant: sorted by (rec, row). */
int gf_tag_low_mapq(gf_ctx* ctx, const gf_alnrec* recs, size_t n, const gf_dpos* table, size_t n_rows,
                    gf_taghit* out, size_t cap, size_t* n_out);
int gf_tag_low_mapq_dev(gf_ctx* ctx, const void* d_recs, size_t n, const gf_dpos* table, size_t n_rows,
                        void* d_out, size_t cap, void* d_n_out);

/* ---- a-6: per-gap local assembly (run_assembly, assemble_gaps.py:82-136: `kmc -k{k}` | `kmc_dump` | `velveth {kv}` |
 * `velvetg -min_contig_lgth 40`).  Pools are packed reads (gf_pack_reads layout) concatenated per gap: pool g owns
 * reads pool_off[g] .. pool_off[g+1]-1.  The k-mer counting and graph arithmetic of the reference live in un-vendored
 * KMC and Velvet, so the semantics are DEFINED by this build (DESIGN.md "Assembly semantics", oracle/gp_oracle.c):
 * canonical k-mers with count >= min_count; nodes = their canonical kv-mers (kv odd, 15 <= kv < k <= 64); contigs =
 * unitigs of >= min_contig bases, each reported once in the orientation min(seq, revcomp(seq)). */
typedef struct {
    uint32_t gap;      /* index of the pool */
    uint16_t k, kv;
    uint32_t n_nodes;  /* kv-mers in the contig (Velvet's NODE_n_length_{n_nodes}) */
    uint32_t length;   /* bases = n_nodes + kv - 1 */
    uint32_t cov_sum;  /* sum over the contig's nodes of their multiplicity among the surviving k-mers (cov = cov_sum / n_nodes) */
    uint32_t reserved;
    uint64_t seq_off;  /* offset of the contig's ASCII bases in the sequence buffer */
} gf_contig;

/* host variant: every (k_list[i], kv_list[i]) pair in turn (assemble_gaps.py:87-122); contigs sorted by
 * (gap, pair index, length descending, sequence ascending).  Returns GF_E_NOSPACE with the needed counts in
 * *n_contigs / *seq_len when a capacity is too small. */
int gf_assemble(gf_ctx* ctx, const uint8_t* pool_packed, const uint32_t* pool_n_mask_or_null, const uint64_t* pool_off,
                size_t n_pools, int read_len, const int* k_list, const int* kv_list, int n_k, int min_count, int min_contig,
                gf_contig* contigs, size_t contig_cap, size_t* n_contigs, char* seq, size_t seq_cap, size_t* seq_len);
/* device variant, one (k, kv) pair: total_reads = pool_off[n_pools] (upper bound used to size the workspace);
 * d_n_contigs: u32, d_seq_len: u64, d_gap_error: n_pools x u32 (1 = that pool overflowed an internal table).
 * Contig order is unspecified. */
int gf_assemble_dev(gf_ctx* ctx, const void* d_pool_packed, const void* d_pool_n_mask_or_null, const void* d_pool_off,
                    size_t n_pools, size_t total_reads, int read_len, int k, int kv, int min_count, int min_contig,
                    void* d_contigs, size_t contig_cap, void* d_n_contigs, void* d_seq, size_t seq_cap, void* d_seq_len,
                    void* d_gap_error);
/* every (k_list[i], kv_list[i]) pair of run_assembly's loop (assemble_gaps.py:87-122) in one call: one launch per pair, all
 * appending to the same contig list (gf_contig.k / .kv tell the pairs apart); counters and error flags as gf_assemble_dev. */
int gf_assemble_multi_dev(gf_ctx* ctx, const void* d_pool_packed, const void* d_pool_n_mask_or_null, const void* d_pool_off,
                          size_t n_pools, size_t total_reads, int read_len, const int* k_list, const int* kv_list, int n_k,
                          int min_count, int min_contig, void* d_contigs, size_t contig_cap, void* d_n_contigs, void* d_seq,
                          size_t seq_cap, void* d_seq_len, void* d_gap_error);
/* What the LAST assembly launch group of this context did (the last (k, kv) pair of a multi-k call): threads per gap of its main launch
 * (1024: one gap per CU; 512: two, each with half of the CU's LDS), the gaps that left it for the middle launch (1 024 threads: pools
 * too deep for, or graphs too large for, half a CU's LDS) and the pools beyond asm_max_pool_reads that the last launch took.
 * Synchronises the stream.  Diagnostic: results never depend on it. */
int gf_assemble_last_launch(gf_ctx* ctx, int* threads_per_gap, uint32_t* to_middle, uint32_t* to_last);

/* counted canonical k-mers of ONE pool, ascending (what `kmc_dump -ci0` lists after `kmc -k{k}`, assemble_gaps.py:96-102).
 * kmers: 2 x uint64 per k-mer (hi, lo), left-aligned KmerUtils layout; counts capped at 10^7 (-cs10000000). */
int gf_count_kmers(gf_ctx* ctx, const uint8_t* pool_packed, const uint32_t* pool_n_mask_or_null, size_t n_reads,
                   int read_len, int k, int min_count, uint64_t* kmers, uint32_t* counts, size_t cap, size_t* n_out);

/* One-pass form of the two hops for device-resident pipelines: the tagger also compacts every MAPQ==0 record (2 % of a
 * typical BAM) into {pos, ref, record index}, and the second hop scans that list instead of re-reading every record
 * (the reference makes a second full `samtools view` pass, run_multi_threads_discordant.py:125-138).  Same hits. */
typedef struct {
    uint32_t pos;
    uint32_t ref;
    uint32_t rec;
} gf_lowrec;
int gf_tag_alignments_low_dev(gf_ctx* ctx, const void* d_recs, size_t n, int insert_size, int sd, int clip_dist,
                              int anchor_mapq, void* d_out, size_t cap, void* d_n_out, void* d_low /* gf_lowrec */,
                              size_t low_cap, void* d_n_low /* u32 */);
/* The same pass over a library that keeps a KEY COLUMN beside its records: 8 bytes per record, {pos, ref | (mapq == 0) << 31}, n + 1 entries
 * (gf_alnrec_keys_dev builds it from the records, once, at ingest).  99 % of the records of a BAM lie far from every gap and are decided by
 * (scaffold, position) alone — the reference's `focal_region.has_key(POS)` (collect_reads_for_gaps.py:104) —, the MAPQ-0 by-product needs one
 * more bit: the tagger streams the key column (a quarter of the bytes) and fetches the 32-byte record only of what passes its bin maps.
 * Same hits, same MAPQ-0 list as gf_tag_alignments_low_dev (order unspecified in both). */
int gf_alnrec_keys_dev(gf_ctx* ctx, const void* d_recs, size_t n, void* d_keys /* 8 * (n + 1) bytes */);
int gf_tag_alignments_keys_dev(gf_ctx* ctx, const void* d_recs, const void* d_keys, size_t n, int insert_size, int sd, int clip_dist,
                               int anchor_mapq, void* d_out, size_t cap, void* d_n_out, void* d_low /* gf_lowrec or NULL */, size_t low_cap,
                               void* d_n_low /* u32 or NULL */);
int gf_tag_low_mapq_compact_dev(gf_ctx* ctx, const void* d_low, const void* d_n_low, size_t low_cap, const gf_dpos* table,
                                size_t n_rows, void* d_out, size_t cap, void* d_n_out);
/* The second-hop table itself on the device (replaces collect_discordant_regions_v2 + sort(1) + the per-scaffold split,
 * run_multi_threads_discordant.py:19-122): one row per DISCORDANT tagger hit whose mate lies on a known scaffold,
 * {mate_ref, mate_pos of the hit's record; scaffold, idx_in_scaffold of the hit's gap}, sorted by (mate scaffold, mate
 * position) with one radix sort.  d_rows (gf_dpos[row_cap]) and d_row_gap (u32[row_cap]: index of each row's gap, for
 * gf_pool_keys_from_second_hop_dev) are the caller's device buffers; *d_n_rows (device u32) = rows found — more than row_cap
 * means the table is truncated (size row_cap from the number of tagger hits).  Rows with equal (scaffold, position) keep no
 * particular order (the reference's -k3n -k4n): they produce the same hits. */
int gf_second_hop_table_dev(gf_ctx* ctx, const void* d_recs, const void* d_taghits, const void* d_n_taghits, size_t hit_cap,
                            void* d_rows, void* d_row_gap, size_t row_cap, void* d_n_rows);
/* Multi-GPU runs shard the reads, but the second hop links a MAPQ-0 record to discordant mates of ANY read: every rank builds the
 * rows of its records (gf_second_hop_table_dev), the ranks all-gather rows / row gaps / counts (fixed part_cap slots), and this
 * call concatenates the n_parts slots and sorts them again — the table of a single-process run up to the order of equal keys.
 * d_rows_all: gf_dpos [n_parts][part_cap]; d_row_gap_all: u32 [n_parts][part_cap]; d_n_rows_all: u32 [n_parts]. */
int gf_second_hop_table_merge_dev(gf_ctx* ctx, const void* d_rows_all, const void* d_row_gap_all, const void* d_n_rows_all, int n_parts,
                                  size_t part_cap, void* d_rows, void* d_row_gap, size_t row_cap, void* d_n_rows);
/* second hop over the compacted MAPQ-0 list against a DEVICE table (d_rows / d_n_rows as written by gf_second_hop_table_dev,
 * or any sorted gf_dpos array in HBM); hit.gap = row index, as with gf_tag_low_mapq */
int gf_tag_low_mapq_table_dev(gf_ctx* ctx, const void* d_low, const void* d_n_low, size_t low_cap, const void* d_rows,
                              const void* d_n_rows, size_t row_cap, void* d_out, size_t cap, void* d_n_out);

/* ---- a-4 / a-5 on the device: per-gap read pools.  The reference joins read IDs against whole FASTQ files
 * ({readId -> set(gapKey)}, run_multi_threads_discordant.py:153-185; stream + append, :209-241, 283-316; `cat` across
 * libraries, merge_reads.py:43-51).  With reads addressed by index (read = 2*pair + mate) that is the SET of (gap, read)
 * keys, per gap ordered by (mate, pair) = left-file stream order then right-file stream order, and a gather.
 * Keys are uint64 (gap << 32 | read); *d_n_keys is a device u32 that the key producers advance. */
int gf_pool_keys_reset(gf_ctx* ctx, void* d_n_keys);
/* screen hits -> keys; pairs != 0 also adds each hit's mate ("pulls candidate read pairs", north_star) */
int gf_pool_keys_from_screen_dev(gf_ctx* ctx, const void* d_hits, const void* d_n_hits, size_t hit_cap, int pairs,
                                 void* d_keys, size_t key_cap, void* d_n_keys);
/* tagger hits -> keys (target read = the record's read, or its mate when to_mate is set: the reference writes
 * discordant/unmap lines to the MATE's list, collect_reads_for_gaps.py:126-159).  For second-hop hits pass the row
 * table so that hit.gap (a row) resolves to its (src_scaffold, src_gap). */
int gf_pool_keys_from_tags_dev(gf_ctx* ctx, const void* d_recs, const void* d_taghits, const void* d_n_taghits,
                               size_t hit_cap, const gf_dpos* lowmapq_table_or_null, size_t n_rows, void* d_keys,
                               size_t key_cap, void* d_n_keys);
/* the three producers above in ONE launch without atomics (the counts are device words, so every key has a fixed place: screen
 * hits and their mates, then tagger hits, then second-hop hits); writes *d_n_keys = number of keys (no reset needed).
 * d_taghits / d_hophits may be NULL. */
int gf_pool_keys_all_dev(gf_ctx* ctx, const void* d_hits, const void* d_n_hits, size_t hit_cap, int pairs, const void* d_recs,
                         const void* d_taghits, const void* d_n_taghits, size_t taghit_cap, const void* d_hophits,
                         const void* d_n_hophits, size_t hophit_cap, const void* d_row_gap, void* d_keys, size_t key_cap,
                         void* d_n_keys);
/* second-hop hits against a device table: hit.gap (a row) resolves through d_row_gap (gf_second_hop_table_dev) */
int gf_pool_keys_from_second_hop_dev(gf_ctx* ctx, const void* d_recs, const void* d_taghits, const void* d_n_taghits,
                                     size_t hit_cap, const void* d_row_gap, void* d_keys, size_t key_cap, void* d_n_keys);
/* keys -> d_pool_off (n_gaps+1 x u64), d_pool_packed (pool_cap_reads reads), d_pool_read_ids (u32 per pooled read, or
 * null).  A gap may have any number of keys (up to 4 096 are sorted in LDS, longer lists in place in global memory: slower, never
 * dropped — the reference has no bound either, run_multi_threads_discordant.py:209-241).  *d_error (u32): bit 31 set =
 * d_pool_off[n_gaps] > pool_cap_reads, the pool buffer was too small (reads beyond it are not written; d_pool_off stays exact). */
int gf_build_pools_dev(gf_ctx* ctx, const void* d_packed_reads, size_t n_reads, int read_len, const void* d_keys,
                       const void* d_n_keys, size_t key_cap, void* d_pool_packed, size_t pool_cap_reads, void* d_pool_off,
                       void* d_pool_read_ids, void* d_error);

/* ---- pools of several sources -> one pool per gap.  Two uses, one primitive:
 *  (a-5) library merge, merge_reads.py:43-51 (`cat` of the per-library gap_reads/{id}.fastq in library order): build one pool
 *        array per library (gf_build_pools_dev into slot l of a [n_lib][cap_rows] buffer), gf_pool_counts_dev per library, then
 *        gf_pools_merge_dev(n_lib, n_src_ranks = 1, rank 0 of world 1);
 *  (§8e) the one exchange step of a multi-GPU run: every rank regroups its pools by OWNER rank (gf_pools_pack_for_owners_dev:
 *        owner(g) = (g / batch) % world; slot (owner * n_lib + lib) of the send buffer holds that owner's gaps' rows in gap
 *        order), the ranks all-gather the per-gap counts and all-to-all the slots (RCCL: equal-sized slots, no host sizes), and
 *        every owner merges its gaps' rows (gf_pools_merge_dev: libraries in order, inside a library the source ranks in order
 *        — with contiguous read shards that is the order of a single-process run over all reads; the reference assembles each gap
 *        exactly once from all of its reads, assemble_gaps.py:296-299).
 * Row = one packed read (gf_packed_read_bytes(read_len) bytes).  d_cnt = u32 per gap.  *d_error (u32, not reset by these calls)
 * gets bit 30 when a send slot, bit 31 when the merged buffer is too small (rows beyond the capacity are dropped). */
int gf_pool_counts_dev(gf_ctx* ctx, const void* d_pool_off /* u64[n_gaps+1] */, size_t n_gaps, void* d_cnt);
int gf_pools_pack_for_owners_dev(gf_ctx* ctx, const void* d_pool_packed, const void* d_pool_off, size_t n_gaps, int read_len,
                                 int world, int batch, int lib, int n_lib, void* d_send /* [world][n_lib][cap_rows] rows */,
                                 size_t cap_rows, void* d_cnt /* u32[n_gaps]: this library's rows per gap */, void* d_error);
/* d_src: [n_src_ranks][n_lib][cap_rows] rows (slot (r, l) = rows of MY gaps from source rank r, library l, in gap order);
 * d_cnt: u32 [n_src_ranks][n_lib][n_gaps] (the all-gathered pack counts; for a local library merge the per-library counts).
 * Writes d_merged_off (u64[n_gaps+1]; gaps this rank does not own are empty) and the rows. */
int gf_pools_merge_dev(gf_ctx* ctx, const void* d_src, size_t cap_rows, const void* d_cnt, int n_lib, int n_src_ranks,
                       size_t n_gaps, int read_len, int rank, int world, int batch, void* d_merged, size_t merged_cap_rows,
                       void* d_merged_off, void* d_error);

/* The same two steps for an EXACT-SIZE exchange (all-to-all-v of the recruited rows): the send / receive buffer is one
 * byte array of per-peer chunks of different sizes, described by device tables with one entry per slot s = peer * n_lib + lib:
 * d_slot_base[s] (u64) = byte offset of the slot's first row (even when the row size is even), d_slot_cap[s] (u32) = its rows,
 * d_cnt_base[s] (u64, multiple of 4) = byte offset of the slot's per-gap counts u32[n_gaps] INSIDE the same buffer — pack writes them
 * there (zero for gaps the peer does not own), merge reads them from there: the counts ride in the all-to-all, no all-gather of
 * counts.  Error bits as above (bit 30: a step produced more rows for a slot than the table gives it). */
int gf_pools_pack_for_owners_v_dev(gf_ctx* ctx, const void* d_pool_packed, const void* d_pool_off, size_t n_gaps, int read_len,
                                   int world, int batch, int lib, int n_lib, void* d_send, const void* d_slot_base,
                                   const void* d_slot_cap, const void* d_cnt_base_or_null, void* d_cnt, void* d_error);
int gf_pools_merge_v_dev(gf_ctx* ctx, const void* d_src, const void* d_slot_base, const void* d_cnt_base, int n_lib,
                         int n_src_ranks, size_t n_gaps, int read_len, int rank, int world, int batch, void* d_merged,
                         size_t merged_cap_rows, void* d_merged_off, void* d_error);

/* ---- §8f-1 on the device: flank anchoring (ContigsSelection, pick_contigs.py:64-358, with exact anchors instead of
 * `bwa mem -T {score}`: the last / first anchor_len bases of the left / right flank given to gf_set_gaps; 8 <= anchor_len <= 32;
 * the reference's scores are 30, then 15: assemble_gaps.py:336, 365).  Per contig the reference's pair choice (:149-297) on the
 * exact-anchor hits: the forward pair (leftmost left anchor, rightmost right anchor) when both forward hits exist, else the
 * reverse-strand pair; a pair counts when the right anchor starts at or behind the left anchor's end (:313-321).
 * d_gap_best[gap] (u64, caller zeroes; atomicMax, so several calls — other anchor lengths, other contig lists — accumulate) =
 * anchor_len << 56 | (span + 1) << 32 | (0x7FFFFFFF - contig index) << 1 | reverse strand: the longest span wins, the earlier
 * contig on ties (:313-321), and a pick at a longer anchor outranks every pick at a shorter one (the pipeline tries 15 only on
 * what 30 left open); 0 = no contig of the gap is anchored = gap not closed; *d_n_closed (u32, caller zeroes) counts the gaps
 * that became non-zero.  The span field holds 24 bits: a span + 1 of 2^24 bases or more saturates at 0xFFFFFF (it still wins over
 * every shorter span and the gap counts as closed; contigs of the per-gap assembly are three orders of magnitude shorter). */
int gf_pick_anchored_dev(gf_ctx* ctx, const void* d_contigs, const void* d_n_contigs, size_t contig_cap, const void* d_seq,
                         int anchor_len, void* d_gap_best, void* d_n_closed);
/* both scores of the pipeline in ONE pass over the contigs (anchor_len_short < anchor_len; 0: anchor_len only): the same words as
 * gf_pick_anchored_dev(anchor_len) followed by gf_pick_anchored_dev(anchor_len_short) leave in d_gap_best */
int gf_pick_anchored2_dev(gf_ctx* ctx, const void* d_contigs, const void* d_n_contigs, size_t contig_cap, const void* d_seq,
                          int anchor_len, int anchor_len_short, void* d_gap_best, void* d_n_closed);

/* ---- §8f-3, first piece: the all-pairs k-mer prefilter of the reference's ContigsMerger (QuickCheckerContigsMatch,
 * ContigsCompactor.cpp:1982-2095, applied by CompactVer3 :836-853 / threadQuickCheck :1073-1098).  A contig SET (one per gap: its
 * contigs.fa) becomes the node list [c0, revcomp(c0), c1, revcomp(c1), ...]; pair (i, j), i <= j, is feasible iff some k-mer of
 * the first or last 30 bases of node j occurs anywhere in node i (KmerUtils' 2-bit k-mers: any symbol but C/G/T counts as A).
 * Only feasible pairs go on to the overlap alignment (gf_overlap_evaluate below).  Contigs need >= 30 bases
 * (GF_E_INVAL; the reference reads out of bounds).  4 <= k <= 16 (the reference's default is 10).
 * seq: the contigs' ASCII bases back to back; contig_off[n_contigs+1]; set_off[n_sets+1] = contig index ranges of the sets.
 * Host variant: triples sorted by (set, i, j). */
typedef struct {
    uint32_t set; /* index of the contig set */
    uint32_t i, j; /* nodes: 2 * contig-in-set + strand (1 = reverse complement), i <= j */
} gf_qcpair;
int gf_quick_check(gf_ctx* ctx, const char* seq, const uint64_t* contig_off, const uint64_t* set_off, size_t n_sets, int k,
                   gf_qcpair* out, size_t cap, size_t* n_out);
/* device variant: max_set_contigs = contigs of the largest set (sizes the pair matrices); d_n_out = u32[2]: [0] triples found (more
 * than cap: truncated), [1] flags of SKIPPED sets (bit 0: a contig shorter than 30 bases, bit 1: more contigs than
 * max_set_contigs) — the host variant turns a non-zero flag into GF_E_INVAL; order unspecified */
int gf_quick_check_dev(gf_ctx* ctx, const void* d_seq, const void* d_contig_off, const void* d_set_off, size_t n_sets,
                       size_t max_set_contigs, int k, void* d_out, size_t cap, void* d_n_out);

/* ---- §8f-3, second stage: the contig merger's pairwise overlap evaluation (ContigsCompactor::Evaluate + IsScoreSignificant +
 * ContigsCompactorAction, ContigsCompactor-v0.2.0/ContigsMerger/ContigsCompactor.cpp:1572-1976, :108-159): an overlap alignment of
 * node i against node j (match +1, `mismatch` truncated to an int as the reference does, `indel`; ends may be clipped by up to
 * `max_clip` bases) classified like IsScoreSignificant.  Parameters = ContigsMerger's options (main.cpp:60-200): -i1 mismatch,
 * -i2 indel, -y max_clip, -c frac_min_overlap, -s frac_loss, -x min_overlap, -z min_overlap_scaffold; GAPPadder runs it with
 * -s 0.4 -i1 -2.0 -i2 -2.0 -x 12 -y 50 (MergeContigs.py:75).  The indel score must be integral (GF_E_UNSUPPORTED otherwise); contigs
 * up to 8190 bases.  Pairs are gf_qcpair {set, i, j} (ANY ordered node pair: i is the reference's pSeq1, j its pSeq2). */
typedef struct {
    double mismatch, indel, max_clip, frac_min_overlap, frac_loss, min_overlap, min_overlap_scaffold;
    double relax;             /* != 0: Evaluate's fRelax mode (FormMergedSeqFromPath, ContigsCompactor.cpp:1489): no significance test, res = 2 */
} gf_ovl_params;
typedef struct {
    int32_t res;              /* 0 no usable overlap, 1 overlap in [min_overlap_scaffold, min_overlap), 2 overlap >= min_overlap; -1: contig too long */
    int32_t row_end, col_end; /* end cell of the alignment (posRowEnd, posColEnd) */
    int32_t nclip;            /* bases clipped at one end */
    int32_t score;
    int32_t contained;        /* res != 0 only from here on: the action's contained flag */
    int32_t merged_len;       /* length of SetMergedStringConcat's string */
    int32_t overlap;          /* GetOverlapSize */
    int32_t containment;      /* IsContainment: such pairs form no edge (ContigsCompactor.cpp:672-674) */
    int32_t first_goes_first; /* 1: MODE_1_2 (node i then node j), 0: MODE_2_1 (:656-670) */
} gf_ovl_result;
int gf_overlap_evaluate(gf_ctx* ctx, const char* seq, const uint64_t* contig_off, const uint64_t* set_off, size_t n_sets, const gf_qcpair* pairs,
                        size_t n_pairs, const gf_ovl_params* params, gf_ovl_result* out);
int gf_overlap_evaluate_dev(gf_ctx* ctx, const void* d_seq, const void* d_contig_off, const void* d_set_off, const void* d_pairs, size_t n_pairs,
                            const gf_ovl_params* params, void* d_out);

/* ---- §8f-3 inside the step: the contig-merge ROUND for the gaps the pick left open (assemble_gaps.py:301-306 run_contigs_merge, which the
 * reference runs before it picks, :335-339; a gap its own contigs close gains nothing from merging).  For every gap g with
 * d_gap_best[g] == 0 and 2 .. 1 024 contigs in the list: exact-containment dedup (MergeContigs.drop_contained: a contig that occurs, on
 * either strand, inside another one goes; of identical ones the first stays), then — 2 .. max_set (<= 128) contigs left, those of
 * 30 .. 8 190 bases taking part — ContigsMerger itself: gf_quick_check's prefilter, gf_overlap_evaluate's edges (class 2, no containment),
 * strongly connected components in topological order, candidate roots and ends, a shortest-path DP per root with -overlap as edge length
 * (the 21 longest paths per root), removal of reverse-complement twin paths, and FormMergedSeqFromPath per path with Evaluate in its
 * relaxed mode (GraphUtils.cpp:625-859, 1028-1178, 1258-1344, 1422-1454; ContigsCompactor.cpp:773-983, 1456-1520).  The merged strings are
 * APPENDED to the contig list as records {gap, k = 0, kv = 0, n_nodes = nodes of the path, length, seq_off} in (gap, sorted path) order
 * (NEW_CONTIG_MERGE_1, _2, ... of each gap); *d_n_contigs and *d_seq_len grow.  Everything is enqueued on the context's stream, no host
 * synchronisation.  A gap's contigs are taken in the order of its contigs.fa: the (k_list[i], kv_list[i]) pairs in list order, inside a pair
 * by (length descending, sequence) — n_k <= 16; n_k == 0: record order.
 * d_stats: u32[32] — [0] open gaps with 2 .. 1 024 contigs, [1] gaps that went through the merger, [2] gaps left alone for their size
 * (more than 1 024 contigs, or more than max_set after the dedup), [3] candidate pairs, [4] prefilter flags, [5] merged contigs, [6] capacity
 * flags of THIS call (1 node buffer, 2 pair list, 32 contig list, 64 sequence buffer: the caller raises), [7] contigs before the round,
 * [8] edges, [9] gaps that got merged contigs, [16] gaps left alone because their graph outgrew the round's limits (4 096 edges, 2 048
 * paths, the job list).  Follow with gf_pick_anchored2_from_dev(d_first = d_stats + 7) to pick among the merged contigs only. */
int gf_merge_open_gaps_dev(gf_ctx* ctx, void* d_contigs, void* d_n_contigs, size_t contig_cap, void* d_seq, void* d_seq_len,
                           size_t seq_cap, const void* d_gap_best, size_t n_gaps, const gf_ovl_params* params, int kmer_len_quick,
                           int max_set, const int* k_list, const int* kv_list, int n_k, void* d_stats);
/* gf_pick_anchored2_dev over the contigs from *d_first (u32, device) on: the pick words of earlier calls stay and compete (atomicMax) */
int gf_pick_anchored2_from_dev(gf_ctx* ctx, const void* d_contigs, const void* d_n_contigs, size_t contig_cap, const void* d_seq,
                               int anchor_len, int anchor_len_short, const void* d_first, void* d_gap_best, void* d_n_closed);

/* ---- the reference's rescue round (assemble_gaps.py:166-217 run_collect_high_quality_unmap_to_contig_reads: `bwa mem` of a gap's high-quality
 * reads against its merged contigs, reads that align CLIPPED to at least two contigs are bridges), for ALL gaps of a round in one host call (no GPU
 * work; ctx may be NULL).  Contigs of gap g = texts [ctg_set_off[g], ctg_set_off[g+1]) of ctg_text (text c at ctg_off[c] .. ctg_off[c+1]), reads
 * likewise.  bwa is replaced by seed and extend as gappadder_amd/assemble_gaps.py::bridging_reads defines it: exact seed of seed_len characters on
 * either strand (at most 8 occurrences of a window per contig strand), first seed per (read, contig, strand, diagonal), clipped = a read end beyond
 * the contig or more than `budget` mismatches on either side of the seed, clipped AT a contig = every placement there clipped.
 * out_bridge[r] = 1 for the reads that are clipped at two contigs at least. */
int gf_bridging_reads(gf_ctx* ctx_or_null, const char* ctg_text, const uint64_t* ctg_off, const uint64_t* ctg_set_off, const char* read_text,
                      const uint64_t* read_off, const uint64_t* read_set_off, size_t n_gaps, int seed_len, int budget, uint8_t* out_bridge);

/* ---- device memory + timing helpers (so a ctypes host needs no other HIP binding) ---------------------- */
int gf_dev_alloc(gf_ctx* ctx, size_t bytes, void** d_ptr);
int gf_dev_free(gf_ctx* ctx, void* d_ptr);
int gf_memcpy_h2d(gf_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int gf_memcpy_d2h(gf_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
int gf_memset_dev(gf_ctx* ctx, void* d_ptr, int value, size_t bytes);
/* per-kernel HIP-event timing on the ctx's stream: enable, run, gf_sync, then read totals. */
#define GF_KERNEL_SCREEN 0
#define GF_KERNEL_TAG 1
#define GF_KERNEL_LOWMAPQ 2
#define GF_KERNEL_ASSEMBLE 3
#define GF_KERNEL_POOL 4
#define GF_KERNEL_SYNTH 5
#define GF_KERNEL_COUNT 6
#define GF_KERNEL_VERIFY 7  /* second kernel of the screen: exact per-candidate verification */
#define GF_KERNEL_INGEST 8  /* FASTQ text -> packed reads */
#define GF_KERNEL_PICK 9    /* flank anchoring */
#define GF_KERNEL_MERGE 10  /* contig-merge prefilter */
int gf_timing_enable(gf_ctx* ctx, int on);
int gf_timing_read(gf_ctx* ctx, int which, double* total_ms, uint64_t* launches);
int gf_timing_reset(gf_ctx* ctx);

/* ---- synthetic workload (bench/test utility, not part of the reference's path; definition: include/gf_synth.h,
 * shared bit-for-bit with oracle/gp_oracle.c).  gf_synth_pairs_dev fills 2*n_pairs packed reads (read 2p+m =
 * mate m+1 of pair first_pair+p) and, optionally, 2*n_pairs gf_alnrec (record 2p+e = forward/reverse end; record.read = 2p+m,
 * the read's index in THIS batch).
 * gf_synth_layout writes the n_scaffolds*gaps_per_scaffold gaps and their flanks (2*(flank_len-5) bases per gap). */
int gf_synth_pairs_dev(gf_ctx* ctx, const void* cfg /* gf_synth_cfg */, uint64_t first_pair, size_t n_pairs,
                       void* d_packed_reads, void* d_alnrecs_or_null);
int gf_synth_layout(const void* cfg /* gf_synth_cfg */, gf_gap* gaps, char* flank_ascii, uint64_t* flank_off);
/* the true bases of [start, start + n) of a scaffold, the planted gaps' interiors included: ground truth for checking filled gaps
 * (the reference's own evaluation compares picked sequences with the true ones: validate_gap_seqs.py:5-75) */
int gf_synth_truth(const void* cfg /* gf_synth_cfg */, uint32_t scaffold, uint64_t start, size_t n, char* out_ascii);

#ifdef __cplusplus
}
#endif
#endif /* GAPFILL_HIP_H */
