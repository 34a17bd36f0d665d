// gf_internal.hpp — context and launch plumbing shared by the translation units of libgapfill_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/gapfill_hip.h"
#include "kmer_dev.hpp"

namespace gf {

constexpr uint32_t EMPTY32 = 0xFFFFFFFFu;
constexpr int TILE_READS = 256;  // reads staged per workgroup pass of the screen filter
constexpr int N_KERNEL_SLOTS = 11;

// flank k-mer index for one k: three levels, all read-only on the device
struct FlankIndex {
    int k = 0;
    int stride = 0;          // distance between probed 16-mers of a read (= k-15)
    // level 1: bitmap over hashed canonical 16-mers (L2-resident filter)
    int bm_log2 = 0;
    uint32_t* d_bitmap = nullptr;
    // coarse copy for the LDS pre-filter (2^lds_log2 bits, lds_log2 <= 20), its fill ratio
    uint32_t* d_bitmap_lds = nullptr;
    int lds_log2 = 0;
    double lds_fill = 1.0;
    // L2-resident 2^24-bit OR-reduction of a level-1 bitmap that is larger than that (0 = none)
    uint32_t* d_bitmap_mid = nullptr;
    int mid_log2 = 0;
    double mid_fill = 1.0;
    // level 2: exact set of canonical 16-mers (open addressing, EMPTY32)
    int s_log2 = 0;
    uint32_t* d_sset = nullptr;
    // level 3: canonical k-mer -> gap, one slot per (k-mer, gap) pair
    int t_log2 = 0;
    // slot = one uint4 {hi.lo32, hi.hi32, gap, 0} for k <= 32, two uint4 {hi, lo}, {gap, 0, 0, 0} for k > 32;
    // gap == EMPTY32 marks a free slot
    void* d_table = nullptr;
    // seed-and-extend verification (min_hits == 1, no repeat mask): per exact-set slot the first occurrence of its 16-mer,
    // the occurrences {flank id, pos | strand<<16 | last<<17 | left room<<18 | right room<<24}, the flanks 2-bit packed
    // (each padded by 4 words on both sides) and the word offset of every flank's first base
    uint32_t* d_sval = nullptr;
    uint32_t* d_occ = nullptr;
    uint32_t* d_fpk = nullptr;
    uint32_t* d_foff = nullptr;
    bool ext_ok = false;     // every flank shorter than 65536 bases (16-bit positions in the occurrence words)
    // the exact set once more for pass B of the partitioned filter, in aligned groups of four slots that carry what stands NEXT to
    // the 16-mer in the flanks: 8 words {key x 4, ext x 4}, a key's home = group hash_s16_set(key) >> 2 (then the following groups);
    // ext = mask over the 2-base codes (nearest base first) left of the canonical 16-mer | the same for the right side << 16,
    // OR-ed over all occurrences (an occurrence with one neighbour sets the whole nibble of that base; no neighbour: nothing)
    uint32_t* d_sgrp = nullptr;
    size_t n_kmers = 0, n_s16 = 0, n_occ = 0;   // (n_occ: entries of d_occ)
    uint32_t max_gaps_per_kmer = 0;
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct TimedLaunch {
    hipEvent_t a, b;
    int which;
};

}  // namespace gf

struct gf_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string last_error;
    std::string screen_kernels;   // the filter kernels the last gf_screen_reads[_dev] launched (gf_screen_kernels)
    gf_ctx* after_filter = nullptr;   // one-shot: this context's stream waits for the end of the next filter pass of the owner (gf_stream_wait_after_filter)
    int n_cu = 256;

    // gaps
    std::vector<gf_gap> gaps;
    uint32_t n_scaffolds = 0;
    std::vector<std::string> flank_left, flank_right;
    gf_gap* d_gaps = nullptr;           // sorted as given (grouped by scaffold, ascending start)
    uint32_t* d_scaf_off = nullptr;     // n_scaffolds+1 offsets into d_gaps
    std::map<int, gf::FlankIndex> index;  // by k
    uint32_t max_gaps_per_kmer = 0;     // 0 = unlimited
    int bitmap_log2_override = 0;
    int index_host = 0;         // 1: build the flank index on the host (comparator of the device builder)
    int screen_variant = 0;     // filter kernel: 0 automatic; 9 plain, 13 pipelined, 16 / 17 256-bucket partitioned with whole-line / unaligned stores (tests run each)
    int screen_verify_batch = 64;  // verify kernel: candidates per wave and pass
    int screen_verify_ext = 1;   // min_hits == 1 without repeat mask: seed-and-extend verification instead of the k-mer table
    int screen_verify_gate = 1;  // verify kernel: consult the k-mer table only around exact 16-mer hits
    int screen_stream_policy = 1;  // pipelined filter: read stream loaded non-temporal (nt): keeps the L2 for the bitmap, -8 % fabric fetches
    int screen_ext = 1;          // 256-bucket filter: check the bases next to a seed against the flanks' (0: 16-base seeds as they are)
    int screen_pf4_cap8 = 0;     // tests: capacity of the 4-byte filter's pair list (0: sized from the reads)
    int screen_lds_log2_max = 20;   // coarse LDS bitmap of the screen: at most 2^20 bits (128 KiB)
    int tag_dbg = 0;             // diagnostics of the key-column tagger (option tag_dbg, refused unless GF_DIAGNOSTICS is set)
    int tag_light = 0;           // alignment tagger: one-wave workgroups, bin map through L1/L2 (runs beside the k-mer filter)
    int asm_lds_pool_kb = 152;
    int asm_threads = 0;         // threads per gap in the assembly kernel: 1024 / 512 / 256, 0 = by the pool bound (assemble.hip)
    int asm_simplify = 8;        // rounds of tip clipping + bubble popping in the assembly: until a round removes nothing, at most this many
                                 // (Velvet's defaults are on; 0: raw unitigs; measured: every C4 / C5 gap converges within two rounds)
    long asm_max_pool_reads = 0; // > 0: no pool has more rows than this (the assembly workspace is then one slice per workgroup, not per row)
    long asm_big_pool_reads = 131072;   // ... and pools beyond asm_max_pool_reads go to a second launch whose slices hold this many rows (a pool beyond this sets its gap_error)
    int asm_tiebreak = 1;        // error removal, equal coverage: 1 = fewer weak nodes (k-mers seen <= min_count + 1 times) win; 0 = sequence order alone
    int asm_keyslot = 1;         // count phase: key-in-slot LDS table when k <= 31 and min_count <= 2 (0: instance ids)
    int asm_ranked = 1;          // count phase: ranked (perfect-hash) table behind the pre-count (k > 32, LDS)
    int asm_last_threads = 0;    // the last launch_assemble: threads per gap of its main launch ...
    bool asm_last_split = false; // ... and whether a middle launch (1 024 threads per gap) followed it (gf_assemble_last_launch)
    int asm_sweep = 0;           // 1: gf_assemble_multi_dev runs the sweep 31/29, 41/39, 51/49 in one launch (measured 1 % slower than one launch per pair: DESIGN.md §5)
    int asm_pre_frac8 = 5;       // count phase: eighths of the LDS region the pre-count's bit arrays may take under an LDS table
    int asm_precount = 1;        // count phase: bit-array pre-count in LDS when min_count is 2 or 3 (0: every window goes to the table)
    void* asm_stats = nullptr;  // device u64[4], added to by every assembled gap: windows, k-mers counted exactly, surviving k-mers, nodes (option asm_stats_ptr)
    void* asm_dbg = nullptr;  // diagnostic: device buffer for per-gap phase stamps (option asm_dbg_ptr)
    // tagger bin maps, one per dist2 (a pipeline tags every library with its own window; the four last used stay): the device copy is
    // [bits | n_scaffolds + 1 bit offsets | per bin: the first gap of the scaffold whose right window reaches the bin]
    struct TagMap {
        int dist2 = -1, shift = 0;
        uint32_t words = 0, fine_words = 0, fine_shift = 0;
        uint64_t used = 0;
        gf::DevBuf map, fine;
    };
    std::vector<TagMap> tag_maps;
    uint64_t tag_map_clock = 0;
    // tagger: bits of the LDS bin map and of the global one behind it; non-temporal record loads.  Measured on the C4 layout (200 M
    // records, rocprof): 2^25-bit global map 2.24 ms; + nt loads (the record stream no longer evicts the map from the L2s) 1.90 ms;
    // 2^23 bits + nt 1.82 ms; 32 / 64 KiB LDS maps 2.85 / 3.13 ms (fewer workgroups per CU)
    int tag_bins_log2 = 17, tag_fine_log2 = 23, tag_nt = 1;
    // second-hop table cache
    std::vector<uint32_t> low_rows, rowgap_rows;
    std::map<int, gf::DevBuf> anchor_tabs;   // by anchor length: the flank anchors of gf_pick_anchored_dev (dropped by gf_set_gaps)
    size_t low_b1 = 0, low_b2 = 0;

    // scratch
    gf::DevBuf cand, cand2, part_ws, tag_stage, verify_stage, bam_stream, bam_recs, asm_table, asm_surv, asm_nodes, asm_jump, asm_big, rowgap, pool_ws, xchg_ws, xchg_ws2, merge_ws, counters, stage_in, stage_out, stage_aux, table;
    size_t bam_n_recs = 0;       // alignment records gf_bam_pack left in bam_recs (for gf_tag_*_bam)
    size_t bam_stream_len = 0;   // inflated BAM bytes gf_bgzf_inflate left in bam_stream
    // timing
    bool timing = false;
    std::vector<gf::TimedLaunch> launches;
    double t_total[gf::N_KERNEL_SLOTS] = {0};
    uint64_t t_count[gf::N_KERNEL_SLOTS] = {0};
};

namespace gf {

int set_hip_error(gf_ctx* ctx, hipError_t e, const char* what);
int ensure(gf_ctx* ctx, DevBuf& b, size_t bytes);
void drop_tag_maps(gf_ctx* ctx);   // (tagger.hip) frees the tagger's bin maps: the gaps or the map sizes changed

#define GF_HIP(ctx, call)                                          \
    do {                                                           \
        hipError_t e__ = (call);                                   \
        if (e__ != hipSuccess) return gf::set_hip_error(ctx, e__, #call); \
    } while (0)

struct LaunchTimer {
    gf_ctx* ctx;
    int which;
    hipEvent_t a = nullptr, b = nullptr;
    LaunchTimer(gf_ctx* c, int w) : ctx(c), which(w) {
        if (ctx->timing) {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            (void)hipEventRecord(a, ctx->stream);
        }
    }
    ~LaunchTimer() {
        if (a) {
            (void)hipEventRecord(b, ctx->stream);
            ctx->launches.push_back({a, b, which});
        }
    }
};

// api.hip: zero up to four small device regions with ONE launch (every hipMemsetAsync is a launch of its own, and the step is
// a chain of ~30 dependent launches)
struct ZeroList {
    uint32_t* p[4];
    uint32_t n[4];   // 32-bit words
};
void zero_regions(gf_ctx* ctx, const ZeroList& z);

// index.cpp
int build_flank_index(gf_ctx* ctx, int k, FlankIndex** out);
void free_flank_index(gf_ctx* ctx, FlankIndex& ix);

// screen.hip
int build_flank_index_dev(gf_ctx* ctx, int k, FlankIndex& ix);   // index_dev.hip
int build_sgrp_dev(gf_ctx* ctx, FlankIndex& ix);                 // index_dev.hip: d_sgrp from d_sset / d_sval / d_occ / d_fpk
int launch_screen(gf_ctx* ctx, const FlankIndex& ix, const void* d_reads, const void* d_nmask, size_t n_reads,
                  int read_len, int min_hits, void* d_out, size_t cap, void* d_n_out);
// tagger.hip
int launch_tag(gf_ctx* ctx, const void* d_recs, size_t n, int insert_size, int sd, int clip_dist, int anchor_mapq,
               void* d_out, size_t cap, void* d_n_out, void* d_low, size_t low_cap, void* d_n_low, const void* d_keys = nullptr);
int launch_alnrec_keys(gf_ctx* ctx, const void* d_recs, size_t n, void* d_keys);
int launch_low_mapq(gf_ctx* ctx, const void* d_recs, size_t n, const gf_dpos* table, size_t n_rows, void* d_out,
                    size_t cap, void* d_n_out, const void* d_low, const void* d_n_low, size_t low_cap);

// assemble.hip
// second hop: hashed bit map of the table rows' neighbourhoods (hop.hip builds it, tagger.hip asks it)
constexpr uint32_t HOP_NEAR_LOG2 = 22, HOP_NEAR_SHIFT = 9;
__host__ __device__ inline uint32_t hop_near_bit(uint32_t scaffold, uint32_t bin) {
    return ((scaffold * 0x9E3779B1u) ^ (bin * 0x85EBCA77u) ^ (bin >> 13)) >> (32 - HOP_NEAR_LOG2);
}
constexpr size_t GF_COUNTER_BYTES = 128;   // ctx->counters: [0, 8) screen, [8, 16) assembly / merge, [16, 32) the assembly sweep
int launch_assemble_sweep(gf_ctx* ctx, const void* d_pool, const void* d_pool_off, size_t n_pools, size_t total_reads, int read_len,
                          int min_count, int min_contig, void* d_contigs, size_t contig_cap, void* d_n_contigs, void* d_seq,
                          size_t seq_cap, void* d_seq_len, void* d_gap_error);
int launch_assemble(gf_ctx* ctx, const void* d_pool, const void* d_nmask, const void* d_pool_off, size_t n_pools,
                    size_t total_reads, int read_len, int k, int kv, int min_count, int min_contig, void* d_contigs,
                    size_t contig_cap, void* d_n_contigs, void* d_seq, size_t seq_cap, void* d_seq_len, void* d_gap_error,
                    void* d_cnt_keys, void* d_cnt_counts, size_t cnt_cap, bool append);

}  // namespace gf
