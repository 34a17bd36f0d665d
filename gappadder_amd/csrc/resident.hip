// resident.hip — BAM + FASTQ FILES into a library that stays in HBM (the device pipeline of the CLI, gappadder_amd/device_collect.py).
//
// The reference joins alignments and reads BY NAME: list lines carry QNAME (collect_reads_for_gaps.py:119-159), the FASTQ line machine
// looks every record's id — first whitespace token of the header, cut at '/', '@' dropped — up in {readId -> set(gapKey)}
// (run_multi_threads_discordant.py:153-185, 209-241).  The device pipeline addresses reads by INDEX (read = 2 * pair + mate,
// "read idx is the ID"), so ingest has to establish that index once:
//   gf_fastq_index_dev   64-bit hash of every FASTQ record's id (the reference's normalisation) + the longest sequence line
//   gf_bam_append_dev    alignment records of one inflated BAM chunk appended to the caller's resident array, with a 64-bit hash of
//                        every QNAME and (optionally) the QNAME bytes in an arena — only the names of the few records that produce
//                        a hit ever cross PCIe, for the list files the reference's file contract asks for
//   gf_read_join_dev     hash table over the FASTQ ids -> rec.read = 2 * record number + mate for every alignment record
// The join is by 64-bit hash; the host verifies the NAMES of the records that produce hits (exact compare) and fails loudly on a
// mismatch, so a collision can never silently recruit a wrong read.
#include <vector>

#include "gf_internal.hpp"

namespace gf {

__device__ __forceinline__ uint64_t name_hash_step(uint64_t h, uint8_t b) { return (h ^ b) * 0x100000001B3ull; }   // FNV-1a 64
__device__ __forceinline__ uint64_t name_hash_final(uint64_t h) {   // splitmix64 finaliser; 0 and ~0 are kept free (table sentinels)
    h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 27; h *= 0x94D049BB133111EBull;
    h ^= h >> 31;
    if (h == 0) h = 1;
    if (h == ~0ull) h = ~0ull - 1;
    return h;
}
constexpr uint64_t NAME_HASH_SEED = 0xCBF29CE484222325ull;

// ---- FASTQ ids ------------------------------------------------------------------------------------------------------------
// one thread per record: header line from hdr_begin[r]: '@' dropped, hashed up to the first whitespace or '/'
// (run_multi_threads_discordant.py:212-214: `fields[0].split("/")[0][1:]`); then the sequence line's length
__global__ __launch_bounds__(256) void fastq_index_kernel(const uint8_t* text, uint64_t n, const unsigned long long* hdr_begin, uint64_t n_reads,
                                                          unsigned long long* id_hash, uint32_t* max_len) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t len = 0;
    if (r < n_reads) {
        uint64_t i = hdr_begin[r];
        // the reference splits the header line on whitespace and takes field 0: leading blanks are skipped by split()
        while (i < n && (text[i] == ' ' || text[i] == '\t')) ++i;
        uint64_t h = NAME_HASH_SEED;
        bool first = true, in_id = true;
        for (; i < n; ++i) {
            const uint8_t c = text[i];
            if (c == '\n') break;
            if (in_id) {
                if (c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || c == '/') in_id = false;
                else if (first) first = false;      // the '@' (whatever the first character is: `[1:]`)
                else h = name_hash_step(h, c);
            }
        }
        id_hash[r] = name_hash_final(h);
        uint64_t s = i + 1, e = s;
        while (e < n && text[e] != '\n') ++e;
        if (e > s && text[e - 1] == '\r') --e;
        len = (uint32_t)(e - s);
    }
    for (int d = 32; d; d >>= 1) len = max(len, (uint32_t)__shfl_xor(len, d));
    if ((threadIdx.x & 63) == 0 && len) atomicMax(max_len, len);
}

// ---- BAM records -> resident arrays ---------------------------------------------------------------------------------------
struct BamStream2 {
    const uint8_t* p;
    uint64_t n, first;
    int32_t n_ref;
};
constexpr uint64_t BAM2_SEG = 65536;
constexpr uint64_t BAM2_TAIL = 1ull << 63;

__device__ __forceinline__ uint32_t r_ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t r_ld16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// per segment (bam.hip's chain: guess[s] = first record start of segment s): the QNAME bytes of the records that start in it
__global__ __launch_bounds__(256) void bam_name_bytes_kernel(BamStream2 B, uint64_t n_seg, const unsigned long long* guess, uint32_t* name_bytes) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    uint64_t e = B.first + (s + 1) * BAM2_SEG;
    if (e > B.n) e = B.n;
    uint64_t o = guess[s];
    uint32_t t = 0;
    if (!(o & BAM2_TAIL))
        while (o < e) {
            if (o + 4 > B.n) break;
            const int64_t bs = (int32_t)r_ld32(B.p + o);
            if (bs < 32 || o + 4 + (uint64_t)bs > B.n) break;
            const uint32_t l_nm = B.p[o + 12];
            if (l_nm == 0 || 36 + (uint64_t)l_nm > (uint64_t)bs + 4) break;      // malformed record: the chain ends here (as in bam_append_kernel)
            t += l_nm - 1u;      // l_read_name counts the NUL
            o += 4 + (uint64_t)bs;
        }
    name_bytes[s] = t;
}

struct BamAppend {
    gf_alnrec* recs;                 // + rec_base applied
    unsigned long long* qhash;       // or null
    uint8_t* names;                  // arena (+ name_base applied) or null
    unsigned long long* name_off;    // [rec] = offset into the arena (absolute), or null
    uint64_t name_base;
    uint32_t* ref_seen;              // per scaffold: |1 a record, |2 a MAPQ-0 record; or null
    unsigned long long* rec_begin;   // per record of THIS chunk: offset of its block_size field in the inflated stream; or null
    uint32_t n_scaffolds;
};

__global__ __launch_bounds__(256) void bam_append_kernel(BamStream2 B, uint64_t n_seg, const unsigned long long* guess, const unsigned long long* rec_off,
                                                         const unsigned long long* name_off_seg, const uint32_t* ref_map, BamAppend A, uint64_t n_total) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    uint64_t e = B.first + (s + 1) * BAM2_SEG;
    if (e > B.n) e = B.n;
    uint64_t o = guess[s];
    if (o & BAM2_TAIL) return;
    uint64_t idx = rec_off[s], noff = name_off_seg[s];
    while (o < e && idx < n_total) {
        if (o + 4 > B.n) break;
        const int64_t bs = (int32_t)r_ld32(B.p + o);
        if (bs < 32 || o + 4 + (uint64_t)bs > B.n) break;
        const uint8_t* r = B.p + o;
        const int32_t ref = (int32_t)r_ld32(r + 4), mref = (int32_t)r_ld32(r + 24);
        const uint32_t l_name = r[12], n_cig = r_ld16(r + 16);
        if (l_name == 0 || 36 + (uint64_t)l_name > (uint64_t)bs + 4) break;      // malformed (l_read_name counts the NUL; the name lies inside the block): chain break
        gf_alnrec a;
        a.pos = r_ld32(r + 8) + 1u;
        a.mate_pos = r_ld32(r + 28) + 1u;
        a.tlen = (int32_t)r_ld32(r + 32);
        a.ref = ref >= 0 && ref < B.n_ref ? ref_map[ref] : 0xFFFFFFFFu;
        a.mate_ref = mref >= 0 && mref < B.n_ref ? ref_map[mref] : 0xFFFFFFFFu;
        a.flag = (uint16_t)r_ld16(r + 18);
        a.mapq = r[13];
        uint32_t cf = 0;
        if (n_cig && 36 + (uint64_t)l_name + 4ull * n_cig <= (uint64_t)bs + 4) {
            const uint8_t* cg = r + 36 + l_name;
            const uint32_t op0 = r_ld32(cg) & 0xFu, op1 = r_ld32(cg + 4 * (n_cig - 1)) & 0xFu;   // 4 = S, 5 = H
            cf = ((op1 == 4 || op1 == 5) ? 2u : 0u) + ((op0 == 4 || op0 == 5) ? 1u : 0u);
        }
        a.clipflag = (uint8_t)cf;
        a.read = 0xFFFFFFFFull;          // no read yet: gf_read_join_dev
        A.recs[idx] = a;
        const uint32_t nl = l_name - 1u;
        if (A.qhash) {
            uint64_t h = NAME_HASH_SEED;
            for (uint32_t i = 0; i < nl; ++i) h = name_hash_step(h, r[36 + i]);
            A.qhash[idx] = name_hash_final(h);
        }
        if (A.name_off) A.name_off[idx] = A.name_base + noff;
        if (A.rec_begin) A.rec_begin[idx] = o;
        if (A.names)
            for (uint32_t i = 0; i < nl; ++i) A.names[noff + i] = r[36 + i];
        noff += nl;
        if (A.ref_seen && a.ref < A.n_scaffolds) {
            const uint32_t bits = 1u | (a.mapq == 0 ? 2u : 0u);
            if ((A.ref_seen[a.ref] & bits) != bits) atomicOr(&A.ref_seen[a.ref], bits);
        }
        ++idx;
        o += 4 + (uint64_t)bs;
    }
}

// ---- the join -------------------------------------------------------------------------------------------------------------
// table: open addressing over the 64-bit id hashes; slot = {hash, record number}; 0 = free
__global__ __launch_bounds__(256) void join_build_kernel(const unsigned long long* id_hash, uint64_t n, unsigned long long* keys, uint32_t* vals,
                                                         uint32_t t_log2, uint32_t* stats) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long h = id_hash[i];
    const uint64_t mask = (1ull << t_log2) - 1;
    uint64_t sl = (h >> 7) & mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&keys[sl], 0ull, h);
        if (prev == 0ull) { vals[sl] = (uint32_t)i; return; }
        if (prev == h) { atomicAdd(&stats[0], 1u); return; }       // the same id twice (or a collision): the host decides
        sl = (sl + 1) & mask;
    }
}

__global__ __launch_bounds__(256) void join_probe_kernel(gf_alnrec* recs, const unsigned long long* qhash, uint64_t n, const unsigned long long* keys,
                                                         const uint32_t* vals, uint32_t t_log2, uint32_t* stats) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long h = qhash[i];
    const uint64_t mask = (1ull << t_log2) - 1;
    uint64_t sl = (h >> 7) & mask;
    uint64_t read = 0xFFFFFFFFull;
    for (;;) {
        const unsigned long long k = keys[sl];
        if (k == 0ull) break;
        if (k == h) {
            // the record's OWN list is "left" when FLAG has 0x40 (first in pair), else "right" (collect_reads_for_gaps.py:93-102)
            read = 2ull * vals[sl] + ((recs[i].flag & 0x40u) ? 0u : 1u);
            break;
        }
        sl = (sl + 1) & mask;
    }
    recs[i].read = read;
    if (read == 0xFFFFFFFFull) atomicAdd(&stats[1], 1u);
}

__global__ __launch_bounds__(256) void fetch_slices_kernel(const uint8_t* src, const unsigned long long* begin, const unsigned long long* end,
                                                           const unsigned long long* dst_off, uint64_t n, uint8_t* dst) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint8_t* s = src + begin[i];
    uint8_t* d = dst + dst_off[i];
    const uint64_t len = end[i] - begin[i];
    for (uint64_t k = lane; k < len; k += 64) d[k] = s[k];
}

// rows gathered by index (N masks of pooled reads: gf_build_pools_dev gathers the packed bases and reports the read ids)
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint32_t* src, uint32_t row_words, const uint32_t* ids, const unsigned long long* d_n,
                                                          uint64_t cap, uint64_t n_src_rows, uint32_t* dst) {
    uint64_t n = *d_n;
    if (n > cap) n = cap;
    const uint64_t total = n * row_words;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / row_words;
        const uint32_t w = (uint32_t)(i - r * row_words);
        const uint32_t id = ids[r];
        dst[i] = id < n_src_rows ? src[(uint64_t)id * row_words + w] : 0xFFFFFFFFu;
    }
}

}  // namespace gf

using namespace gf;

// bam.hip: the record chain of the stream gf_bgzf_inflate left on the device (guess / walk / check rounds); leaves in pool_ws the
// per-segment first-record offsets and record counts
namespace gf {
int bam_chain(gf_ctx* ctx, size_t n_bytes, size_t first, size_t n_ref, unsigned long long** d_guess, unsigned long long** d_rec_off, uint64_t* n_seg,
              size_t* n_recs, size_t* n_consumed, size_t extra_ws, uint8_t** d_extra);
void bam_scan_launch(gf_ctx* ctx, const uint32_t* cnt, uint32_t n, unsigned long long* off);   // u32 counts -> u64 exclusive offsets, total in off[n]
}

extern "C" {

int gf_fastq_index_dev(gf_ctx* ctx, const void* d_text, size_t n_bytes, const void* d_hdr_begin, size_t n_reads, void* d_id_hash, void* d_max_len) {
    if (!ctx || !d_id_hash || !d_max_len || (n_reads && (!d_text || !d_hdr_begin))) return GF_E_INVAL;
    if (!n_reads) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    LaunchTimer tm(ctx, GF_KERNEL_INGEST);
    hipLaunchKernelGGL(fastq_index_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, ctx->stream, (const uint8_t*)d_text, (uint64_t)n_bytes,
                       (const unsigned long long*)d_hdr_begin, (uint64_t)n_reads, (unsigned long long*)d_id_hash, (uint32_t*)d_max_len);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_bam_append_dev(gf_ctx* ctx, size_t n_bytes, size_t first, const uint32_t* ref_map, size_t n_ref, void* d_recs, size_t rec_base, size_t rec_cap,
                      void* d_qhash_or_null, void* d_names_or_null, size_t name_base, size_t name_cap, void* d_name_off_or_null,
                      void* d_ref_seen_or_null, size_t n_scaffolds, void* d_rec_begin_or_null, size_t* n_recs, size_t* n_name_bytes, size_t* n_consumed) {
    if (!ctx || !n_recs || !n_name_bytes || !n_consumed || !d_recs || (n_ref && !ref_map) || first > n_bytes || n_ref > 0x7FFFFFFF || rec_base > rec_cap)
        return GF_E_INVAL;
    *n_recs = 0;
    *n_name_bytes = 0;
    *n_consumed = first;
    if (n_bytes != ctx->bam_stream_len || !ctx->bam_stream.p) return GF_E_STATE;   // no stream left on the device by gf_bgzf_inflate
    if (first == n_bytes) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long *d_guess = nullptr, *d_rec_off = nullptr;
    uint64_t n_seg = 0;
    uint8_t* d_extra = nullptr;
    const uint64_t seg_guess = (n_bytes - first + BAM2_SEG - 1) / BAM2_SEG;
    const size_t b_nb = (seg_guess * 4 + 63) & ~(size_t)63, b_no = ((seg_guess + 1) * 8 + 63) & ~(size_t)63, b_map = (n_ref * 4 + 63) & ~(size_t)63;
    int rc = bam_chain(ctx, n_bytes, first, n_ref, &d_guess, &d_rec_off, &n_seg, n_recs, n_consumed, b_nb + b_no + b_map + 64, &d_extra);
    if (rc) return rc;
    const size_t total = *n_recs;
    if (!total) return GF_OK;
    uint32_t* d_nb = (uint32_t*)d_extra;
    unsigned long long* d_no = (unsigned long long*)(d_extra + b_nb);
    uint32_t* d_map = (uint32_t*)(d_extra + b_nb + b_no);
    const BamStream2 B{(const uint8_t*)ctx->bam_stream.p, (uint64_t)n_bytes, (uint64_t)first, (int32_t)n_ref};
    const unsigned grid = (unsigned)((n_seg + 255) / 256);
    unsigned long long name_total = 0;
    {
        LaunchTimer tm(ctx, GF_KERNEL_INGEST);
        hipLaunchKernelGGL(bam_name_bytes_kernel, dim3(grid), dim3(256), 0, ctx->stream, B, n_seg, d_guess, d_nb);
        bam_scan_launch(ctx, d_nb, (uint32_t)n_seg, d_no);
        GF_HIP(ctx, hipMemcpyAsync(&name_total, d_no + n_seg, 8, hipMemcpyDeviceToHost, ctx->stream));
        GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *n_name_bytes = (size_t)name_total;
    if (rec_base + total > rec_cap || (d_names_or_null && name_base + name_total > name_cap)) return GF_E_NOSPACE;
    if (n_ref) GF_HIP(ctx, hipMemcpyAsync(d_map, ref_map, n_ref * 4, hipMemcpyHostToDevice, ctx->stream));
    BamAppend A;
    A.recs = (gf_alnrec*)d_recs + rec_base;
    A.qhash = d_qhash_or_null ? (unsigned long long*)d_qhash_or_null + rec_base : nullptr;
    A.names = d_names_or_null ? (uint8_t*)d_names_or_null + name_base : nullptr;
    A.name_off = d_name_off_or_null ? (unsigned long long*)d_name_off_or_null + rec_base : nullptr;
    A.name_base = name_base;
    A.ref_seen = (uint32_t*)d_ref_seen_or_null;
    A.rec_begin = (unsigned long long*)d_rec_begin_or_null;
    A.n_scaffolds = (uint32_t)n_scaffolds;
    {
        LaunchTimer tm(ctx, GF_KERNEL_INGEST);
        hipLaunchKernelGGL(bam_append_kernel, dim3(grid), dim3(256), 0, ctx->stream, B, n_seg, d_guess, d_rec_off, d_no, d_map, A, (uint64_t)total);
    }
    if (d_name_off_or_null) {   // the end of the last name = the start of the next chunk's first
        const unsigned long long end = name_base + name_total;
        GF_HIP(ctx, hipMemcpyAsync((unsigned long long*)d_name_off_or_null + rec_base + total, &end, 8, hipMemcpyHostToDevice, ctx->stream));
    }
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_read_join_dev(gf_ctx* ctx, const void* d_id_hash, size_t n_ids, void* d_recs, const void* d_qhash, size_t n_recs, void* d_stats) {
    if (!ctx || !d_stats || (n_ids && !d_id_hash) || (n_recs && (!d_recs || !d_qhash)) || n_ids >= 0x7FFFFFFFull) return GF_E_INVAL;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    GF_HIP(ctx, hipMemsetAsync(d_stats, 0, 16, ctx->stream));
    uint32_t t_log2 = 10;
    while ((1ull << t_log2) < 2 * (uint64_t)n_ids + 16) ++t_log2;
    const size_t slots = (size_t)1 << t_log2;
    int rc;
    if ((rc = ensure(ctx, ctx->table, slots * 12 + 64))) return rc;
    ctx->low_rows.clear();      // ctx->table no longer holds the host second-hop table launch_low_mapq may think it cached (as hop.hip does)
    unsigned long long* keys = (unsigned long long*)ctx->table.p;
    uint32_t* vals = (uint32_t*)((uint8_t*)ctx->table.p + slots * 8);
    GF_HIP(ctx, hipMemsetAsync(keys, 0, slots * 8, ctx->stream));
    LaunchTimer tm(ctx, GF_KERNEL_INGEST);
    if (n_ids)
        hipLaunchKernelGGL(join_build_kernel, dim3((unsigned)((n_ids + 255) / 256)), dim3(256), 0, ctx->stream, (const unsigned long long*)d_id_hash,
                           (uint64_t)n_ids, keys, vals, t_log2, (uint32_t*)d_stats);
    if (n_recs)
        hipLaunchKernelGGL(join_probe_kernel, dim3((unsigned)((n_recs + 255) / 256)), dim3(256), 0, ctx->stream, (gf_alnrec*)d_recs,
                           (const unsigned long long*)d_qhash, (uint64_t)n_recs, keys, vals, t_log2, (uint32_t*)d_stats);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_fetch_slices(gf_ctx* ctx, const void* d_src, size_t src_len, const uint64_t* begin, const uint64_t* end, size_t n, uint8_t* dst, size_t cap,
                    size_t* n_bytes) {
    if (!ctx || !n_bytes || (n && (!begin || !end || !d_src))) return GF_E_INVAL;
    *n_bytes = 0;
    if (n == 0) return GF_OK;
    std::vector<unsigned long long> h(3 * n);   // begin | end | dst offsets
    size_t total = 0;
    for (size_t i = 0; i < n; ++i) {
        if (begin[i] > end[i] || end[i] > src_len) return GF_E_INVAL;
        h[i] = begin[i];
        h[n + i] = end[i];
        h[2 * n + i] = total;
        total += (size_t)(end[i] - begin[i]);
    }
    *n_bytes = total;
    if (total > cap || (total && !dst)) return GF_E_NOSPACE;
    if (total == 0) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    const size_t b_idx = (3 * n * 8 + 63) & ~(size_t)63;
    if ((rc = ensure(ctx, ctx->stage_out, b_idx + total + 64))) return rc;
    unsigned long long* d_idx = (unsigned long long*)ctx->stage_out.p;
    uint8_t* d_dst = (uint8_t*)ctx->stage_out.p + b_idx;
    GF_HIP(ctx, hipMemcpyAsync(d_idx, h.data(), 3 * n * 8, hipMemcpyHostToDevice, ctx->stream));
    if ((n + 3) / 4 >= 0xFFFFFFFFull) return GF_E_UNSUPPORTED;
    hipLaunchKernelGGL(fetch_slices_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, (const uint8_t*)d_src, d_idx, d_idx + n, d_idx + 2 * n,
                       (uint64_t)n, d_dst);
    GF_HIP(ctx, hipMemcpyAsync(dst, d_dst, total, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_gather_rows_dev(gf_ctx* ctx, const void* d_src, size_t n_src_rows, size_t row_bytes, const void* d_ids, const void* d_n, size_t cap, void* d_dst) {
    if (!ctx || !d_src || !d_ids || !d_n || !d_dst || row_bytes == 0 || (row_bytes & 3)) return GF_E_INVAL;
    if (!cap) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    LaunchTimer tm(ctx, GF_KERNEL_POOL);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)std::min<size_t>(ctx->n_cu * 8, (cap * (row_bytes / 4) + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint32_t*)d_src, (uint32_t)(row_bytes / 4), (const uint32_t*)d_ids, (const unsigned long long*)d_n, (uint64_t)cap,
                       (uint64_t)n_src_rows, (uint32_t*)d_dst);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

}  // extern "C"
