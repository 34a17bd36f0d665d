// ingest.hip — FASTQ text -> 2-bit packed reads (+ N mask) on the device ("ingest").
//
// The reference reads FASTQ with a 4-state line machine in CPython (run_multi_threads_discordant.py:205-232: header, sequence,
// '+', quality; the id is the first whitespace token of the header cut at '/').  Here the raw text of a FASTQ file (or a chunk
// that starts at a record boundary) is copied to HBM as it is and three small kernels turn it into the packed layout of
// gf_pack_reads: count the newlines per tile, scan, record where every header and sequence line starts/ends, pack.  The host
// keeps the text for the ids; the device returns the byte offset of every record's header so that ids can be cut lazily.
#include <vector>

#include "gf_internal.hpp"

namespace gf {

constexpr uint32_t ING_THREADS = 256, ING_PER_THREAD = 64, ING_TILE = ING_THREADS * ING_PER_THREAD;

__device__ __forceinline__ uint32_t count_nl_word(uint32_t w) {   // bytes of w equal to '\n'
    const uint32_t x = w ^ 0x0A0A0A0Au;
    const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);   // 0x80 in every zero byte of x
    return __popc(z);
}

// newlines of the thread's 64-byte segment (bytes past n count as nothing)
__device__ __forceinline__ uint32_t count_nl_segment(const uint8_t* text, uint64_t n, uint64_t a) {
    uint32_t c = 0;
    if (a + ING_PER_THREAD <= n && ((uintptr_t)(text + a) & 15) == 0) {
        const uint4* p = reinterpret_cast<const uint4*>(text + a);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4 v = p[j];
            c += count_nl_word(v.x) + count_nl_word(v.y) + count_nl_word(v.z) + count_nl_word(v.w);
        }
    } else {
        for (uint64_t i = a; i < a + ING_PER_THREAD && i < n; ++i) c += text[i] == '\n';
    }
    return c;
}

__global__ __launch_bounds__(ING_THREADS) void nl_count_kernel(const uint8_t* text, uint64_t n, uint32_t* tile_cnt) {
    __shared__ uint32_t part[ING_THREADS / 64];
    const uint64_t a = (uint64_t)blockIdx.x * ING_TILE + (uint64_t)threadIdx.x * ING_PER_THREAD;
    uint32_t c = a < n ? count_nl_segment(text, n, a) : 0;
    for (int d = 32; d; d >>= 1) c += __shfl_down(c, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// single-block exclusive scan of tile counts (u32 -> u64 offsets, total in off[n])
__global__ __launch_bounds__(1024) void nl_scan_kernel(const uint32_t* cnt, uint32_t n, unsigned long long* off) {
    __shared__ unsigned long long part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = (n + 1023) / 1024;
    const uint32_t a = tid * chunk < n ? tid * chunk : n, b = a + chunk < n ? a + chunk : n;
    unsigned long long s = 0;
    for (uint32_t i = a; i < b; ++i) s += cnt[i];
    part[tid] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const unsigned long long v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned long long run = tid ? part[tid - 1] : 0;
    for (uint32_t i = a; i < b; ++i) {
        off[i] = run;
        run += cnt[i];
    }
    if (tid == 1023) off[n] = part[1023];
}

// Line index L (0-based) ends at newline number L.  Record r = lines 4r .. 4r+3:
//   newline 4r   ends the header  -> seq_begin[r] = pos + 1
//   newline 4r+1 ends the sequence -> seq_end[r] = pos
//   newline 4r+3 ends the quality  -> hdr_begin[r+1] = pos + 1     (hdr_begin[0] = 0)
__global__ __launch_bounds__(ING_THREADS) void nl_mark_kernel(const uint8_t* text, uint64_t n, const unsigned long long* tile_off,
                                                              uint64_t cap_reads, unsigned long long* hdr_begin,
                                                              unsigned long long* seq_begin, unsigned long long* seq_end) {
    __shared__ uint32_t wsum[ING_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint64_t a = (uint64_t)blockIdx.x * ING_TILE + (uint64_t)tid * ING_PER_THREAD;
    const uint32_t c = a < n ? count_nl_segment(text, n, a) : 0;
    uint32_t incl = c;   // inclusive scan inside the wave
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d);
        if ((int)lane >= d) incl += v;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t before = incl - c;
    for (uint32_t i = 0; i < w; ++i) before += wsum[i];
    if (blockIdx.x == 0 && tid == 0 && cap_reads) hdr_begin[0] = 0;
    if (!c) return;
    unsigned long long idx = tile_off[blockIdx.x] + before;   // number of the first newline of this segment
    auto mark = [&](uint64_t i) {
        const unsigned long long r = idx >> 2;
        const uint32_t m = (uint32_t)idx & 3u;
        if (m == 0) { if (r < cap_reads) seq_begin[r] = i + 1; }
        else if (m == 1) { if (r < cap_reads) seq_end[r] = i; }
        else if (m == 3) { if (r + 1 < cap_reads) hdr_begin[r + 1] = i + 1; }
        ++idx;
    };
    if (a + ING_PER_THREAD <= n && ((uintptr_t)(text + a) & 15) == 0) {
        const uint4* p = reinterpret_cast<const uint4*>(text + a);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4 v = p[j];
            const uint32_t ww[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t x = ww[q] ^ 0x0A0A0A0Au;
                uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);   // 0x80 per newline byte
                while (z) {
                    const uint32_t b = (__ffs(z) - 1) >> 3;
                    z &= z - 1;
                    mark(a + 16 * j + 4 * q + b);
                }
            }
        }
    } else {
        for (uint64_t i = a; i < a + ING_PER_THREAD && i < n; ++i)
            if (text[i] == '\n') mark(i);
    }
}

// one thread per record: sequence line -> ceil(read_len/4) packed bytes (+ N mask words); short reads are padded with
// masked A, a longer line is truncated and flagged
__global__ __launch_bounds__(256) void fastq_pack_kernel(const uint8_t* text, uint64_t n, uint64_t n_reads, const unsigned long long* seq_begin,
                                                         const unsigned long long* seq_end, uint32_t read_len, uint8_t* packed,
                                                         uint32_t* nmask, uint32_t* status) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t rb = (read_len + 3) / 4, nmw = (read_len + 31) / 32;
    const uint64_t s = seq_begin[r];
    uint64_t e = seq_end[r];
    if (e > n) e = n;
    if (e > s && text[e - 1] == '\r') --e;     // CRLF files
    uint64_t len = e > s ? e - s : 0;
    if (len > read_len) { atomicOr(status, 1u); len = read_len; }
    uint8_t* o = packed + r * rb;
    uint32_t mword = 0;
    for (uint32_t i0 = 0; i0 < rb * 4; i0 += 4) {
        // four characters at a time (one unaligned 32-bit load while the line and the text last)
        uint32_t four = 0x4E4E4E4Eu;   // "NNNN"
        if (i0 + 4 <= len && s + i0 + 4 <= n) {
            four = *reinterpret_cast<const uint32_t*>(text + s + i0);
        } else {
            for (uint32_t j = 0; j < 4; ++j)
                if (i0 + j < len) four = (four & ~(0xFFu << (8 * j))) | ((uint32_t)text[s + i0 + j] << (8 * j));
        }
        uint32_t byte = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t i = i0 + j;
            if (i >= read_len) break;
            const char ch = (char)((four >> (8 * j)) & 0xFFu);
            byte |= base_code(ch) << (6 - 2 * j);
            const bool acgt = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T' || ch == 'a' || ch == 'c' || ch == 'g' || ch == 't';
            if (!acgt) mword |= 1u << (i & 31);
            if ((i & 31) == 31 || i == read_len - 1) {
                if (nmask) nmask[r * nmw + (i >> 5)] = mword;
                mword = 0;
            }
        }
        o[i0 >> 2] = (uint8_t)byte;
    }
}

// ---- SAM text -> 32-byte alignment records --------------------------------------------------------------------------
// One thread per line.  Columns as the reference reads them (collect_reads_for_gaps.py:76-91: whitespace-split, fields 0-8):
// FLAG, RNAME, POS, MAPQ, CIGAR, RNEXT, PNEXT, TLEN.  RNAME / RNEXT are looked up in the scaffold-name table ('=' -> RNAME's
// index; unknown -> 0xFFFFFFFF); clipflag per GapReadsCollector.is_clipped (:13-26).  Header lines ('@') and lines with fewer
// than nine fields yield no record; the others are numbered in line order.

__device__ __forceinline__ bool is_ws(uint8_t c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

struct SamNames {
    const uint8_t* blob;        // names back to back
    const uint32_t* off;        // n + 1 offsets
    const uint32_t* table;      // open addressing over FNV-1a of the name -> name index, EMPTY32 free
    uint32_t t_log2, n;
};

__device__ __forceinline__ uint32_t fnv1a(const uint8_t* p, uint32_t n) {
    uint32_t h = 2166136261u;
    for (uint32_t i = 0; i < n; ++i) h = (h ^ p[i]) * 16777619u;
    return h;
}

__device__ __forceinline__ uint32_t sam_name_index(const SamNames& N, const uint8_t* s, uint32_t len) {
    if (N.n == 0) return 0xFFFFFFFFu;
    uint32_t sl = fnv1a(s, len) >> (32 - N.t_log2);
    for (;;) {
        const uint32_t idx = N.table[sl];
        if (idx == 0xFFFFFFFFu) return 0xFFFFFFFFu;
        const uint32_t a = N.off[idx], b = N.off[idx + 1];
        if (b - a == len) {
            bool eq = true;
            for (uint32_t i = 0; i < len && eq; ++i) eq = N.blob[a + i] == s[i];
            if (eq) return idx;
        }
        sl = (sl + 1) & ((1u << N.t_log2) - 1);
    }
}

__device__ __forceinline__ long long sam_int(const uint8_t* s, uint32_t len) {
    bool neg = false;
    uint32_t i = 0;
    if (len && (s[0] == '-' || s[0] == '+')) { neg = s[0] == '-'; i = 1; }
    long long v = 0;
    for (; i < len; ++i) {
        const uint32_t d = (uint32_t)s[i] - '0';
        if (d > 9) break;
        v = v * 10 + d;
    }
    return neg ? -v : v;
}

// every newline position (line L ends at newline number L)
__global__ __launch_bounds__(ING_THREADS) void nl_all_kernel(const uint8_t* text, uint64_t n, const unsigned long long* tile_off,
                                                             uint64_t cap, unsigned long long* nl_pos) {
    __shared__ uint32_t wsum[ING_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint64_t a = (uint64_t)blockIdx.x * ING_TILE + (uint64_t)tid * ING_PER_THREAD;
    const uint32_t c = a < n ? count_nl_segment(text, n, a) : 0;
    uint32_t incl = c;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d);
        if ((int)lane >= d) incl += v;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t before = incl - c;
    for (uint32_t i = 0; i < w; ++i) before += wsum[i];
    if (!c) return;
    unsigned long long idx = tile_off[blockIdx.x] + before;
    for (uint64_t i = a; i < a + ING_PER_THREAD && i < n; ++i)
        if (text[i] == '\n') { if (idx < cap) nl_pos[idx] = i; ++idx; }
}

// pass 1: one flag per line (1 = yields a record); pass 2 (after a scan of the flags' block sums): parse and store
__device__ __forceinline__ bool sam_fields(const uint8_t* text, uint64_t a, uint64_t b, uint32_t (&fs)[9], uint32_t (&fl)[9]) {
    // [a, b) = the line without its newline; fs/fl = start (relative to a) and length of the first nine fields
    uint64_t i = a;
    if (i < b && text[i] == '@') return false;
    int nf = 0;
    while (i < b && nf < 9) {
        while (i < b && is_ws(text[i])) ++i;
        if (i >= b) break;
        const uint64_t s = i;
        while (i < b && !is_ws(text[i])) ++i;
        fs[nf] = (uint32_t)(s - a);
        fl[nf] = (uint32_t)(i - s);
        ++nf;
    }
    return nf == 9;
}

__global__ __launch_bounds__(256) void sam_flag_kernel(const uint8_t* text, uint64_t n, const unsigned long long* nl_pos, uint64_t n_lines,
                                                       uint64_t n_nl, uint32_t* block_cnt) {
    __shared__ uint32_t part[4];
    const uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t ok = 0;
    if (l < n_lines) {
        const uint64_t a = l ? nl_pos[l - 1] + 1 : 0, b = l < n_nl ? nl_pos[l] : n;
        uint32_t fs[9], fl[9];
        ok = sam_fields(text, a, b, fs, fl) ? 1u : 0u;
    }
    uint32_t c = ok;
    for (int d = 32; d; d >>= 1) c += __shfl_down(c, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void sam_parse_kernel(const uint8_t* text, uint64_t n, const unsigned long long* nl_pos, uint64_t n_lines,
                                                        uint64_t n_nl, const unsigned long long* block_off, SamNames N, gf_alnrec* recs,
                                                        unsigned long long* line_begin, uint64_t cap) {
    __shared__ uint32_t wsum[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint64_t l = (uint64_t)blockIdx.x * blockDim.x + tid;
    uint32_t fs[9], fl[9];
    uint64_t a = 0;
    bool ok = false;
    if (l < n_lines) {
        a = l ? nl_pos[l - 1] + 1 : 0;
        const uint64_t b = l < n_nl ? nl_pos[l] : n;
        ok = sam_fields(text, a, b, fs, fl);
    }
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) wsum[w] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint64_t idx = block_off[blockIdx.x] + __popcll(bal & ((1ull << lane) - 1));
    for (uint32_t i = 0; i < w; ++i) idx += wsum[i];
    if (!ok || idx >= cap) return;
    const uint8_t* s = text + a;
    gf_alnrec r;
    r.flag = (uint16_t)(sam_int(s + fs[1], fl[1]) & 0xFFFF);
    r.ref = sam_name_index(N, s + fs[2], fl[2]);
    r.pos = (uint32_t)sam_int(s + fs[3], fl[3]);
    const long long mq = sam_int(s + fs[4], fl[4]);
    r.mapq = (uint8_t)(mq > 255 ? 255 : mq);
    uint32_t cf = 0;
    {   // CIGAR: +2 when it ends in S/H, +1 when its first operation is S/H
        const uint8_t* c = s + fs[5];
        const uint32_t cl = fl[5];
        if (c[cl - 1] == 'S' || c[cl - 1] == 'H') cf = 2;
        for (uint32_t i = 0; i < cl; ++i) {
            if (c[i] >= '0' && c[i] <= '9') continue;
            if (c[i] == 'S' || c[i] == 'H') cf += 1;
            break;
        }
    }
    r.clipflag = (uint8_t)cf;
    r.mate_ref = (fl[6] == 1 && s[fs[6]] == '=') ? r.ref : sam_name_index(N, s + fs[6], fl[6]);
    r.mate_pos = (uint32_t)sam_int(s + fs[7], fl[7]);
    r.tlen = (int32_t)sam_int(s + fs[8], fl[8]);
    r.read = idx;
    recs[idx] = r;
    if (line_begin) line_begin[idx] = a;
}

}  // namespace gf

using namespace gf;

extern "C" {

int gf_fastq_pack_dev(gf_ctx* ctx, const void* d_text, size_t n_bytes, int read_len, void* d_packed, size_t cap_reads,
                      void* d_n_mask_or_null, void* d_hdr_begin_or_null, void* d_n_reads, void* d_status) {
    if (!ctx || !d_n_reads || !d_status || (n_bytes && !d_text) || (cap_reads && !d_packed) || read_len <= 0 || read_len > 1000)
        return GF_E_INVAL;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    GF_HIP(ctx, hipMemsetAsync(d_n_reads, 0, 8, ctx->stream));
    GF_HIP(ctx, hipMemsetAsync(d_status, 0, 4, ctx->stream));
    if (n_bytes == 0) return GF_OK;
    const size_t n_tiles = (n_bytes + ING_TILE - 1) / ING_TILE;
    if (n_tiles >= 0xFFFFFFFFull) return GF_E_UNSUPPORTED;
    int rc;
    // workspace: tile counts (u32), tile offsets (u64, n_tiles + 1), then 3 x u64 per record slot
    const size_t b_cnt = (n_tiles * 4 + 63) & ~(size_t)63, b_off = ((n_tiles + 1) * 8 + 63) & ~(size_t)63;
    const size_t slots = cap_reads + 1;
    if ((rc = ensure(ctx, ctx->pool_ws, b_cnt + b_off + 3 * slots * 8 + 64))) return rc;
    uint8_t* ws = (uint8_t*)ctx->pool_ws.p;
    uint32_t* cnt = (uint32_t*)ws;
    unsigned long long* off = (unsigned long long*)(ws + b_cnt);
    unsigned long long* hdr = d_hdr_begin_or_null ? (unsigned long long*)d_hdr_begin_or_null : (unsigned long long*)(ws + b_cnt + b_off);
    unsigned long long* sb = (unsigned long long*)(ws + b_cnt + b_off + slots * 8);
    unsigned long long* se = sb + slots;
    LaunchTimer tm(ctx, GF_KERNEL_INGEST);
    hipLaunchKernelGGL(nl_count_kernel, dim3((unsigned)n_tiles), dim3(ING_THREADS), 0, ctx->stream, (const uint8_t*)d_text, (uint64_t)n_bytes, cnt);
    hipLaunchKernelGGL(nl_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, cnt, (uint32_t)n_tiles, off);
    // the record count needs the newline total on the host (grid size of the pack kernel)
    unsigned long long n_nl = 0;
    uint8_t last = '\n';
    GF_HIP(ctx, hipMemcpyAsync(&n_nl, off + n_tiles, 8, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipMemcpyAsync(&last, (const uint8_t*)d_text + n_bytes - 1, 1, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned long long n_lines = n_nl + (last != '\n' ? 1 : 0);
    unsigned long long n_reads = n_lines / 4;
    uint32_t status = (n_lines & 3) ? 2u : 0u;                      // trailing partial record: ignored, flagged
    // the quality line of the last record has no newline: complete when `text` is a whole file, possibly cut when it is a piece of one
    // (the caller of a piece then drops that record and brings its bytes back with the next piece)
    if (last != '\n' && n_reads && n_nl == 4 * n_reads - 1) status |= 8u;
    if (n_reads > cap_reads) { status |= 4u; }                      // capacity: *d_n_reads reports the number found
    const unsigned long long n_do = n_reads < cap_reads ? n_reads : cap_reads;
    GF_HIP(ctx, hipMemcpyAsync(d_n_reads, &n_reads, 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(nl_mark_kernel, dim3((unsigned)n_tiles), dim3(ING_THREADS), 0, ctx->stream, (const uint8_t*)d_text, (uint64_t)n_bytes,
                       off, (uint64_t)slots, hdr, sb, se);
    if (n_do) {
        hipLaunchKernelGGL(fastq_pack_kernel, dim3((unsigned)((n_do + 255) / 256)), dim3(256), 0, ctx->stream, (const uint8_t*)d_text,
                           (uint64_t)n_bytes, (uint64_t)n_do, sb, se, (uint32_t)read_len, (uint8_t*)d_packed,
                           (uint32_t*)d_n_mask_or_null, (uint32_t*)d_status);
    }
    if (status) {
        uint32_t cur = 0;   // merge the host-side flags into the device word after the kernel's atomicOr
        GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        GF_HIP(ctx, hipMemcpy(&cur, d_status, 4, hipMemcpyDeviceToHost));
        cur |= status;
        GF_HIP(ctx, hipMemcpy(d_status, &cur, 4, hipMemcpyHostToDevice));
    }
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_fastq_pack(gf_ctx* ctx, const char* text, size_t n_bytes, int read_len, uint8_t* packed, size_t cap_reads,
                  uint32_t* n_mask_or_null, uint64_t* hdr_begin_or_null, size_t* n_reads, uint32_t* status) {
    if (!ctx || !n_reads || !status || (n_bytes && !text) || (cap_reads && !packed) || read_len <= 0) return GF_E_INVAL;
    *n_reads = 0;
    *status = 0;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    const size_t rb = gf_packed_read_bytes(read_len), nmw = (size_t)(read_len + 31) / 32;
    const size_t b_text = (n_bytes + 63 + 64) & ~(size_t)63, b_pack = (cap_reads * rb + 63) & ~(size_t)63,
                 b_mask = n_mask_or_null ? (cap_reads * nmw * 4 + 63) & ~(size_t)63 : 0, b_hdr = ((cap_reads + 1) * 8 + 63) & ~(size_t)63;
    int rc;
    if ((rc = ensure(ctx, ctx->stage_in, b_text))) return rc;
    if ((rc = ensure(ctx, ctx->stage_out, b_pack + b_mask + b_hdr + 64))) return rc;
    uint8_t* d_text = (uint8_t*)ctx->stage_in.p;
    uint8_t* d_pack = (uint8_t*)ctx->stage_out.p;
    uint8_t* d_mask = d_pack + b_pack;
    uint8_t* d_hdr = d_mask + b_mask;
    uint8_t* d_cnt = d_hdr + b_hdr;    // [0..8) n_reads, [8..12) status
    if (n_bytes) GF_HIP(ctx, hipMemcpyAsync(d_text, text, n_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = gf_fastq_pack_dev(ctx, d_text, n_bytes, read_len, d_pack, cap_reads, n_mask_or_null ? d_mask : nullptr, d_hdr, d_cnt, d_cnt + 8);
    if (rc) return rc;
    unsigned long long n = 0;
    GF_HIP(ctx, hipMemcpyAsync(&n, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipMemcpyAsync(status, d_cnt + 8, 4, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_reads = (size_t)n;
    if (n > cap_reads) return GF_E_NOSPACE;
    if (n) {
        GF_HIP(ctx, hipMemcpyAsync(packed, d_pack, n * rb, hipMemcpyDeviceToHost, ctx->stream));
        if (n_mask_or_null) GF_HIP(ctx, hipMemcpyAsync(n_mask_or_null, d_mask, n * nmw * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (hdr_begin_or_null) GF_HIP(ctx, hipMemcpyAsync(hdr_begin_or_null, d_hdr, n * 8, hipMemcpyDeviceToHost, ctx->stream));
        GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return GF_OK;
}

int gf_sam_pack(gf_ctx* ctx, const char* text, size_t n_bytes, const char* names_blob, const uint32_t* name_off, size_t n_names,
                gf_alnrec* recs, size_t cap_recs, uint64_t* line_begin_or_null, size_t* n_recs) {
    if (!ctx || !n_recs || (n_bytes && !text) || (cap_recs && !recs) || (n_names && (!names_blob || !name_off))) return GF_E_INVAL;
    *n_recs = 0;
    if (n_bytes == 0) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    // name table on the host: FNV-1a (the device function's twin) -> index
    int t_log2 = 4;
    while (((size_t)1 << t_log2) < 2 * n_names + 2) ++t_log2;
    std::vector<uint32_t> table((size_t)1 << t_log2, 0xFFFFFFFFu);
    for (size_t i = 0; i < n_names; ++i) {
        uint32_t h = 2166136261u;
        for (uint32_t q = name_off[i]; q < name_off[i + 1]; ++q) h = (h ^ (uint8_t)names_blob[q]) * 16777619u;
        uint32_t sl = h >> (32 - t_log2);
        while (table[sl] != 0xFFFFFFFFu) sl = (sl + 1) & (((uint32_t)1 << t_log2) - 1);
        table[sl] = (uint32_t)i;
    }
    const size_t blob_bytes = n_names ? name_off[n_names] : 0;
    const size_t n_tiles = (n_bytes + ING_TILE - 1) / ING_TILE;
    if (n_tiles >= 0xFFFFFFFFull) return GF_E_UNSUPPORTED;
    int rc;
    // stage: text | names blob | name offsets | table
    const size_t b_text = (n_bytes + 127) & ~(size_t)63, b_blob = (blob_bytes + 63) & ~(size_t)63,
                 b_noff = ((n_names + 1) * 4 + 63) & ~(size_t)63, b_tab = (table.size() * 4 + 63) & ~(size_t)63;
    if ((rc = ensure(ctx, ctx->stage_in, b_text + b_blob + b_noff + b_tab))) return rc;
    uint8_t* d_text = (uint8_t*)ctx->stage_in.p;
    uint8_t* d_blob = d_text + b_text;
    uint32_t* d_noff = (uint32_t*)(d_blob + b_blob);
    uint32_t* d_tab = (uint32_t*)((uint8_t*)d_noff + b_noff);
    GF_HIP(ctx, hipMemcpyAsync(d_text, text, n_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (blob_bytes) GF_HIP(ctx, hipMemcpyAsync(d_blob, names_blob, blob_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (n_names) GF_HIP(ctx, hipMemcpyAsync(d_noff, name_off, (n_names + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    GF_HIP(ctx, hipMemcpyAsync(d_tab, table.data(), table.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    // newlines
    const size_t b_cnt = (n_tiles * 4 + 63) & ~(size_t)63, b_off = ((n_tiles + 1) * 8 + 63) & ~(size_t)63;
    if ((rc = ensure(ctx, ctx->pool_ws, b_cnt + b_off + 64))) return rc;
    uint32_t* cnt = (uint32_t*)ctx->pool_ws.p;
    unsigned long long* off = (unsigned long long*)((uint8_t*)ctx->pool_ws.p + b_cnt);
    LaunchTimer tm(ctx, GF_KERNEL_INGEST);
    hipLaunchKernelGGL(nl_count_kernel, dim3((unsigned)n_tiles), dim3(ING_THREADS), 0, ctx->stream, d_text, (uint64_t)n_bytes, cnt);
    hipLaunchKernelGGL(nl_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, cnt, (uint32_t)n_tiles, off);
    unsigned long long n_nl = 0;
    GF_HIP(ctx, hipMemcpyAsync(&n_nl, off + n_tiles, 8, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned long long n_lines = n_nl + (text[n_bytes - 1] != '\n' ? 1 : 0);
    const size_t lblocks = (size_t)((n_lines + 255) / 256);
    // per-line workspace: newline positions, flag block sums + their scan, records, line offsets
    const size_t b_nl = ((n_nl + 1) * 8 + 63) & ~(size_t)63, b_bc = (lblocks * 4 + 63) & ~(size_t)63, b_bo = ((lblocks + 1) * 8 + 63) & ~(size_t)63;
    const size_t b_rec = (cap_recs * sizeof(gf_alnrec) + 63) & ~(size_t)63, b_lb = line_begin_or_null ? (cap_recs * 8 + 63) & ~(size_t)63 : 0;
    if ((rc = ensure(ctx, ctx->stage_out, b_nl + b_bc + b_bo + b_rec + b_lb + 64))) return rc;
    uint8_t* o = (uint8_t*)ctx->stage_out.p;
    unsigned long long* d_nl = (unsigned long long*)o;
    uint32_t* d_bc = (uint32_t*)(o + b_nl);
    unsigned long long* d_bo = (unsigned long long*)(o + b_nl + b_bc);
    gf_alnrec* d_recs = (gf_alnrec*)(o + b_nl + b_bc + b_bo);
    unsigned long long* d_lb = line_begin_or_null ? (unsigned long long*)(o + b_nl + b_bc + b_bo + b_rec) : nullptr;
    hipLaunchKernelGGL(nl_all_kernel, dim3((unsigned)n_tiles), dim3(ING_THREADS), 0, ctx->stream, d_text, (uint64_t)n_bytes, off,
                       (uint64_t)(n_nl + 1), d_nl);
    if (lblocks) {
        if (lblocks >= 0xFFFFFFFFull) return GF_E_UNSUPPORTED;
        hipLaunchKernelGGL(sam_flag_kernel, dim3((unsigned)lblocks), dim3(256), 0, ctx->stream, d_text, (uint64_t)n_bytes, d_nl,
                           (uint64_t)n_lines, (uint64_t)n_nl, d_bc);
        hipLaunchKernelGGL(nl_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_bc, (uint32_t)lblocks, d_bo);
        SamNames N{d_blob, d_noff, d_tab, (uint32_t)t_log2, (uint32_t)n_names};
        hipLaunchKernelGGL(sam_parse_kernel, dim3((unsigned)lblocks), dim3(256), 0, ctx->stream, d_text, (uint64_t)n_bytes, d_nl,
                           (uint64_t)n_lines, (uint64_t)n_nl, d_bo, N, d_recs, d_lb, (uint64_t)cap_recs);
    }
    unsigned long long total = 0;
    if (lblocks) GF_HIP(ctx, hipMemcpyAsync(&total, d_bo + lblocks, 8, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_recs = (size_t)total;
    if (total > cap_recs) return GF_E_NOSPACE;
    if (total) {
        GF_HIP(ctx, hipMemcpyAsync(recs, d_recs, total * sizeof(gf_alnrec), hipMemcpyDeviceToHost, ctx->stream));
        if (line_begin_or_null) GF_HIP(ctx, hipMemcpyAsync(line_begin_or_null, d_lb, total * 8, hipMemcpyDeviceToHost, ctx->stream));
        GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

}  // extern "C"
