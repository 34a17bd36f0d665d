// index_dev.hip — the flank k-mer index built on the device ("canonical k-mer extract/hash of flanking contigs", north_star).
// The reference has no such index (its flank FASTA only feeds bwa, pick_contigs.py:74-86); the k-mer arithmetic follows
// KmerUtils.cpp:22-87 (2-bit codes, MSB-first) extended to canonical 128-bit k-mers.  index.hip holds the same construction on
// the host (std::sort, 4.6 s for the 19 840 gaps of the human-scale layout); it stays as the comparator of
// tests/test_gpu_parity.py and for the repeat-mask-free path is replaced by this one:
//   extract   one workgroup per flank, one thread per position: validity from the ACGT run around it, the canonical k-mer
//             (key, gap) and the canonical 16-mer occurrence (key << 32 | global position; flank id and pos/strand/room word aside)
//   k-mers    rocPRIM merge sort on (key, gap) -> duplicates and (option) k-mers shared by too many gaps dropped by the head of
//             each run -> open-addressing table: a slot is claimed with one CAS on its gap word (keys are unique by then)
//   16-mers   rocPRIM radix sort on the 64-bit composite -> heads claim a slot of the exact set, record their first occurrence
//             (sval) and set the key's bits in the level-1 bitmap, its 2^24-bit reduction and the LDS copy; every occurrence
//             writes its {flank, info} words in sorted order
//   flanks    packed 2 bits per base for the seed-and-extend verification
// Everything is sized from exact counts (two small read-backs), so the bitmaps and capacities equal the host builder's.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/device/device_merge_sort.hpp>
#include <rocprim/device/device_radix_sort.hpp>

#include "gf_internal.hpp"

namespace gf {

namespace {

struct KEnt {
    uint64_t hi, lo;
    uint32_t gap, pad;
};
struct KEntLess {
    __host__ __device__ bool operator()(const KEnt& a, const KEnt& b) const {
        return a.hi < b.hi || (a.hi == b.hi && (a.lo < b.lo || (a.lo == b.lo && a.gap < b.gap)));
    }
};

__device__ __forceinline__ bool acgt(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

__global__ __launch_bounds__(256) void idx_extract_kernel(const char* fa, const uint32_t* off, int k, KEnt* ent, unsigned long long* s16,
                                                          uint32_t* pinfo, uint32_t* counters) {
    const uint32_t f = blockIdx.x, base = off[f], len = off[f + 1] - base;
    const char* s = fa + base;
    uint32_t nk = 0, ns = 0;
    for (uint32_t p = threadIdx.x; p < len; p += blockDim.x) {
        bool all16 = p + 16 <= len;
        uint32_t w = 0;
        if (all16)
            for (uint32_t q = 0; q < 16; ++q) {
                const char c = s[p + q];
                all16 = all16 && acgt(c);
                w = (w << 2) | base_code(c);
            }
        uint32_t lroom = 0, rroom = 0;
        if (all16) {
            while (lroom < 63 && lroom < p && acgt(s[p - 1 - lroom])) ++lroom;
            while (rroom < 63 && p + 16 + rroom < len && acgt(s[p + 16 + rroom])) ++rroom;
        }
        const uint32_t idx = base + p;
        KEnt e{~0ull, ~0ull, 0xFFFFFFFFu, 0};
        if (all16 && rroom + 16 >= (uint32_t)k) {   // the k-mer [p, p + k) lies in the run
            K128 v{0, 0};
            for (int j = 0; j < k; ++j) {
                const uint64_t c = base_code(s[p + j]);
                if (j < 32) v.hi |= c << (62 - 2 * j);
                else v.lo |= c << (62 - 2 * (j - 32));
            }
            v = canonical(v, k);
            e.hi = v.hi; e.lo = v.lo; e.gap = f >> 1;
            ++nk;
        }
        ent[idx] = e;
        unsigned long long sk = ~0ull;
        if (all16 && lroom + 16 + rroom >= (uint32_t)k) {   // the 16-mer lies in an ACGT run of at least k bases
            const uint32_t key = canon16(w);
            sk = ((unsigned long long)key << 32) | idx;
            pinfo[2 * (size_t)idx] = f;
            pinfo[2 * (size_t)idx + 1] = p | (key != w ? 1u << 16 : 0u) | lroom << 18 | rroom << 24;
            ++ns;
        }
        s16[idx] = sk;
    }
    __shared__ uint32_t red[2];
    if (threadIdx.x < 2) red[threadIdx.x] = 0;
    __syncthreads();
    if (nk) atomicAdd(&red[0], nk);
    if (ns) atomicAdd(&red[1], ns);
    __syncthreads();
    if (threadIdx.x == 0 && red[0]) atomicAdd(&counters[0], red[0]);
    if (threadIdx.x == 1 && red[1]) atomicAdd(&counters[1], red[1]);
}

// sorted (key, gap): the head of every key run keeps one entry per gap, or none when the key has more gaps than allowed
__global__ __launch_bounds__(256) void idx_kmer_mark_kernel(const KEnt* e, uint32_t n, uint32_t max_gaps, uint8_t* keep, uint32_t* counters) {
    uint32_t kept = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (i && e[i - 1].hi == e[i].hi && e[i - 1].lo == e[i].lo) continue;   // not a head
        uint32_t j = i, distinct = 0;
        while (j < n && e[j].hi == e[i].hi && e[j].lo == e[i].lo) {
            if (j == i || e[j].gap != e[j - 1].gap) ++distinct;
            ++j;
        }
        const bool drop = max_gaps && distinct > max_gaps;
        for (uint32_t q = i; q < j; ++q) {
            const bool kq = !drop && (q == i || e[q].gap != e[q - 1].gap);
            keep[q] = kq;
            kept += kq;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) kept += __shfl_xor(kept, d);
    if ((threadIdx.x & 63) == 0 && kept) atomicAdd(&counters[2], kept);
}

__global__ __launch_bounds__(256) void idx_fill_gap_kernel(uint32_t* tab, size_t tcap, uint32_t sw, uint32_t gw) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tcap; i += (size_t)gridDim.x * blockDim.x) tab[i * sw + gw] = EMPTY32;
}

// slot = {hi.lo32, hi.hi32, gap, 0} (k <= 32) or {hi, lo}{gap, 0, 0, 0}; the gap word claims the slot
__global__ __launch_bounds__(256) void idx_kmer_insert_kernel(const KEnt* e, const uint8_t* keep, uint32_t n, uint32_t* tab, int t_log2, int wide) {
    const uint32_t mask = (1u << t_log2) - 1, sw = wide ? 8 : 4, gw = wide ? 4 : 2;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!keep[i]) continue;
        const KEnt x = e[i];
        uint32_t sl = hash_kmer(K128{x.hi, x.lo}, t_log2);
        while (atomicCAS(&tab[(size_t)sl * sw + gw], EMPTY32, x.gap) != EMPTY32) sl = (sl + 1) & mask;
        uint32_t* t = &tab[(size_t)sl * sw];
        t[0] = (uint32_t)x.hi; t[1] = (uint32_t)(x.hi >> 32);
        if (wide) { t[2] = (uint32_t)x.lo; t[3] = (uint32_t)(x.lo >> 32); }
    }
}

__global__ __launch_bounds__(256) void idx_s16_count_kernel(const unsigned long long* s, uint32_t n, uint32_t* counters) {
    uint32_t heads = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        heads += i == 0 || (uint32_t)(s[i - 1] >> 32) != (uint32_t)(s[i] >> 32);
    for (int d = 32; d >= 1; d >>= 1) heads += __shfl_xor(heads, d);
    if ((threadIdx.x & 63) == 0 && heads) atomicAdd(&counters[3], heads);
}

struct S16Build {
    const unsigned long long* s;
    const uint32_t* pinfo;
    uint32_t n;
    uint32_t *sset, *sval, *occ, *bm, *cbm, *mbm;
    int s_log2, bm_log2, lds_log2, mid_log2;
};

__global__ __launch_bounds__(256) void idx_s16_build_kernel(S16Build B) {
    const uint32_t smask = (1u << B.s_log2) - 1;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < B.n; i += gridDim.x * blockDim.x) {
        const uint32_t key = (uint32_t)(B.s[i] >> 32), at = (uint32_t)B.s[i];
        const bool head = i == 0 || (uint32_t)(B.s[i - 1] >> 32) != key;
        const bool last = i + 1 == B.n || (uint32_t)(B.s[i + 1] >> 32) != key;
        if (head) {
            uint32_t sl = hash_s16_set(key, B.s_log2);
            while (atomicCAS(&B.sset[sl], EMPTY32, key) != EMPTY32) sl = (sl + 1) & smask;
            B.sval[sl] = i;
            const uint32_t h = hash_s16_bitmap(key, B.bm_log2);
            // both bits of the key in one word (three in the large bitmaps: hash_s16_bit3)
            atomicOr(&B.bm[h >> 5], (1u << (h & 31)) | (1u << hash_s16_bit2(key)) | (B.bm_log2 >= S16_BIT3_MIN_LOG2 ? 1u << hash_s16_bit3(key, B.bm_log2) : 0u));
            const uint32_t c = h >> (B.bm_log2 - B.lds_log2);
            atomicOr(&B.cbm[c >> 5], 1u << (c & 31));
            if (B.mbm) {
                const uint32_t m = h >> (B.bm_log2 - B.mid_log2);
                atomicOr(&B.mbm[m >> 5], 1u << (m & 31));
            }
        }
        B.occ[2 * (size_t)i] = B.pinfo[2 * (size_t)at];
        B.occ[2 * (size_t)i + 1] = B.pinfo[2 * (size_t)at + 1] | (last ? 1u << 17 : 0u);
    }
}

__global__ void idx_wrap_kernel(uint32_t* sset, uint32_t* sval, uint32_t scap) {   // a 4-slot read never needs the modulo
    if (threadIdx.x < 4) { sset[scap + threadIdx.x] = sset[threadIdx.x]; sval[scap + threadIdx.x] = sval[threadIdx.x]; }
}

__global__ __launch_bounds__(256) void idx_popcount_kernel(const uint32_t* w, size_t n, unsigned long long* out) {
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += __popc(w[i]);
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// flank f: 4 zero words | ceil(len/16) words, base i at bits 30-2(i%16) of word i/16 | 4 zero words (the buffer is pre-zeroed)
__global__ __launch_bounds__(256) void idx_pack_flanks_kernel(const char* fa, const uint32_t* off, const uint32_t* foff, uint32_t* fpk) {
    const uint32_t f = blockIdx.x, base = off[f], len = off[f + 1] - base;
    for (uint32_t wi = threadIdx.x; wi < (len + 15) / 16; wi += blockDim.x) {
        uint32_t v = 0;
        for (uint32_t q = 0; q < 16 && wi * 16 + q < len; ++q) v |= base_code(fa[base + wi * 16 + q]) << (30 - 2 * q);
        fpk[foff[f] + wi] = v;
    }
}

// The grouped copy of the exact set (FlankIndex::d_sgrp).  One thread per slot of the plain set: the neighbours of every occurrence
// of its 16-mer, read from the packed flanks, in the orientation of the CANONICAL 16-mer (an occurrence whose text is the reverse
// complement of the canonical form swaps sides and complements: code -> code ^ 15 = the mask bit-reversed; a 16-mer that is its
// own reverse complement gets both readings).
__device__ __forceinline__ uint32_t ext_mask2(const uint32_t* fw, uint32_t at1, uint32_t at2, uint32_t room) {
    // mask over the codes (nearest << 2 | next) of the two bases at at1 (nearest) and at2; room = how many of them exist
    if (room == 0) return 0u;
    const uint32_t b1 = (fw[at1 >> 4] >> (30 - 2 * (at1 & 15))) & 3u;
    if (room == 1) return 0xFu << (4 * b1);
    const uint32_t b2 = (fw[at2 >> 4] >> (30 - 2 * (at2 & 15))) & 3u;
    return 1u << (4 * b1 + b2);
}
__global__ __launch_bounds__(256) void idx_sgrp_build_kernel(const uint32_t* sset, const uint32_t* sval, const uint32_t* occ, const uint32_t* fpk,
                                                             const uint32_t* foff, uint32_t scap, int s_log2, int ext_ok, uint32_t* sgrp) {
    const uint32_t gmask = (scap >> 2) - 1;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < scap; i += gridDim.x * blockDim.x) {
        const uint32_t key = sset[i];
        if (key == EMPTY32) continue;
        uint32_t L = 0, R = 0;
        if (!ext_ok) L = R = 0xFFFFu;   // (positions do not fit the occurrence words: no statement about the neighbours)
        else {
            const bool pal = revpairs32(~key) == key;
            uint32_t oi = sval[i];
            for (;;) {
                const uint32_t fid = occ[2 * (size_t)oi], info = occ[2 * (size_t)oi + 1];
                const uint32_t pos = info & 0xFFFFu, lroom = (info >> 18) & 63u, rroom = (info >> 24) & 63u;
                const bool fo = (info >> 16) & 1u;
                const uint32_t* fw = fpk + foff[fid];
                const uint32_t tr = ext_mask2(fw, pos + 16, pos + 17, rroom < 2 ? rroom : 2u);            // right of the text 16-mer
                const uint32_t tl = ext_mask2(fw, pos - 1, pos - 2, lroom < 2 ? lroom : 2u);              // left of it (pos >= lroom)
                const uint32_t trc = __brev(tr) >> 16, tlc = __brev(tl) >> 16;                            // complemented codes
                if (!fo || pal) { R |= tr; L |= tl; }
                if (fo || pal) { R |= tlc; L |= trc; }
                if ((info >> 17) & 1u) break;
                ++oi;
            }
        }
        uint32_t g = hash_s16_set(key, s_log2) >> 2, j = 0;
        while (atomicCAS(&sgrp[(size_t)g * 8 + j], EMPTY32, key) != EMPTY32)
            if (++j == 4) { j = 0; g = (g + 1) & gmask; }
        sgrp[(size_t)g * 8 + 4 + j] = L | (R << 16);
    }
}

int ceil_log2_sz(size_t v) {
    int l = 0;
    while (((size_t)1 << l) < v) ++l;
    return l;
}

template <typename T>
int dev_alloc(gf_ctx* ctx, T** p, size_t n) {
    GF_HIP(ctx, hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)));
    return GF_OK;
}

}  // namespace

int build_sgrp_dev(gf_ctx* ctx, FlankIndex& ix) {
    const size_t scap = (size_t)1 << ix.s_log2;
    GF_HIP(ctx, hipMalloc((void**)&ix.d_sgrp, scap * 2 * 4));
    GF_HIP(ctx, hipMemsetAsync(ix.d_sgrp, 0xFF, scap * 2 * 4, ctx->stream));
    hipLaunchKernelGGL(idx_sgrp_build_kernel, dim3((unsigned)ctx->n_cu * 8), dim3(256), 0, ctx->stream, ix.d_sset, ix.d_sval, ix.d_occ, ix.d_fpk, ix.d_foff,
                       (uint32_t)scap, ix.s_log2, ix.ext_ok ? 1 : 0, ix.d_sgrp);
    GF_HIP(ctx, hipGetLastError());
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GF_OK;
}

// Same FlankIndex as build_flank_index's host path (index.hip); table/set layouts differ only in which of several equal-hash
// keys took which slot.
int build_flank_index_dev(gf_ctx* ctx, int k, FlankIndex& ix) {
    const size_t ng = ctx->gaps.size();
    const uint32_t nf = (uint32_t)(2 * ng);
    // flank text + offsets (the host keeps the flanks as strings: gf_set_gaps)
    std::vector<uint32_t> off(nf + 1, 0), foff(nf + 1, 0);
    uint64_t total = 0, fwords = 0;
    ix.ext_ok = true;
    for (uint32_t f = 0; f < nf; ++f) {
        const std::string& s = (f & 1) ? ctx->flank_right[f >> 1] : ctx->flank_left[f >> 1];
        off[f] = (uint32_t)total;
        total += s.size();
        fwords += 4;
        foff[f] = (uint32_t)fwords;
        fwords += (s.size() + 15) / 16 + 4;
        ix.ext_ok = ix.ext_ok && s.size() < 65536;
    }
    if (total >= 0xFFFFFFF0ull || fwords >= 0xFFFFFFF0ull) return GF_E_UNSUPPORTED;
    off[nf] = (uint32_t)total;
    foff[nf] = (uint32_t)fwords;
    std::string blob;
    blob.reserve(total);
    for (uint32_t f = 0; f < nf; ++f) blob += (f & 1) ? ctx->flank_right[f >> 1] : ctx->flank_left[f >> 1];
    const uint32_t P = (uint32_t)total;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    ix.k = k;
    ix.stride = k - 15;
    ix.max_gaps_per_kmer = ctx->max_gaps_per_kmer;
    int rc;
    // ---- workspace (freed at the end): text, offsets, entries in/out, 16-mer composites in/out, per-position words, counters
    char* d_fa = nullptr;
    uint32_t *d_off = nullptr, *d_pinfo = nullptr, *d_cnt = nullptr;
    KEnt *d_e0 = nullptr, *d_e1 = nullptr;
    unsigned long long *d_s0 = nullptr, *d_s1 = nullptr;
    uint8_t* d_keep = nullptr;
    void* d_temp = nullptr;
    auto cleanup = [&]() {
        for (void* p : {(void*)d_fa, (void*)d_off, (void*)d_pinfo, (void*)d_cnt, (void*)d_e0, (void*)d_e1, (void*)d_s0, (void*)d_s1, (void*)d_keep, d_temp})
            if (p) (void)hipFree(p);
    };
#define IDX_TRY(expr)                  \
    do {                               \
        if ((rc = (expr))) { cleanup(); return rc; } \
    } while (0)
#define IDX_HIP(call)                                           \
    do {                                                        \
        hipError_t e__ = (call);                                \
        if (e__ != hipSuccess) { cleanup(); return set_hip_error(ctx, e__, #call); } \
    } while (0)
    IDX_TRY(dev_alloc(ctx, &d_fa, (size_t)P + 64));
    IDX_TRY(dev_alloc(ctx, &d_off, (size_t)nf + 1));
    IDX_TRY(dev_alloc(ctx, &d_pinfo, 2 * (size_t)P));
    IDX_TRY(dev_alloc(ctx, &d_cnt, 16));
    IDX_TRY(dev_alloc(ctx, &d_e0, (size_t)P));
    IDX_TRY(dev_alloc(ctx, &d_e1, (size_t)P));
    IDX_TRY(dev_alloc(ctx, &d_s0, (size_t)P));
    IDX_TRY(dev_alloc(ctx, &d_s1, (size_t)P));
    IDX_TRY(dev_alloc(ctx, &d_keep, (size_t)P));
    hipStream_t st = ctx->stream;
    if (P) IDX_HIP(hipMemcpyAsync(d_fa, blob.data(), P, hipMemcpyHostToDevice, st));
    IDX_HIP(hipMemcpyAsync(d_off, off.data(), ((size_t)nf + 1) * 4, hipMemcpyHostToDevice, st));
    IDX_HIP(hipMemsetAsync(d_cnt, 0, 64, st));
    uint32_t cnt[4] = {0, 0, 0, 0};   // valid k-mer positions, valid 16-mer positions, kept (k-mer, gap) pairs, distinct 16-mers
    const unsigned grid = (unsigned)ctx->n_cu * 8;
    if (nf && P) {
        hipLaunchKernelGGL(idx_extract_kernel, dim3(nf), dim3(256), 0, st, d_fa, d_off, k, d_e0, d_s0, d_pinfo, d_cnt);
        size_t tb1 = 0, tb2 = 0;
        IDX_HIP(rocprim::merge_sort(nullptr, tb1, d_e0, d_e1, (size_t)P, KEntLess(), st));
        IDX_HIP(rocprim::radix_sort_keys(nullptr, tb2, d_s0, d_s1, (size_t)P, 0, 64, st));
        IDX_HIP(hipMalloc(&d_temp, std::max(tb1, tb2) + 256));
        IDX_HIP(rocprim::merge_sort(d_temp, tb1, d_e0, d_e1, (size_t)P, KEntLess(), st));
        IDX_HIP(rocprim::radix_sort_keys(d_temp, tb2, d_s0, d_s1, (size_t)P, 0, 64, st));
        IDX_HIP(hipMemcpyAsync(cnt, d_cnt, 8, hipMemcpyDeviceToHost, st));
        IDX_HIP(hipStreamSynchronize(st));
        if (cnt[0]) hipLaunchKernelGGL(idx_kmer_mark_kernel, dim3(grid), dim3(256), 0, st, d_e1, cnt[0], ctx->max_gaps_per_kmer, d_keep, d_cnt);
        if (cnt[1]) hipLaunchKernelGGL(idx_s16_count_kernel, dim3(grid), dim3(256), 0, st, d_s1, cnt[1], d_cnt);
        IDX_HIP(hipMemcpyAsync(cnt + 2, d_cnt + 2, 8, hipMemcpyDeviceToHost, st));
        IDX_HIP(hipStreamSynchronize(st));
    }
    ix.n_kmers = cnt[2];
    ix.n_s16 = cnt[3];
    ix.n_occ = cnt[1];
    // ---- level 3 table
    ix.t_log2 = std::max(8, ceil_log2_sz(2 * (size_t)cnt[2] + 2));
    const size_t tcap = (size_t)1 << ix.t_log2;
    const int wide = k > 32;
    const size_t slot_words = wide ? 8 : 4;
    uint32_t* d_tab = nullptr;
    IDX_TRY(dev_alloc(ctx, &d_tab, tcap * slot_words));
    ix.d_table = d_tab;
    IDX_HIP(hipMemsetAsync(d_tab, 0, tcap * slot_words * 4, st));
    hipLaunchKernelGGL(idx_fill_gap_kernel, dim3(grid), dim3(256), 0, st, d_tab, tcap, (uint32_t)slot_words, wide ? 4u : 2u);   // free slots
    if (cnt[0]) hipLaunchKernelGGL(idx_kmer_insert_kernel, dim3(grid), dim3(256), 0, st, d_e1, d_keep, cnt[0], d_tab, ix.t_log2, wide);
    // ---- level 2 set, occurrence lists, bitmaps
    ix.s_log2 = std::max(8, ceil_log2_sz(2 * (size_t)cnt[3] + 2));
    const size_t scap = (size_t)1 << ix.s_log2;
    int bl = std::max(24, ceil_log2_sz(16 * (size_t)cnt[3] + 1));
    bl = std::min(28, bl);   // (the 256-bucket filter holds a 1/256 slice in LDS: beyond 2^28 bits fewer bits per key, not another kernel)
    if (ctx->bitmap_log2_override) bl = std::min(31, std::max(10, ctx->bitmap_log2_override));
    ix.bm_log2 = bl;
    ix.lds_log2 = std::min(ctx->screen_lds_log2_max, bl);
    ix.mid_log2 = bl > 24 ? 24 : 0;
    const size_t bwords = ((size_t)1 << bl) / 32, cwords = ((size_t)1 << ix.lds_log2) / 32, mwords = ix.mid_log2 ? ((size_t)1 << 24) / 32 : 0;
    IDX_TRY(dev_alloc(ctx, &ix.d_sset, scap + 4));
    IDX_TRY(dev_alloc(ctx, &ix.d_sval, scap + 4));
    IDX_TRY(dev_alloc(ctx, &ix.d_occ, 2 * (size_t)cnt[1] + 2));
    IDX_TRY(dev_alloc(ctx, &ix.d_bitmap, bwords));
    IDX_TRY(dev_alloc(ctx, &ix.d_bitmap_lds, cwords));
    if (mwords) IDX_TRY(dev_alloc(ctx, &ix.d_bitmap_mid, mwords));
    IDX_TRY(dev_alloc(ctx, &ix.d_fpk, (size_t)fwords + 1));
    IDX_TRY(dev_alloc(ctx, &ix.d_foff, (size_t)nf + 1));
    IDX_HIP(hipMemsetAsync(ix.d_sset, 0xFF, (scap + 4) * 4, st));
    IDX_HIP(hipMemsetAsync(ix.d_sval, 0, (scap + 4) * 4, st));
    IDX_HIP(hipMemsetAsync(ix.d_occ, 0, (2 * (size_t)cnt[1] + 2) * 4, st));
    IDX_HIP(hipMemsetAsync(ix.d_bitmap, 0, bwords * 4, st));
    IDX_HIP(hipMemsetAsync(ix.d_bitmap_lds, 0, cwords * 4, st));
    if (mwords) IDX_HIP(hipMemsetAsync(ix.d_bitmap_mid, 0, mwords * 4, st));
    IDX_HIP(hipMemsetAsync(ix.d_fpk, 0, ((size_t)fwords + 1) * 4, st));
    IDX_HIP(hipMemcpyAsync(ix.d_foff, foff.data(), ((size_t)nf + 1) * 4, hipMemcpyHostToDevice, st));
    if (cnt[1]) {
        S16Build B;
        B.s = d_s1; B.pinfo = d_pinfo; B.n = cnt[1];
        B.sset = ix.d_sset; B.sval = ix.d_sval; B.occ = ix.d_occ; B.bm = ix.d_bitmap; B.cbm = ix.d_bitmap_lds; B.mbm = ix.d_bitmap_mid;
        B.s_log2 = ix.s_log2; B.bm_log2 = bl; B.lds_log2 = ix.lds_log2; B.mid_log2 = ix.mid_log2;
        hipLaunchKernelGGL(idx_s16_build_kernel, dim3(grid), dim3(256), 0, st, B);
    }
    hipLaunchKernelGGL(idx_wrap_kernel, dim3(1), dim3(64), 0, st, ix.d_sset, ix.d_sval, (uint32_t)scap);
    if (nf && P) hipLaunchKernelGGL(idx_pack_flanks_kernel, dim3(nf), dim3(64), 0, st, d_fa, d_off, ix.d_foff, ix.d_fpk);
    // fill ratios of the two reduced bitmaps (kernel choice in launch_screen)
    unsigned long long* d_pop = reinterpret_cast<unsigned long long*>(d_cnt + 8);
    hipLaunchKernelGGL(idx_popcount_kernel, dim3(64), dim3(256), 0, st, ix.d_bitmap_lds, cwords, d_pop);
    if (mwords) hipLaunchKernelGGL(idx_popcount_kernel, dim3(64), dim3(256), 0, st, ix.d_bitmap_mid, mwords, d_pop + 1);
    unsigned long long pop[2] = {0, 0};
    IDX_HIP(hipMemcpyAsync(pop, d_pop, 16, hipMemcpyDeviceToHost, st));
    IDX_HIP(hipStreamSynchronize(st));
    IDX_HIP(hipGetLastError());
    ix.lds_fill = (double)pop[0] / (double)((size_t)1 << ix.lds_log2);
    if (mwords) ix.mid_fill = (double)pop[1] / (double)((size_t)1 << 24);
    IDX_TRY(build_sgrp_dev(ctx, ix));
    cleanup();
#undef IDX_TRY
#undef IDX_HIP
    return GF_OK;
}

}  // namespace gf
