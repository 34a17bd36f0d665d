// screen.hip — flank-k-mer screen of packed reads (north_star: "canonical k-mer extract/hash ... of streamed
// reads, flank-k-mer lookup to tag reads").  Predicate shape of IsReadContainingFreqKmers
// (ContigsCompactor-v0.2.0/ContigsMerger/KmerUtils.cpp:215-241) applied per gap on canonical k-mers.
//
// Two kernels:
//   screen_filter_kernel  streams every read once (coalesced 16-B loads of the packed bytes into LDS), probes
//                         only ceil((L-15)/(k-15)) 16-mers per read against an L2-resident bitmap (any k-mer
//                         shared with a flank contains one of those 16-mers), confirms bitmap hits
//                         in an exact 16-mer set, and compacts the surviving read ids with a wave ballot +
//                         prefix count into a candidate list.  This is the HBM-streaming kernel.
//   screen_verify_kernel  one wavefront per candidate read: every k-mer position, canonical form, exact table
//                         lookup, per-gap position count >= min_hits, hit emit.
#include "gf_internal.hpp"

namespace gf {

struct FilterParams {
    const uint8_t* reads;
    uint64_t n_reads;
    uint32_t rb;       // bytes per read
    uint32_t stride2;  // 2 * stride (bits between probed 16-mers)
    uint32_t np;       // probes per read
    const uint32_t* bitmap;
    const uint32_t* sset;
    uint32_t bm_log2, s_log2;
    uint32_t* cand;
    uint32_t* n_cand;
};

// how the filter's bitmap words are fetched: every probe is a 4-byte read of a random 128-B line of an
// L2-resident table, so the L2->CU transfer per probe is what bounds the kernel
enum { LOAD_PLAIN = 0, LOAD_NT = 1, LOAD_SC1 = 2, LOAD_SC01 = 3 };

template <int MODE>
__device__ __forceinline__ uint32_t probe_load(const uint32_t* p) {
    if (MODE == LOAD_NT) return __builtin_nontemporal_load(p);
    if (MODE == LOAD_SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == LOAD_SC01) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return *p;
}

// PU = probes issued back-to-back before their results are consumed
template <int MODE, int PU>
__global__ __launch_bounds__(256) void screen_filter_kernel(FilterParams P) {
    extern __shared__ uint32_t tile[];  // TILE_READS * rb bytes + 16 B pad
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t tile_bytes = TILE_READS * P.rb;
    const uint64_t total_bytes = P.n_reads * P.rb;
    const uint64_t n_tiles = (P.n_reads + TILE_READS - 1) / TILE_READS;
    uint8_t* tb = reinterpret_cast<uint8_t*>(tile);
    const uint32_t smask = (1u << P.s_log2) - 1;
    constexpr uint32_t GROUP = (32 / PU) * PU;  // probes whose results fit one 32-bit mask

    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t byte0 = t * tile_bytes;
        const uint32_t nbytes = (uint32_t)((total_bytes - byte0) < tile_bytes ? (total_bytes - byte0) : tile_bytes);
        const uint32_t n16 = nbytes & ~15u;
        const uint8_t* src = P.reads + byte0;
        for (uint32_t i = tid * 16; i < n16; i += 256 * 16)
            *reinterpret_cast<uint4*>(tb + i) = *reinterpret_cast<const uint4*>(src + i);
        for (uint32_t i = n16 + tid; i < nbytes; i += 256) tb[i] = src[i];
        if (tid < 16) tb[nbytes + tid] = 0;
        __syncthreads();

        const uint64_t r = t * TILE_READS + tid;
        bool cand = false;
        if (r < P.n_reads) {
            const uint32_t bit0 = tid * P.rb * 8;
            for (uint32_t g0 = 0; g0 < P.np && !cand; g0 += GROUP) {
                const uint32_t g1 = g0 + GROUP < P.np ? g0 + GROUP : P.np;
                uint32_t mask = 0;
                for (uint32_t j0 = g0; j0 < g1; j0 += PU) {
                    uint32_t word[PU], hb[PU];
#pragma unroll
                    for (int u = 0; u < PU; ++u) {
                        const uint32_t j = j0 + u;
                        word[u] = 0;
                        hb[u] = 0;
                        if (j < g1) {
                            const uint32_t key = canon16(stream32(tile, bit0 + j * P.stride2));
                            const uint32_t h = hash_s16_bitmap(key, P.bm_log2);
                            hb[u] = h & 31;
                            word[u] = probe_load<MODE>(P.bitmap + (h >> 5));
                        }
                    }
#pragma unroll
                    for (int u = 0; u < PU; ++u) mask |= ((word[u] >> hb[u]) & 1u) << (j0 - g0 + u);
                }
                // level 2: confirm each bitmap hit in the exact canonical-16-mer set
                while (mask && !cand) {
                    const uint32_t j = g0 + __ffs(mask) - 1;
                    mask &= mask - 1;
                    const uint32_t key = canon16(stream32(tile, bit0 + j * P.stride2));
                    uint32_t s = hash_s16_set(key, P.s_log2);
                    uint32_t v;
                    while ((v = P.sset[s]) != EMPTY32) {
                        if (v == key) { cand = true; break; }
                        s = (s + 1) & smask;
                    }
                }
            }
        }
        // wave ballot + prefix count compaction of candidate reads
        const unsigned long long bal = __ballot(cand);
        if (bal) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(P.n_cand, (uint32_t)__popcll(bal));
            base = __shfl(base, 0);
            if (cand) P.cand[base + __popcll(bal & ((1ull << lane) - 1))] = (uint32_t)r;
        }
        __syncthreads();
    }
}

struct VerifyParams {
    const uint8_t* reads;
    const uint32_t* nmask;  // may be null
    uint32_t rb, read_len, k, nmw;
    const uint32_t* cand;
    const uint32_t* n_cand;
    const uint64_t* thi;
    const uint64_t* tlo;
    const uint32_t* tgap;
    uint32_t t_log2;
    uint32_t min_hits;
    uint32_t list_cap;
    gf_hit* out;
    uint32_t cap;
    uint32_t* n_out;
    uint32_t* overflow;
};

template <bool WIDE>
__global__ __launch_bounds__(64) void screen_verify_kernel(VerifyParams P) {
    extern __shared__ uint32_t sm[];  // [0..63] read words (<= 252 bytes + pad) | [64..71] n-mask | list
    __shared__ uint32_t s_n;
    uint32_t* rw = sm;
    uint32_t* nm = sm + 72;
    uint32_t* list = sm + 80;
    const uint32_t lane = threadIdx.x;
    const uint32_t n_cand = *P.n_cand;
    const uint32_t npos = P.read_len - P.k + 1;
    const uint32_t tmask = (1u << P.t_log2) - 1;
    uint8_t* rb8 = reinterpret_cast<uint8_t*>(rw);

    for (uint32_t c = blockIdx.x; c < n_cand; c += gridDim.x) {
        const uint32_t r = P.cand[c];
        const uint8_t* src = P.reads + (uint64_t)r * P.rb;
        for (uint32_t i = lane; i < P.rb + 24; i += 64) rb8[i] = i < P.rb ? src[i] : 0;
        if (lane < 8) nm[lane] = (P.nmask && lane < P.nmw) ? P.nmask[(uint64_t)r * P.nmw + lane] : 0;
        if (lane == 0) s_n = 0;
        __syncthreads();
        for (uint32_t p = lane; p < npos; p += 64) {
            bool ok = true;
            if (P.nmask) {  // any N inside [p, p+k) ?
                for (uint32_t q = p; q < p + P.k; q += 1) {
                    if ((nm[q >> 5] >> (q & 31)) & 1u) { ok = false; break; }
                }
            }
            if (!ok) continue;
            const K128 f = stream_kmer(rw, 2 * p, (int)P.k);
            const K128 cn = canonical(f, (int)P.k);
            uint32_t s = hash_kmer(cn, (int)P.t_log2);
            uint32_t g;
            while ((g = P.tgap[s]) != EMPTY32) {
                bool eq = P.thi[s] == cn.hi;
                if (WIDE) eq = eq && P.tlo[s] == cn.lo;
                if (eq) {
                    const uint32_t idx = atomicAdd(&s_n, 1u);
                    if (idx < P.list_cap) list[idx] = g;
                }
                s = (s + 1) & tmask;
            }
        }
        __syncthreads();
        uint32_t n = s_n;
        if (n > P.list_cap) {
            if (lane == 0) atomicAdd(P.overflow, 1u);
            n = P.list_cap;
        }
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const uint32_t i = i0 + lane;
            bool emit = false;
            uint32_t g = 0;
            if (i < n) {
                g = list[i];
                uint32_t cnt = 0;
                bool first = true;
                for (uint32_t j = 0; j < n; ++j) {
                    if (list[j] == g) {
                        ++cnt;
                        if (j < i) first = false;
                    }
                }
                emit = first && cnt >= P.min_hits;
            }
            const unsigned long long bal = __ballot(emit);
            if (bal) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(P.n_out, (uint32_t)__popcll(bal));
                base = __shfl(base, 0);
                if (emit) {
                    const uint32_t o = base + __popcll(bal & ((1ull << lane) - 1));
                    if (o < P.cap) P.out[o] = gf_hit{g, r};
                }
            }
        }
        __syncthreads();
    }
}

int launch_screen(gf_ctx* ctx, const FlankIndex& ix, const void* d_reads, const void* d_nmask, size_t n_reads,
                  int read_len, int min_hits, void* d_out, size_t cap, void* d_n_out) {
    if (read_len < ix.k || read_len > 1000) return GF_E_INVAL;
    if (n_reads >= 0xFFFFFFFFull || cap > 0xFFFFFFFFull) return GF_E_INVAL;
    const uint32_t rb = (uint32_t)((read_len + 3) / 4);
    if (rb > 250) return GF_E_UNSUPPORTED;
    int rc;
    if ((rc = ensure(ctx, ctx->cand, std::max<size_t>(n_reads, 1) * 4))) return rc;
    if ((rc = ensure(ctx, ctx->counters, 64))) return rc;
    uint32_t* d_cnt = (uint32_t*)ctx->counters.p;  // [0] n_cand, [1] overflow
    GF_HIP(ctx, hipMemsetAsync(d_cnt, 0, 8, ctx->stream));
    GF_HIP(ctx, hipMemsetAsync(d_n_out, 0, 4, ctx->stream));
    if (n_reads == 0) return GF_OK;

    FilterParams F;
    F.reads = (const uint8_t*)d_reads;
    F.n_reads = n_reads;
    F.rb = rb;
    F.stride2 = 2 * ix.stride;
    F.np = (uint32_t)((read_len - 16) / ix.stride + 1);
    F.bitmap = ix.d_bitmap;
    F.sset = ix.d_sset;
    F.bm_log2 = ix.bm_log2;
    F.s_log2 = ix.s_log2;
    F.cand = (uint32_t*)ctx->cand.p;
    F.n_cand = d_cnt;
    const size_t n_tiles = (n_reads + TILE_READS - 1) / TILE_READS;
    const size_t lds = TILE_READS * rb + 16;
    const unsigned grid = (unsigned)std::min<size_t>(n_tiles, (size_t)ctx->n_cu * (ctx->screen_wg_per_cu > 0 ? ctx->screen_wg_per_cu : 8));
    {
        LaunchTimer tm(ctx, GF_KERNEL_SCREEN);
        void (*kern)(FilterParams) = screen_filter_kernel<LOAD_PLAIN, 9>;
        switch (ctx->screen_variant) {
            case 1: kern = screen_filter_kernel<LOAD_NT, 3>; break;
            case 2: kern = screen_filter_kernel<LOAD_SC1, 3>; break;
            case 3: kern = screen_filter_kernel<LOAD_SC01, 3>; break;
            case 4: kern = screen_filter_kernel<LOAD_PLAIN, 3>; break;
            case 5: kern = screen_filter_kernel<LOAD_NT, 9>; break;
            case 6: kern = screen_filter_kernel<LOAD_SC1, 9>; break;
            case 7: kern = screen_filter_kernel<LOAD_PLAIN, 1>; break;
            default: break;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, ctx->stream, F);
    }
    GF_HIP(ctx, hipGetLastError());

    VerifyParams V;
    V.reads = (const uint8_t*)d_reads;
    V.nmask = (const uint32_t*)d_nmask;
    V.rb = rb;
    V.read_len = read_len;
    V.k = ix.k;
    V.nmw = (read_len + 31) / 32;
    if (V.nmw > 8) return GF_E_UNSUPPORTED;
    V.cand = F.cand;
    V.n_cand = d_cnt;
    V.thi = ix.d_thi;
    V.tlo = ix.d_tlo;
    V.tgap = ix.d_tgap;
    V.t_log2 = ix.t_log2;
    V.min_hits = min_hits < 1 ? 1 : min_hits;
    const uint32_t npos = read_len - ix.k + 1;
    size_t want = ix.max_gaps_per_kmer ? (size_t)npos * ix.max_gaps_per_kmer : (size_t)npos * 64;
    V.list_cap = (uint32_t)std::min<size_t>(std::max<size_t>(want, 256), 15000);
    V.out = (gf_hit*)d_out;
    V.cap = (uint32_t)cap;
    V.n_out = (uint32_t*)d_n_out;
    V.overflow = d_cnt + 1;
    const size_t lds2 = (80 + V.list_cap) * 4;
    const unsigned grid2 = (unsigned)ctx->n_cu * 32;  // one wave per block: fill every wave slot
    {
        LaunchTimer tm(ctx, GF_KERNEL_VERIFY);
        if (ix.k > 32)
            hipLaunchKernelGGL(screen_verify_kernel<true>, dim3(grid2), dim3(64), lds2, ctx->stream, V);
        else
            hipLaunchKernelGGL(screen_verify_kernel<false>, dim3(grid2), dim3(64), lds2, ctx->stream, V);
    }
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

}  // namespace gf
