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
    // candidates are buffered per workgroup and appended to the global list with ONE atomic per ~768 of them:
    // a single global counter serialises returning atomics at ~11 ns each (MI355X_MICROARCH.md "dequeue" row)
    constexpr uint32_t CBUF = 1024;
    __shared__ uint32_t cbuf[CBUF];
    __shared__ uint32_t cbuf_n, cbuf_base;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    if (tid == 0) cbuf_n = 0;
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
        // wave ballot + prefix count compaction of candidate reads into the workgroup buffer
        const unsigned long long bal = __ballot(cand);
        if (bal) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&cbuf_n, (uint32_t)__popcll(bal));
            base = __shfl(base, 0);
            if (cand) cbuf[base + __popcll(bal & ((1ull << lane) - 1))] = (uint32_t)r;
        }
        __syncthreads();
        if (cbuf_n > CBUF - TILE_READS) {  // uniform: no room for another tile's worth -> flush
            const uint32_t n = cbuf_n;
            if (tid == 0) cbuf_base = atomicAdd(P.n_cand, n);
            __syncthreads();
            for (uint32_t i = tid; i < n; i += 256) P.cand[cbuf_base + i] = cbuf[i];
            __syncthreads();
            if (tid == 0) cbuf_n = 0;
            __syncthreads();
        }
    }
    __syncthreads();
    {
        const uint32_t n = cbuf_n;
        if (n) {
            if (tid == 0) cbuf_base = atomicAdd(P.n_cand, n);
            __syncthreads();
            for (uint32_t i = tid; i < n; i += 256) P.cand[cbuf_base + i] = cbuf[i];
        }
    }
}

struct VerifyParams {
    const uint32_t* reads32;  // packed reads viewed as little-endian words
    uint64_t n_words;         // whole words of the packed array
    uint32_t tail_bytes;      // bytes after the last whole word (0..3)
    const uint32_t* nmask;    // may be null
    uint32_t rb, read_len, k, nmw;
    const uint32_t* cand;
    const uint32_t* n_cand;
    const uint4* table;       // k <= 32: one uint4 per slot {hi.lo32, hi.hi32, gap, 0}; k > 32: two {hi, lo}, {gap,0,0,0}
    uint32_t t_log2;
    uint32_t min_hits;
    uint32_t list_cap;
    gf_hit* out;
    uint32_t cap;
    uint32_t* n_out;
    uint32_t* overflow;      // counter: candidates whose (position, gap) list exceeded list_cap
    uint32_t* overflow_list; // their read ids (re-verified by a second launch with a large list), or null
};

__device__ __forceinline__ uint32_t packed_word(const VerifyParams& P, uint64_t w) {
    if (w < P.n_words) return P.reads32[w];
    uint32_t v = 0;
    if (w == P.n_words) {
        const uint8_t* t = reinterpret_cast<const uint8_t*>(P.reads32 + P.n_words);
        for (uint32_t i = 0; i < P.tail_bytes; ++i) v |= (uint32_t)t[i] << (8 * i);
    }
    return v;
}

// One wavefront per workgroup.  A wave takes 64 candidates at a time: lane j fetches candidate j's packed read
// into LDS (one round of global latency for 64 reads), then the whole wave verifies the candidates one by one —
// lane = k-mer position, two positions per lane in flight, one 16-B (32-B for k > 32) slot load per probe step.
template <bool WIDE>
__global__ __launch_bounds__(64) void screen_verify_kernel(VerifyParams P) {
    extern __shared__ uint32_t sm[];  // [64][rw] read words | list[list_cap]
    constexpr uint32_t OBUF = 128;    // hits buffered per wave: one global atomic per >= 64 hits (see filter kernel)
    __shared__ gf_hit obuf[OBUF];
    __shared__ uint32_t obuf_n;
    const uint32_t lane = threadIdx.x;
    if (lane == 0) obuf_n = 0;
    const uint32_t rw = (P.rb + 24) / 4 + 1;  // words per staged read, zero padded (stream_kmer reads past the end)
    uint32_t* list = sm + 64 * rw;
    const uint32_t n_cand = *P.n_cand;
    const uint32_t npos = P.read_len - P.k + 1;
    const uint32_t tmask = (1u << P.t_log2) - 1;

    for (uint32_t c0 = blockIdx.x * 64; c0 < n_cand; c0 += gridDim.x * 64) {
        const uint32_t nb = n_cand - c0 < 64 ? n_cand - c0 : 64;
        const uint32_t my_r = lane < nb ? P.cand[c0 + lane] : 0;
        {   // stage my candidate's read, re-aligned to a word boundary
            const uint64_t o = (uint64_t)my_r * P.rb;
            const uint64_t w0 = o >> 2;
            const uint32_t sh = (uint32_t)(o & 3) * 8;
            const uint32_t nw = (P.rb + 3) / 4;
            uint32_t prev = lane < nb ? packed_word(P, w0) : 0;
            for (uint32_t i = 0; i < rw; ++i) {
                uint32_t next = 0;
                if (lane < nb && i < nw) next = packed_word(P, w0 + i + 1);
                uint32_t v = sh ? (prev >> sh) | (next << (32 - sh)) : prev;
                if (i >= nw) v = 0;
                else if (i == nw - 1 && (P.rb & 3)) v &= (1u << ((P.rb & 3) * 8)) - 1;  // drop the next read's bytes
                sm[lane * rw + i] = v;
                prev = next;
            }
        }
        __syncthreads();
        for (uint32_t j = 0; j < nb; ++j) {
            const uint32_t r = __shfl(my_r, j);
            const uint32_t* rwp = sm + j * rw;
            uint32_t n = 0;  // (position, gap) matches of this read; wave-uniform, appended by ballot + prefix count
            for (uint32_t pp = 0; pp < npos; pp += 128) {
                K128 cn[2];
                uint32_t slot[2];
                bool act[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint32_t p = pp + lane + 64 * u;
                    act[u] = p < npos;
                    if (act[u] && P.nmask) {  // any N inside [p, p+k) ?
                        for (uint32_t q = p; q < p + P.k; ++q)
                            if ((P.nmask[(uint64_t)r * P.nmw + (q >> 5)] >> (q & 31)) & 1u) { act[u] = false; break; }
                    }
                    cn[u] = K128{0, 0};
                    slot[u] = 0;
                    if (act[u]) {
                        cn[u] = canonical(stream_kmer(rwp, 2 * p, (int)P.k), (int)P.k);
                        slot[u] = hash_kmer(cn[u], (int)P.t_log2);
                    }
                }
                while (__any(act[0] || act[1])) {  // wave-uniform probe steps; finished lanes idle
                    uint4 a[2], b[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        a[u] = make_uint4(0, 0, EMPTY32, 0);
                        b[u] = make_uint4(EMPTY32, 0, 0, 0);
                        if (act[u]) {
                            if (WIDE) { a[u] = P.table[2 * (uint64_t)slot[u]]; b[u] = P.table[2 * (uint64_t)slot[u] + 1]; }
                            else a[u] = P.table[slot[u]];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const uint32_t g = WIDE ? b[u].x : a[u].z;
                        bool eq = false;
                        if (act[u]) {
                            if (g == EMPTY32) act[u] = false;
                            else {
                                eq = (((uint64_t)a[u].y << 32) | a[u].x) == cn[u].hi;
                                if (WIDE) eq = eq && (((uint64_t)a[u].w << 32) | a[u].z) == cn[u].lo;
                                slot[u] = (slot[u] + 1) & tmask;
                            }
                        }
                        const unsigned long long bal = __ballot(eq);
                        if (bal) {
                            const uint32_t o = n + __popcll(bal & ((1ull << lane) - 1));
                            if (eq && o < P.list_cap) list[o] = g;
                            n += (uint32_t)__popcll(bal);
                        }
                    }
                }
            }
            __syncthreads();
            if (n > P.list_cap) {  // rare (repeat-rich flanks): defer this read to the large-list launch
                if (lane == 0) {
                    const uint32_t o = atomicAdd(P.overflow, 1u);
                    if (P.overflow_list) P.overflow_list[o] = r;
                }
                n = 0;
            }
            // distinct gaps and their position counts.  Up to 256 matches: entries held in registers, one wave
            // step per DISTINCT gap (ballot + popcount); longer lists (pass 2 only): quadratic scan in LDS.
            if (n <= 256) {
                uint32_t v[4];
                bool todo[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t i = lane + 64 * u;
                    todo[u] = i < n;
                    v[u] = todo[u] ? list[i] : 0;
                }
                while (true) {
                    const unsigned long long b0 = __ballot(todo[0]), b1 = __ballot(todo[1]), b2 = __ballot(todo[2]),
                                             b3 = __ballot(todo[3]);
                    if (!(b0 | b1 | b2 | b3)) break;
                    uint32_t g;
                    if (b0) g = __shfl(v[0], __ffsll((long long)b0) - 1);
                    else if (b1) g = __shfl(v[1], __ffsll((long long)b1) - 1);
                    else if (b2) g = __shfl(v[2], __ffsll((long long)b2) - 1);
                    else g = __shfl(v[3], __ffsll((long long)b3) - 1);
                    uint32_t cnt = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool m = todo[u] && v[u] == g;
                        cnt += (uint32_t)__popcll(__ballot(m));
                        if (m) todo[u] = false;
                    }
                    if (cnt >= P.min_hits) {
                        if (lane == 0) obuf[obuf_n] = gf_hit{g, r};
                        __syncthreads();
                        if (lane == 0) obuf_n = obuf_n + 1;
                        __syncthreads();
                    }
                    if (obuf_n >= OBUF - 64) {
                        const uint32_t nn = obuf_n;
                        uint32_t gb = 0;
                        if (lane == 0) gb = atomicAdd(P.n_out, nn);
                        gb = __shfl(gb, 0);
                        for (uint32_t q = lane; q < nn; q += 64)
                            if (gb + q < P.cap) P.out[gb + q] = obuf[q];
                        __syncthreads();
                        if (lane == 0) obuf_n = 0;
                        __syncthreads();
                    }
                }
            } else
            for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                const uint32_t i = i0 + lane;
                bool emit = false;
                uint32_t g = 0;
                if (i < n) {
                    g = list[i];
                    uint32_t cnt = 0;
                    bool first = true;
                    for (uint32_t q = 0; q < n; ++q) {
                        if (list[q] == g) {
                            ++cnt;
                            if (q < i) first = false;
                        }
                    }
                    emit = first && cnt >= P.min_hits;
                }
                const unsigned long long bal = __ballot(emit);
                if (bal) {  // single wave per block: obuf_n is only touched here, in lock-step
                    const uint32_t base = obuf_n;
                    if (emit) obuf[base + __popcll(bal & ((1ull << lane) - 1))] = gf_hit{g, r};
                    __syncthreads();
                    if (lane == 0) obuf_n = base + (uint32_t)__popcll(bal);
                    __syncthreads();
                    if (obuf_n >= OBUF - 64) {
                        const uint32_t nn = obuf_n;
                        uint32_t gb = 0;
                        if (lane == 0) gb = atomicAdd(P.n_out, nn);
                        gb = __shfl(gb, 0);
                        for (uint32_t q = lane; q < nn; q += 64)
                            if (gb + q < P.cap) P.out[gb + q] = obuf[q];
                        __syncthreads();
                        if (lane == 0) obuf_n = 0;
                        __syncthreads();
                    }
                }
            }
            __syncthreads();
        }
        __syncthreads();
    }
    __syncthreads();
    if (obuf_n) {
        const uint32_t nn = obuf_n;
        uint32_t gb = 0;
        if (lane == 0) gb = atomicAdd(P.n_out, nn);
        gb = __shfl(gb, 0);
        for (uint32_t q = lane; q < nn; q += 64)
            if (gb + q < P.cap) P.out[gb + q] = obuf[q];
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
    GF_HIP(ctx, hipMemsetAsync(d_cnt, 0, 16, ctx->stream));  // [0] n_cand [1] error overflow [2] n_cand2
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
    V.reads32 = (const uint32_t*)d_reads;
    V.n_words = ((uint64_t)n_reads * rb) / 4;
    V.tail_bytes = (uint32_t)(((uint64_t)n_reads * rb) & 3);
    V.nmask = (const uint32_t*)d_nmask;
    V.rb = rb;
    V.read_len = read_len;
    V.k = ix.k;
    V.nmw = (read_len + 31) / 32;
    if (V.nmw > 8) return GF_E_UNSUPPORTED;
    V.cand = F.cand;
    V.n_cand = d_cnt;
    V.table = (const uint4*)ix.d_table;
    V.t_log2 = ix.t_log2;
    V.min_hits = min_hits < 1 ? 1 : min_hits;
    const uint32_t npos = read_len - ix.k + 1;
    V.out = (gf_hit*)d_out;
    V.cap = (uint32_t)cap;
    V.n_out = (uint32_t*)d_n_out;
    const unsigned grid2 = (unsigned)ctx->n_cu * 32;  // one wave per block, every wave slot of the chip
    auto launch_verify = [&](const VerifyParams& VP) {
        const size_t lds2 = (64 * ((rb + 24) / 4 + 1) + VP.list_cap) * 4;
        LaunchTimer tm(ctx, GF_KERNEL_VERIFY);
        if (ix.k > 32)
            hipLaunchKernelGGL(screen_verify_kernel<true>, dim3(grid2), dim3(64), lds2, ctx->stream, VP);
        else
            hipLaunchKernelGGL(screen_verify_kernel<false>, dim3(grid2), dim3(64), lds2, ctx->stream, VP);
    };
    // pass 1: small per-wave list (keeps every wave slot of the chip busy); reads that overflow it are queued
    if ((rc = ensure(ctx, ctx->cand2, std::max<size_t>(n_reads, 1) * 4))) return rc;
    V.list_cap = std::max<uint32_t>(256, 2 * npos);
    V.overflow = d_cnt + 2;
    V.overflow_list = (uint32_t*)ctx->cand2.p;
    launch_verify(V);
    GF_HIP(ctx, hipGetLastError());
    // pass 2: the queued reads with a list as large as LDS allows; overflowing that is an error (d_cnt[1])
    size_t want = ix.max_gaps_per_kmer ? (size_t)npos * ix.max_gaps_per_kmer : 15000;
    V.list_cap = (uint32_t)std::min<size_t>(std::max<size_t>(want, 1024), 15000);
    V.cand = (const uint32_t*)ctx->cand2.p;
    V.n_cand = d_cnt + 2;
    V.overflow = d_cnt + 1;
    V.overflow_list = nullptr;
    launch_verify(V);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

}  // namespace gf
