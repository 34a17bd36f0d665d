// screen.hip — flank-k-mer screen of packed reads (north_star: "canonical k-mer extract/hash ... of streamed
// reads, flank-k-mer lookup to tag reads").  Predicate shape of IsReadContainingFreqKmers
// (ContigsCompactor-v0.2.0/ContigsMerger/KmerUtils.cpp:215-241) applied per gap on canonical k-mers.
//
// Filter (streams every read once; probes only ceil((L-15)/(k-15)) 16-mers per read, because any k-mer shared with a flank
// contains one of them; emits candidate read ids):
//   screen_filter_pipe_kernel  default: coarse bitmap in LDS -> level-1 bitmap in L2 -> exact 16-mer set, software-pipelined
//   screen_filter_wave_kernel  the same three levels without the pipeline (more than 10 probes per read; ablation)
//   screen_filter_kernel       no LDS level (key sets that fill the coarse bitmap)
// Verification of the candidates (exact):
//   screen_verify_ext_kernel   seed and extend against the packed flanks (min_hits == 1, no repeat mask)
//   screen_verify_kernel       every k-mer position through the k-mer -> gap table, per-gap position count >= min_hits
#include "gf_internal.hpp"

namespace gf {

struct FilterParams {
    const uint8_t* reads;
    uint64_t n_reads;
    uint32_t rb;       // bytes per read
    uint32_t stride2;  // 2 * stride (bits between probed 16-mers)
    uint32_t first2;   // 2 * first: bit offset of the first probed 16-mer in a read (probe j sits at first + j * stride)
    uint32_t np;       // probes per read
    const uint32_t* bitmap;
    const uint32_t* sset;
    uint32_t bm_log2, s_log2;
    uint32_t* cand;
    uint32_t* n_cand;
    // LDS pre-filter variant: a coarser copy of the bitmap (bit i = OR of the 2^(bm_log2-lds_log2) bits it covers)
    const uint32_t* bitmap_lds;
    uint32_t lds_log2;
    const uint32_t* bitmap_mid;   // plain kernel: L2-resident OR-reduction of a level-1 bitmap larger than the L2 (or null)
    uint32_t mid_log2;
    uint32_t stream_policy; // pipelined kernel: cache policy of the read stream (0 default, 1 nt, 2 sc1, 3 sc0 sc1 nt)
};

// PU = probes issued back-to-back before their results are consumed
template <int PU>
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
                            const uint32_t key = canon16(stream32(tile, bit0 + P.first2 + j * P.stride2));
                            const uint32_t h = hash_s16_bitmap(key, P.bm_log2);
                            hb[u] = (h & 31) | (hash_s16_bit2(key) << 8);   // both bits of the key in its word
                            word[u] = h;                                    // (the level-1 word replaces it below)
                        }
                    }
                    if (P.bitmap_mid) {   // big key sets: an L2-resident reduced bitmap first, the fabric only for what passes
                        uint32_t mw[PU];
#pragma unroll
                        for (int u = 0; u < PU; ++u) mw[u] = (j0 + u < g1) ? P.bitmap_mid[word[u] >> (P.bm_log2 - P.mid_log2 + 5)] : 0;
#pragma unroll
                        for (int u = 0; u < PU; ++u) {
                            const uint32_t c = word[u] >> (P.bm_log2 - P.mid_log2);
                            const bool pass = (j0 + u < g1) && ((mw[u] >> (c & 31)) & 1u);
                            word[u] = pass ? P.bitmap[word[u] >> 5] : 0u;
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < PU; ++u) word[u] = (j0 + u < g1) ? P.bitmap[word[u] >> 5] : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < PU; ++u) {
                        const uint32_t both = (word[u] >> (hb[u] & 31)) & (word[u] >> (hb[u] >> 8));
                        mask |= (both & 1u) << (j0 - g0 + u);
                    }
                }
                // level 2: confirm each bitmap hit in the exact canonical-16-mer set
                while (mask && !cand) {
                    const uint32_t j = g0 + __ffs(mask) - 1;
                    mask &= mask - 1;
                    const uint32_t key = canon16(stream32(tile, bit0 + P.first2 + j * P.stride2));
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

constexpr uint32_t WOBUF = 96;   // candidates buffered per wave (LDS); flushed with one global atomic when >= 32

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // LDS ops of one wave execute in order: only the compiler must not reorder
    __builtin_amdgcn_wave_barrier();
}


// ---- software-pipelined wave kernel ---------------------------------------------------------------------------------
// A wave that streams its tiles and probes them on the spot pays two dependent L2 round trips per 64-read tile (level-1 bitmap word, then the exact
// set), ~2.5 us a tile and wave whatever the probe count; with 11 waves per CU that chain, not bandwidth, set its floor
// (measured: 0.31 ms with no probes, 0.73 ms with one).  Here each wave keeps THREE tiles in flight in registers:
//   stage A (tile t)    stage the tile in LDS, scramble its NP 16-mers, test the coarse LDS bitmap, ISSUE the level-1 loads
//   stage B (tile t-1)  level-1 words have arrived: pick the (up to two) passing probes, ISSUE their exact-set loads
//   stage C (tile t-2)  exact-set slots have arrived: candidate or not, append
// so no iteration waits for a load issued in the same iteration.  A probe is carried as its scrambled key p = key * M
// (bijective): every bitmap index is a shift of p, and key = p * M^-1 when the exact set is consulted.  The exact set is
// read four consecutive slots at a time (linear probing, table padded by three wrap-around slots), which settles almost
// every lookup in one request; the leftovers (3rd+ passing probe of a lane, a run of four foreign keys) take a serial path.
constexpr uint32_t S16_MUL = 0x9E3779B1u;   // multiplier of hash_s16_bitmap
constexpr uint32_t mul_inverse_u32(uint32_t a) {
    uint32_t x = a;   // Newton: x <- x (2 - a x) doubles the correct low bits
    for (int i = 0; i < 6; ++i) x *= 2u - a * x;
    return x;
}
constexpr uint32_t S16_MUL_INV = mul_inverse_u32(S16_MUL);
static_assert(S16_MUL * S16_MUL_INV == 1u, "inverse of the level-1 multiplier");

struct __attribute__((packed, aligned(4))) Slots4 { uint32_t x, y, z, w; };

__device__ __forceinline__ bool sset_walk(const FilterParams& P, uint32_t key, uint32_t sl) {
    const uint32_t smask = (1u << P.s_log2) - 1;
    for (;;) {
        const uint32_t v = P.sset[sl & smask];
        if (v == key) return true;
        if (v == EMPTY32) return false;
        ++sl;
    }
}

// Loads of the pipelined kernel are issued through inline asm and awaited with explicit s_waitcnt: the compiler's own
// counter bookkeeping falls back to vmcnt(0) for loop-carried loads, which would drain the pipeline every step.  vmcnt
// counts vector-memory operations in issue order, so "wait until at most N are outstanding" is safe whenever at least N
// operations were issued after the awaited one; every step therefore issues the same number of loads (idle slots read a
// dummy address), and anything the compiler issues on its own only makes a wait longer, never shorter.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// global loads with a scalar base and a 32-bit per-lane byte offset
__device__ __forceinline__ void vm_load128(u32x4& d, uint32_t voff, const void* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase));
}
__device__ __forceinline__ void vm_load128_nt(u32x4& d, uint32_t voff, const void* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(d) : "v"(voff), "s"(sbase));
}
__device__ __forceinline__ void vm_load128_sc1(u32x4& d, uint32_t voff, const void* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(d) : "v"(voff), "s"(sbase));
}
__device__ __forceinline__ void vm_load128_sc01nt(u32x4& d, uint32_t voff, const void* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1 nt" : "=v"(d) : "v"(voff), "s"(sbase));
}
__device__ __forceinline__ void vm_load32(uint32_t& d, uint32_t voff, const void* sbase) {
    asm volatile("global_load_dword %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase));
}
template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// after a wait: uses of x are ordered behind it
__device__ __forceinline__ void vm_ready(uint32_t& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void vm_ready(u32x4& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));   // the builtin returns int
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
    // VALU write of an SGPR -> VMEM read of it needs 5 wait states; the hazard recogniser does not look into inline asm
    uint32_t hi2 = hi;
    asm volatile("s_nop 4" : "+s"(hi2), "+s"(lo));
    return (const void*)(((uint64_t)hi2 << 32) | lo);
}

// ---- partitioned filter, second form: 256 buckets, so that a bucket's slice of the LEVEL-1 bitmap itself (2^bm_log2 / 256 bits:
// 128 KiB at 2^28) is what pass B holds in LDS — both bits of a key are tested without leaving the CU, and only the ~1 % that pass
// go on to the exact set.  The 16-bucket form above stops 52 % of the pairs in LDS and sends the rest to the L2 at its random-request
// rate (1.4e11/s chip-wide: 1.5 of pass B's 2.2 ms, PMC in DESIGN.md §7).  256 rows per WAVE were tried in round 1 (flush
// bookkeeping: 18 ms); here the unit is the WORKGROUP: sixteen waves scramble one 64-read tile each, every pair takes its rank inside
// its bucket with ONE LDS atomic on a 256-bin histogram, a scan turns the histogram into offsets, the pairs are placed in bucket
// order in LDS and leave as runs (about 16 pairs = 128 B per bucket and iteration) into the workgroup's own part of each bucket.
constexpr uint32_t PF2_NB_LOG2 = 8, PF2_NB = 1u << PF2_NB_LOG2;
constexpr uint32_t PF2_WAVES = 16, PF2_GROUP = 4;                       // waves per workgroup; probes sorted per iteration and read
constexpr uint32_t PF2_TILES = 2;                                      // 64-read tiles per wave and iteration
constexpr uint32_t PF2_BATCH = PF2_WAVES * PF2_TILES * 64 * PF2_GROUP; // 8192 pairs = 64 KiB
static_assert(PF2_WAVES > PF2_NB / 64, "waves 1..4 keep the parts' fill while wave 0 scans");
// Measured on 112.5 M reads, k=51 (2^28-bit bitmap): 16 waves x 1 tile 2.47 ms, 16 x 2 tiles + alternating histograms 2.22 ms (longer
// runs per bucket: 32 pairs = 256 B); 8 waves x 2 tiles with the sort buffer overlaid on the tiles, three workgroups per CU: 2.68 ms
// (128-B runs, three times the parts); 16 x 3 tiles overlaid: 5.6 ms (36 scrambled keys per lane in registers spill).
// What bounds it (same launch, parts of the kernel switched off): loads without stores 1.34 ms, stores without loads 1.57 ms, both
// 2.25-2.37 ms = 8.2 GB at 3.5 TB/s.  Twelve producer waves sorting into one of two 6-byte-entry LDS buffers while four copier waves
// write the other buffer out (stores off every producer's path) took the same 2.377 ms, `nt` stores 2.82 ms, `sc1` stores 2.26 ms:
// the mix of a 4.3-GB read stream and 65 536 scattered 256-B write runs is what the memory system delivers at this rate.


// (pass B of the 256-bucket filter queues the pairs that pass the bitmap and looks them up in the exact set 64 at a time)
constexpr uint32_t PF2_PEND = 128;   // per wave: < 64 waiting + <= 64 from one step

// ---- 256-bucket filter with 4-BYTE pairs.  The 8-byte (key, read) pairs of pf2_* triple the stream (38 B of read -> + 32 B written
// + 32 B read back: 3.06 x the algorithmic bytes at C4) and both passes run at the rate the memory system moves those bytes.  What
// pass B needs of a pair: the 24 key bits below the bucket (both bitmap bits and the exact-set key derive from them) — and the read
// only for the ~0.6 % of the pairs that are in the exact set.  So an entry is  key bits << 8 | OCTET of the read inside the
// workgroup's batch (2048 reads = 256 octets of 8 consecutive reads), and the rest of the read id is recovered, exactly:
//   * which BATCH (tile iteration) a pair belongs to follows from its POSITION in the part: pass A records the part's fill before
//     every group (`fills`, staged in LDS and written as 64-byte rows: 3 % of the pair bytes), pass B searches it for the few
//     pairs that need it;
//   * which of the octet's 8 reads: pf4_resolve_kernel fetches the octet (304 contiguous bytes) and keeps the read(s) that have
//     an aligned 16-mer with this scrambled key — those are exactly the reads the pair can have come from, and each of them IS a
//     candidate (it has a seed in the exact set); the `seen` bit per read keeps one entry per read, as before.
struct Part4Params {
    FilterParams F;
    uint32_t n_writers, cap;      // parts: [bucket][writer][cap] entries
    uint32_t* pairs;              // entry = low 24 bits of the scrambled key << 8 | octet in the batch
    uint32_t* count;              // [bucket][writer]
    uint32_t* fills;              // [bucket][writer][gs]: fill of the part before group g, g = 0 .. n_groups
    uint32_t gs, n_groups, n_grp; // row stride; groups in all; groups per tile iteration
    uint32_t tiles_wg;            // tiles per workgroup and tile iteration
    uint32_t* seen;               // one bit per read
    unsigned long long* cand8;    // pairs found in the exact set: {writer << 56 | position in the part << 32 | batch octet << 24 | key bits}; ~0 = unused
    uint8_t* chunk_b;             // bucket of every PF4_CHUNK entries of that list
    uint32_t* n_cand8;
    uint32_t cap8;
    // Chance candidates: a 16-mer seed against 1.1e7 flank 16-mers lets 0.5 % of the probes through by chance.  The probes are
    // therefore spaced for (16 + ext)-base seeds — ext <= 2 bases to the right of the 16-mer, as many as leave the probe count
    // unchanged — and what stands next to the 16-mer in the flanks rides along with the exact set (FlankIndex::d_sgrp): pass B gets
    // it in the request that answers the look-up and forwards it (`cand8x`), the resolve step — the read is in LDS there — drops
    // the pair when the read's own neighbours are none of the flanks' (each base divides the chance rate by 4).
    const uint32_t* sgrp;         // grouped exact set {key x 4, ext x 4}
    uint32_t ext;                 // bases checked next to the seed (0: none)
    uint32_t* cand8x;             // per list entry: the ext word of the pair's 16-mer
};
constexpr uint32_t PF4_OBUF = 80;     // list entries buffered per wave of pass B (8 + 4 bytes each)
// the pair's key in the grouped exact set, from group g on: found -> its ext word
__device__ __forceinline__ bool pf4_sgrp_walk(const Part4Params& Q, uint32_t key, uint32_t g, uint32_t& ext) {
    const uint32_t gmask = (1u << (Q.F.s_log2 - 2)) - 1;
    for (;;) {
        const uint32_t* G = Q.sgrp + (size_t)(g & gmask) * 8;
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t v = G[j];
            if (v == key) { ext = G[4 + j]; return true; }
            if (v == EMPTY32) return false;
        }
        ++g;
    }
}
// do the bases right of the probe at bit offset `bit` of the read staged at `words` (bit offsets from the start of `words`) agree
// with what some flank has next to this 16-mer?  w16 = the read's 16-mer, key = its canonical form
template <typename P>
__device__ __forceinline__ bool pf4_ext_ok(P words, uint32_t bit, uint32_t w16, uint32_t key, uint32_t ext, uint32_t xw) {
    if (ext == 0) return true;
    const uint32_t nb = stream32(words, bit + 32) >> 28;          // the two bases behind the 16-mer: nearest << 2 | next
    const bool ro = key != w16;                                    // the read shows the reverse complement of the canonical form:
    const uint32_t code = ro ? nb ^ 15u : nb;                      // its right side is the canonical LEFT side, complemented
    const uint32_t mask = ro ? (xw & 0xFFFFu) : (xw >> 16);
    return ext >= 2 ? (mask >> code) & 1u : ((mask >> (code & 12u)) & 15u) != 0;
}
constexpr uint32_t PF4_CHUNK = 256;   // entries of the pair list a wave of pass B reserves at a time
constexpr uint32_t PF4_STAGE = 16;   // groups of fill history staged in LDS (one 64-byte row per bucket and flush)

// does read r have an aligned 16-mer with scrambled key pk?  (slow path: bytes from global memory)
__device__ __forceinline__ bool pf4_read_has_key(const FilterParams& P, uint64_t r, uint32_t pk) {
    const uint8_t* rd = P.reads + r * P.rb;
    for (uint32_t j = 0; j < P.np; ++j) {
        const uint32_t bit = P.first2 + j * P.stride2, by = bit >> 3, sh = bit & 7;
        uint64_t v = 0;
        for (uint32_t q = 0; q < 5; ++q) v = (v << 8) | ((by + q < P.rb) ? rd[by + q] : 0);
        const uint32_t w16 = (uint32_t)((v << sh) >> 8);
        if (canon16(w16) * S16_MUL == pk) return true;
    }
    return false;
}
// octet (read >> 3) of the pair at position `pos` of part (b, w) with batch octet `oc`: the batch is the group g with
// fills[g] <= pos < fills[g + 1] — searched from the proportional guess (the fills grow almost linearly)
__device__ __forceinline__ uint32_t pf4_octet(const Part4Params& Q, uint32_t b, uint32_t w, uint32_t pos, uint32_t oc) {
    const uint32_t* F = Q.fills + ((size_t)b * Q.n_writers + w) * Q.gs;
    const uint32_t G = Q.n_groups, n = F[G];
    uint32_t lo = (uint32_t)((uint64_t)pos * G / (n ? n : 1u)), hi;
    if (lo >= G) lo = G - 1;
    if (F[lo] <= pos) {
        uint32_t st = 1;
        hi = lo + 1;
        while (hi < G && F[hi] <= pos) { lo = hi; st <<= 1; hi = lo + st < G ? lo + st : G; }
    } else {
        uint32_t st = 1;
        hi = lo;
        lo = hi > st ? hi - st : 0;
        while (lo > 0 && F[lo] > pos) { hi = lo; st <<= 1; lo = hi > st ? hi - st : 0; }
    }
    while (hi - lo > 1) {   // F[lo] <= pos < F[hi]
        const uint32_t mid = (lo + hi) >> 1;
        if (F[mid] <= pos) lo = mid; else hi = mid;
    }
    return (uint32_t)(((uint64_t)(lo / Q.n_grp) * Q.n_writers * Q.tiles_wg + (uint64_t)w * Q.tiles_wg) * 8) + oc;
}
// every read of an octet that can have produced the pair is marked a candidate — by one lane on its own (pair list full, or a
// part that ran full in pass A)
__device__ __forceinline__ void pf4_resolve_octet_serial(const Part4Params& Q, uint32_t octet, uint32_t pk) {
    const FilterParams& P = Q.F;
    for (uint32_t sub = 0; sub < 8; ++sub) {
        const uint64_t r = (uint64_t)octet * 8 + sub;
        if (r < P.n_reads && pf4_read_has_key(P, r, pk)) atomicOr(&Q.seen[r >> 5], 1u << (r & 31));
    }
}
// G = probes sorted per group: PF2_GROUP, or the read's whole probe count when that is smaller (k = 51: three — a fourth, dead probe
// slot costs every lane its instructions all the same)
template <uint32_t G, bool BYTES>   // BYTES: the probes start at byte boundaries: one byte permute fetches them
__global__ __launch_bounds__(64 * PF2_WAVES) void pf4_scatter_kernel(Part4Params Q, uint32_t slice_words) {
    extern __shared__ uint32_t sm[];   // [16 waves x PF2_TILES tiles][keys: BATCH x 4 B][octets: BATCH x 1 B][fill stage 256 x 17][hist 3 x 256][offs 258][written 2 x 256]
    const FilterParams& P = Q.F;
    constexpr uint32_t NT = 64 * PF2_WAVES;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t* tiles = sm + wv * PF2_TILES * slice_words;
    uint32_t* skey = sm + PF2_WAVES * PF2_TILES * slice_words;
    uint8_t* sidx = reinterpret_cast<uint8_t*>(skey + PF2_BATCH);
    uint32_t* stage = reinterpret_cast<uint32_t*>(sidx + PF2_BATCH);
    uint32_t* hist3 = stage + PF2_NB * (PF4_STAGE + 1);
    uint32_t* offs = hist3 + 3 * PF2_NB;
    uint32_t* written2 = offs + PF2_NB + 2;
    const uint32_t writer = blockIdx.x;
    const uint32_t tile_bytes = 64 * P.rb;
    const uint64_t total_bytes = P.n_reads * P.rb;
    const uint64_t n_tiles = (P.n_reads + 63) / 64;
    auto part = [&](uint32_t b) { return Q.pairs + ((size_t)b * Q.n_writers + writer) * Q.cap; };
    auto fill_row = [&](uint32_t b) { return Q.fills + ((size_t)b * Q.n_writers + writer) * Q.gs; };
    uint32_t* dummy = Q.pairs + (size_t)PF2_NB * Q.n_writers * Q.cap;   // 64 x 4 bytes behind the parts
    for (uint32_t i = tid; i < PF2_NB; i += NT) { written2[i] = 0; hist3[i] = 0; hist3[PF2_NB + i] = 0; hist3[2 * PF2_NB + i] = 0; }
    constexpr int NPF = 4;   // 64 reads x <= 64 B
    u32x4 pf[PF2_TILES][NPF];
    auto prefetch = [&](uint64_t t0) {
#pragma unroll
        for (uint32_t q = 0; q < PF2_TILES; ++q) {
            const uint64_t t = t0 + q;
            const bool on = t < n_tiles;
            const uint64_t byte0 = on ? t * tile_bytes : 0;
            const uint32_t nbytes = on ? (uint32_t)((total_bytes - byte0) < tile_bytes ? (total_bytes - byte0) : tile_bytes) : 0u;
            const void* base = uniform_ptr(nbytes >= 16 ? (const void*)(P.reads + byte0) : (const void*)Q.count);   // (idle: 16 bytes of the workspace)
#pragma unroll
            for (int c = 0; c < NPF; ++c) {
                const uint32_t i = lane + c * 64;
                vm_load128(pf[q][c], i < (nbytes >> 4) ? i * 16 : 0u, base);
            }
        }
    };
    const uint64_t t_step = (uint64_t)gridDim.x * PF2_WAVES * PF2_TILES;
    const uint64_t n_iter = (n_tiles + t_step - 1) / t_step;
    prefetch(((uint64_t)blockIdx.x * PF2_WAVES + wv) * PF2_TILES);
    uint32_t hsel = 0, wsel = 0, g = 0;
    uint32_t stores_since = 0;   // copy-out stores this wave has issued since its last prefetch (wave-uniform)
    __syncthreads();
    for (uint64_t it = 0; it < n_iter; ++it) {
        const uint64_t t0 = it * t_step + ((uint64_t)blockIdx.x * PF2_WAVES + wv) * PF2_TILES;
        const uint32_t octet0 = (uint32_t)((it * t_step + (uint64_t)blockIdx.x * PF2_WAVES * PF2_TILES) * 8);   // octet of batch index 0
        // this tile's loads were issued before the previous iteration's copy-out stores: those may stay in flight
        switch (stores_since < 12u ? stores_since : 12u) {
            case 0: vm_wait<0>(); break;   case 1: vm_wait<1>(); break;   case 2: vm_wait<2>(); break;   case 3: vm_wait<3>(); break;
            case 4: vm_wait<4>(); break;   case 5: vm_wait<5>(); break;   case 6: vm_wait<6>(); break;   case 7: vm_wait<7>(); break;
            case 8: vm_wait<8>(); break;   case 9: vm_wait<9>(); break;   case 10: vm_wait<10>(); break; case 11: vm_wait<11>(); break;
            default: vm_wait<12>(); break;
        }
#pragma unroll
        for (uint32_t q = 0; q < PF2_TILES; ++q)
#pragma unroll
            for (int c = 0; c < NPF; ++c) vm_ready(pf[q][c]);
        stores_since = 0;
#pragma unroll
        for (uint32_t q = 0; q < PF2_TILES; ++q) {
            const uint64_t t = t0 + q;
            if (t >= n_tiles) continue;
            uint8_t* tb = reinterpret_cast<uint8_t*>(tiles + q * slice_words);
            const uint64_t byte0 = t * tile_bytes;
            const uint32_t nbytes = (uint32_t)((total_bytes - byte0) < tile_bytes ? (total_bytes - byte0) : tile_bytes);
            const uint32_t n16 = nbytes & ~15u;
#pragma unroll
            for (int c = 0; c < NPF; ++c) {
                const uint32_t i = lane + c * 64;
                if (i < (n16 >> 4)) *reinterpret_cast<u32x4*>(tb + (uint64_t)i * 16) = pf[q][c];
            }
            for (uint32_t i = n16 + lane; i < nbytes; i += 64) tb[i] = P.reads[byte0 + i];
            if (lane < 16) tb[nbytes + lane] = 0;
        }
        wave_lds_sync();
        prefetch(t0 + t_step);
        const uint32_t bit0 = lane * P.rb * 8;
        for (uint32_t j0 = 0; j0 < P.np; j0 += G, ++g) {
            uint32_t* hist = hist3 + hsel * PF2_NB;        // all zero (start / zeroed during the copy-out before last)
            const uint32_t* written = written2 + wsel * PF2_NB;
            uint32_t pk[PF2_TILES][G], rank[PF2_TILES][G];
#pragma unroll
            for (uint32_t q = 0; q < PF2_TILES; ++q) {
                const bool live = t0 + q < n_tiles && (t0 + q) * 64 + lane < P.n_reads;
#pragma unroll
                for (uint32_t u = 0; u < G; ++u) {
                    const bool on = live && j0 + u < P.np;
                    pk[q][u] = on ? canon16(BYTES ? stream32_bytes(tiles + q * slice_words, (bit0 + P.first2 + (j0 + u) * P.stride2) >> 3) : stream32(tiles + q * slice_words, bit0 + P.first2 + (j0 + u) * P.stride2)) * S16_MUL : 0u;
                    rank[q][u] = on ? atomicAdd(&hist[pk[q][u] >> (32 - PF2_NB_LOG2)], 1u) : EMPTY32;
                }
            }
            __syncthreads();
            if (wv == 0) {   // exclusive scan of the 256 bins: four per lane
                uint32_t v[4], sum = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] = hist[lane * 4 + q]; sum += v[q]; }
                uint32_t inc = sum;
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t y = __shfl_up(inc, d);
                    if ((int)lane >= d) inc += y;
                }
                uint32_t run = inc - sum;
#pragma unroll
                for (int q = 0; q < 4; ++q) { offs[lane * 4 + q] = run; run += v[q]; }
                if (lane == 63) offs[PF2_NB] = inc;
            } else if (wv <= PF2_NB / 64) {   // the parts' fill before (history) and after this group
                const uint32_t i = tid - 64;
                const uint32_t w0 = written[i], w = w0 + hist[i];
                stage[i * (PF4_STAGE + 1) + (g % PF4_STAGE)] = w0;
                written2[(wsel ^ 1u) * PF2_NB + i] = w < Q.cap ? w : Q.cap;
            }
            __syncthreads();
#pragma unroll
            for (uint32_t q = 0; q < PF2_TILES; ++q)
#pragma unroll
                for (uint32_t u = 0; u < G; ++u)
                    if (rank[q][u] != EMPTY32) {
                        const uint32_t at = offs[pk[q][u] >> (32 - PF2_NB_LOG2)] + rank[q][u];
                        skey[at] = pk[q][u];
                        sidx[at] = (uint8_t)((wv * PF2_TILES + q) * 8 + (lane >> 3));
                    }
            __syncthreads();
            const uint32_t n_pairs = offs[PF2_NB];
            const uint32_t hz = hsel == 0 ? 2 : hsel - 1;
            for (uint32_t i = tid; i < PF2_NB; i += NT) hist3[hz * PF2_NB + i] = 0;                  // the histogram of the group after next
            for (uint32_t i0 = 0; i0 < n_pairs; i0 += NT) {         // (whole waves stay in the loop: one store per wave and trip)
                const uint32_t i = i0 + tid;
                const bool valid = i < n_pairs;
                const uint32_t key = valid ? skey[i] : 0u;
                const uint32_t oc = valid ? (uint32_t)sidx[i] : 0u;
                const uint32_t b = key >> (32 - PF2_NB_LOG2);
                const uint32_t at = valid ? written[b] + (i - offs[b]) : 0u;
                const bool spill = valid && at >= Q.cap;
                {
                    uint32_t* dst = (valid && !spill) ? part(b) + at : dummy + lane;
                    const uint32_t e = (key << 8) | oc;
                    asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(e) : "memory");
                    ++stores_since;
                }
                if (spill) {   // a part that is full (degenerate inputs): tested on the spot
                    const uint32_t h = key >> (32 - P.bm_log2);
                    const uint32_t wd = P.bitmap[h >> 5];
                    if ((wd >> (h & 31)) & (wd >> (key & 31)) & 1u) {
                        const uint32_t k16 = key * S16_MUL_INV;
                        if (sset_walk(P, k16, hash_s16_set(k16, P.s_log2))) pf4_resolve_octet_serial(Q, octet0 + oc, key);
                    }
                }
            }
            if ((g % PF4_STAGE) == PF4_STAGE - 1 || g + 1 == Q.n_groups) {   // the staged fill rows leave as 64-byte pieces
                const uint32_t g_lo = g - g % PF4_STAGE;
                for (uint32_t i = tid; i < PF2_NB * PF4_STAGE; i += NT) {
                    const uint32_t b = i / PF4_STAGE, j = i % PF4_STAGE;
                    if (g_lo + j <= g) fill_row(b)[g_lo + j] = stage[b * (PF4_STAGE + 1) + j];
                }
            }
            hsel = hsel == 2 ? 0 : hsel + 1;
            wsel ^= 1u;
        }
    }
    vm_wait<0>();   // the last prefetch (idle tiles) still targets this wave's registers
    __syncthreads();
    for (uint32_t i = tid; i < PF2_NB; i += NT) {
        const uint32_t w = written2[wsel * PF2_NB + i];
        Q.count[(size_t)i * Q.n_writers + writer] = w;
        fill_row(i)[Q.n_groups] = w;
    }
}

// ---- pass A with WHOLE-LINE stores.  The kernel above writes a bucket's pairs of one group (about 24 at k = 51) where the part's fill
// stands: 96-byte runs that start anywhere.  Measured with the same write pattern beside a read stream (tools/scratch/wbench.hip:
// 65 536 parts filled front to back, 8.2 GB read): unaligned 96-byte runs 2.96 ms (WRITE_SIZE 1.25 x the bytes), aligned 64-byte
// pieces 2.35 ms (1.33 x: the L2 line is 128 bytes), aligned 128-byte lines 2.21 ms with twice the bytes written — 0.2 ms over the
// read stream alone (1.98 ms).  So a bucket's pairs leave as whole, aligned 32-entry lines: what a group leaves over (< 32 entries
// per bucket) waits in an LDS line per bucket (`carry`, two per bucket: the line being filled, and the one that takes the group's
// tail while the filled line is on its way out) and is the head of the bucket's next line.  A pair's position in its part is still
// its generation order (T_old + rank), which is all pass B and the fill history need.
//   A new pair with position p_rel = c + rank relative to the open line (c = T_old & 31, total = c + the group's pairs):
//     p_rel < 32               -> the open line                          carry[sel][b][p_rel]
//     p_rel >= total & ~31     -> the tail: head of the next open line   carry[sel ^ 1][b][p_rel & 31]
//     otherwise                -> a whole line between the two           sent[offs[b] + p_rel - 32]   (few: ~24 pairs per bucket and group)
//   Between the ranks and the placement four waves prepare the buckets' words, one bucket per lane: c, sel, total (`desc`), the
//   fill history, the list of completed open lines (`lga` = line index in `pairs`, `lsrc` = where
//   the line stands in LDS), room in `sent` for the few buckets with whole lines between (an LDS counter: their order is free).
//   Copy-out: the listed lines leave, 32 lanes per line.
// A wave stages ONE tile at a time (its second tile waits in the prefetch registers until the first one's probes are taken): the
// 64 KiB of open lines fit for reads up to 160 bases.  All probes of a read are in one group (np <= 4).
constexpr uint32_t PF4_LINE = 32;
constexpr uint32_t pf4_stage_of(uint32_t G) { return G >= 4 ? 8u : 16u; }   // groups of fill history staged in LDS (what fits beside the lines)
constexpr size_t pf4_lines_lds_bytes(size_t slice_words, uint32_t G) {
    return ((size_t)PF2_WAVES * slice_words + (size_t)G * PF2_WAVES * PF2_TILES * 64 + 2 * PF2_NB * PF4_LINE + PF2_NB * (pf4_stage_of(G) + 1) +
            3 * PF2_NB + 2 * PF2_NB + 4 * PF2_NB + 2 * (PF2_NB + (size_t)G * PF2_WAVES * PF2_TILES * 64 / PF4_LINE) + 8) * 4;
}
template <int LO, int HI>
__device__ __forceinline__ void vm_wait_range(uint32_t n) {   // s_waitcnt vmcnt(clamp(n, LO, HI)), n wave-uniform: the count is an immediate
    if constexpr (LO == HI) vm_wait<LO>();
    else {
        constexpr int MID = (LO + HI + 1) / 2;
        if (n >= (uint32_t)MID) vm_wait_range<MID, HI>(n); else vm_wait_range<LO, MID - 1>(n);
    }
}
// NG = groups per tile iteration: a read's G x NG probe slots are taken from the staged tile at once and sorted G at a time — k = 31 on
// 150-base reads has eight probes per read = two groups of four through the same branch-free machinery (before: pf4_scatter_kernel<4>,
// 16.1 ms per launch at C5 against this kernel's 9.8 ms for C4's three probes).  Probe slots beyond np (np < G x NG) are dead.
template <uint32_t G, bool BYTES, uint32_t NG = 1>   // BYTES: the probes start at byte boundaries (k = 51, 31, ... at 2 bits per base): one byte permute fetches them
__global__ __launch_bounds__(64 * PF2_WAVES) void pf4_scatter_lines_kernel(Part4Params Q, uint32_t slice_words) {
    extern __shared__ uint32_t sm[];   // [16 waves x 1 tile][sent][carry 2 x 256 x 32][fill stage 256 x (ST + 1)][hist 3 x 256][written 2 x 256][desc 256][offs 256][lga, lsrc: 2 x (256 + sent lines)][cnt 8]
    const FilterParams& P = Q.F;
    constexpr uint32_t NT = 64 * PF2_WAVES;
    constexpr uint32_t ST = pf4_stage_of(G), LN = PF4_LINE, LM = PF4_LINE - 1;
    constexpr uint32_t NSENT = G * PF2_WAVES * PF2_TILES * 64, NLINE = PF2_NB + NSENT / LN;
    constexpr uint32_t TMASK = 0x7FFFFFFFu;
    const uint32_t tid0 = threadIdx.x, wv = (uint32_t)__builtin_amdgcn_readfirstlane(tid0 >> 6);
    uint32_t tid = tid0, lane = tid0 & 63;
    uint32_t* tile = sm + wv * slice_words;
    uint32_t* sent = sm + PF2_WAVES * slice_words;
    uint32_t* carry = sent + NSENT;            // (behind `sent`: a pair's place is one index into both)
    uint32_t* stage = carry + 2 * PF2_NB * LN;
    uint32_t* hist3 = stage + PF2_NB * (ST + 1);
    uint32_t* written2 = hist3 + 3 * PF2_NB;   // generated so far (<= cap) | open line's carry buffer << 31
    uint32_t* desc = written2 + 2 * PF2_NB;    // per bucket FOUR words, see the bucket-word phase: where a pair of the group goes is base + its position, base one of three
    uint32_t* lga = desc + 4 * PF2_NB;             // lines that leave in this group: line index in `pairs` — [0, 256): completed open lines; behind: the lines of `sent`
    uint32_t* lsrc = lga + NLINE;              // ... and where the line stands in LDS (word index from `sent`)
    uint32_t* cnt = lsrc + NLINE;              // [g & 1] completed open lines, [2 + (g & 1)] words of `sent` taken
    const uint32_t writer = blockIdx.x;
    const uint32_t tile_bytes = 64 * P.rb;
    const uint64_t total_bytes = P.n_reads * P.rb;
    const uint64_t n_tiles = (P.n_reads + 63) / 64;
    const uint32_t cap_lines = Q.cap >> 5;     // (the capacity is a multiple of 64 entries; the host checked that every line index fits 32 bits)
    auto part = [&](uint32_t b) { return Q.pairs + ((size_t)b * Q.n_writers + writer) * Q.cap; };
    auto fill_row = [&](uint32_t b) { return Q.fills + ((size_t)b * Q.n_writers + writer) * Q.gs; };
    const uint32_t dummy_line = PF2_NB * Q.n_writers * cap_lines;   // 128 bytes behind the parts
    for (uint32_t i = tid; i < PF2_NB; i += NT) { written2[i] = 0; hist3[i] = 0; hist3[PF2_NB + i] = 0; hist3[2 * PF2_NB + i] = 0; }
    if (tid < 8) cnt[tid] = 0;
    constexpr int NPF = 4;   // 64 reads x <= 64 B
    u32x4 pf[PF2_TILES][NPF];
    auto prefetch = [&](uint64_t t, uint32_t q) {
        const bool on = t < n_tiles;
        const uint64_t byte0 = on ? t * tile_bytes : 0;
        const uint32_t nbytes = on ? (uint32_t)((total_bytes - byte0) < tile_bytes ? (total_bytes - byte0) : tile_bytes) : 0u;
        const void* base = uniform_ptr(nbytes >= 16 ? (const void*)(P.reads + byte0) : (const void*)Q.count);   // (idle: 16 bytes of the workspace)
#pragma unroll
        for (int c = 0; c < NPF; ++c) {
            const uint32_t i = lane + c * 64;
            vm_load128(pf[q][c], i < (nbytes >> 4) ? i * 16 : 0u, base);
        }
    };
    const uint64_t t_step = (uint64_t)gridDim.x * PF2_WAVES * PF2_TILES;
    const uint64_t n_iter = (n_tiles + t_step - 1) / t_step;
#pragma unroll
    for (uint32_t q = 0; q < PF2_TILES; ++q) prefetch(((uint64_t)blockIdx.x * PF2_WAVES + wv) * PF2_TILES + q, q);
    uint32_t hsel = 0, wsel = 0;
    uint32_t stores_since = 0;   // copy-out stores this wave has issued since its last prefetch (wave-uniform)
    __syncthreads();
    for (uint64_t it = 0; it < n_iter; ++it) {   // NG groups per iteration: g = it * NG + gi
        asm volatile("" : "+v"(tid), "+v"(lane));   // (opaque: what derives from them is computed where it is used, not kept in registers across the iteration)
        const uint64_t t0 = it * t_step + ((uint64_t)blockIdx.x * PF2_WAVES + wv) * PF2_TILES;
        const uint32_t octet0 = (uint32_t)((it * t_step + (uint64_t)blockIdx.x * PF2_WAVES * PF2_TILES) * 8);   // octet of batch index 0
        uint32_t pka[PF2_TILES][G * NG];      // the raw 16-mers of every probe slot of the iteration's tiles
        const uint32_t bit0 = lane * P.rb * 8;
#pragma unroll
        for (uint32_t q = 0; q < PF2_TILES; ++q) {
            // tile q's loads were issued before the loads of the tiles behind it and the previous copy-out's stores: those may stay in flight
            vm_wait_range<(PF2_TILES - 1) * NPF, (PF2_TILES - 1) * NPF + 15>((uint32_t)__builtin_amdgcn_readfirstlane(stores_since) + (PF2_TILES - 1) * NPF);
#pragma unroll
            for (int c = 0; c < NPF; ++c) vm_ready(pf[q][c]);
            const uint64_t t = t0 + q;
            if (t < n_tiles) {
                uint8_t* tb = reinterpret_cast<uint8_t*>(tile);
                const uint64_t byte0 = t * tile_bytes;
                const uint32_t nbytes = (uint32_t)((total_bytes - byte0) < tile_bytes ? (total_bytes - byte0) : tile_bytes);
                const uint32_t n16 = nbytes & ~15u;
#pragma unroll
                for (int c = 0; c < NPF; ++c) {
                    const uint32_t i = lane + c * 64;
                    if (i < (n16 >> 4)) *reinterpret_cast<u32x4*>(tb + (uint64_t)i * 16) = pf[q][c];
                }
                for (uint32_t i = n16 + lane; i < nbytes; i += 64) tb[i] = P.reads[byte0 + i];
                if (lane < 16) tb[nbytes + lane] = 0;
            }
            wave_lds_sync();
            prefetch(t + t_step, q);
#pragma unroll
            for (uint32_t u = 0; u < G * NG; ++u)     // (a slot beyond np reads inside the staged tile + pad all the same; its pair is never ranked)
                pka[q][u] = BYTES ? stream32_bytes(tile, (bit0 + P.first2 + (NG == 1 || u < P.np ? u : 0u) * P.stride2) >> 3)
                                  : stream32(tile, bit0 + P.first2 + (NG == 1 || u < P.np ? u : 0u) * P.stride2);
            wave_lds_sync();   // the tile's probes are taken (LDS operations of a wave execute in order): the next tile may take its place
        }
        stores_since = 0;
#pragma unroll
      for (uint32_t gi = 0; gi < NG; ++gi) {
        const uint32_t g = (uint32_t)it * NG + gi;
        uint32_t* hist = hist3 + hsel * PF2_NB;        // all zero (start / zeroed during the copy-out before last)
        const uint32_t* written = written2 + wsel * PF2_NB;
        uint32_t pk[PF2_TILES][G], rank[PF2_TILES][G];
#pragma unroll
        for (uint32_t q = 0; q < PF2_TILES; ++q) {
            const bool live = t0 + q < n_tiles && (t0 + q) * 64 + lane < P.n_reads;
            // (with NG == 1 the kernel is launched with G == np — all probes of a read in one group —; a tile's probes share ONE execution
            // mask; the test against np below is the same for every lane)
            if (live) {
#pragma unroll
                for (uint32_t u = 0; u < G; ++u) {
                    if (NG == 1 || gi * G + u < P.np) {
                        pk[q][u] = canon16(pka[q][gi * G + u]) * S16_MUL;
                        rank[q][u] = atomicAdd(&hist[pk[q][u] >> (32 - PF2_NB_LOG2)], 1u);
                    } else { pk[q][u] = 0u; rank[q][u] = EMPTY32; }
                }
            } else {
#pragma unroll
                for (uint32_t u = 0; u < G; ++u) { pk[q][u] = 0u; rank[q][u] = EMPTY32; }
            }
        }
        __syncthreads();
        if (wv >= 1 && wv <= PF2_NB / 64) {   // the buckets' words, one bucket per lane
            const uint32_t i = tid - 64, w = written[i], t_old = w & TMASK, sel = w >> 31, n_new = hist[i];
            const uint32_t room = Q.cap - t_old, ne = n_new < room ? n_new : room;   // (pairs beyond the part's capacity are not placed)
            const uint32_t c = t_old & LM, total = c + ne, full = total & ~LM;
            const bool done = total >= LN;
            stage[i * (ST + 1) + (g % ST)] = t_old;
            written2[(wsel ^ 1u) * PF2_NB + i] = (t_old + ne) | ((sel ^ (done ? 1u : 0u)) << 31);
            // A pair with position p_rel = c + rank goes to word base + p_rel of `sent`, base one of three by where p_rel lies (the
            // placement picks it with two compares): the open line (p_rel < 32), the whole lines between (< full; they start at `of`), the
            // tail = head of the next open line.  The four words leave and are fetched as ONE 16-byte LDS access; the fourth holds
            // c | full / 32 << 5 | the pairs that fit the part << 14 (a rank beyond it: the part is full, the pair is tested on the spot).
            const uint32_t of = full > LN ? atomicAdd(&cnt[2 + (g & 1u)], full - LN) : 0u;    // whole lines between the open line and the tail
            uint4 dv;
            dv.x = NSENT + sel * (PF2_NB * LN) + i * LN;
            dv.y = of - LN;
            dv.z = NSENT + (sel ^ 1u) * (PF2_NB * LN) + i * LN - full;
            dv.w = c | ((full >> 5) << 5) | (ne << 14);        // (full / 32 <= 257: nine bits; ne <= 8 192: fourteen)
            reinterpret_cast<uint4*>(desc)[i] = dv;
            const unsigned long long bal = __ballot(done);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&cnt[g & 1u], (uint32_t)__popcll(bal));   // (the four waves' lists follow each other in any order)
            base = (uint32_t)__builtin_amdgcn_readfirstlane(base);
            if (done) {   // the open line leaves: from carry[sel][b] to position t_old & ~31 of the part
                const uint32_t at = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1));
                lga[at] = (i * Q.n_writers + writer) * cap_lines + (t_old >> 5);
                lsrc[at] = NSENT + sel * (PF2_NB * LN) + i * LN;
            }
        }
        __syncthreads();
        uint32_t spilled = 0, midm = 0;
        const uint32_t dummy_word = (uint32_t)((cnt + 7) - sent);   // (an unused counter word takes the entries of dead lanes and of pairs beyond a full part)
#pragma unroll
        for (uint32_t q = 0; q < PF2_TILES; ++q) {
            uint4 dv[G];
#pragma unroll
            for (uint32_t u = 0; u < G; ++u) dv[u] = reinterpret_cast<const uint4*>(desc)[pk[q][u] >> (32 - PF2_NB_LOG2)];   // the buckets' words first: independent LDS reads
#pragma unroll
            for (uint32_t u = 0; u < G; ++u) {   // straight-line: no branch per pair (the rare cases are collected as bit masks and handled behind the loop)
                const uint32_t key = pk[q][u], m = dv[u].w;
                const uint32_t p_rel = (m & LM) + rank[q][u];
                const bool ok = rank[q][u] < (m >> 14);                      // (a dead lane's rank is EMPTY32: never below)
                const bool lo = p_rel < LN, mid = !lo && (p_rel >> 5) < ((m >> 5) & 0x1FFu);   // (full is a multiple of 32)
                const uint32_t at = (lo ? dv[u].x : mid ? dv[u].y : dv[u].z) + p_rel;
                sent[ok ? at : dummy_word] = (key << 8) | ((wv * PF2_TILES + q) * 8 + (lane >> 3));
                if (!ok) spilled |= 1u << (q * G + u);
                if (ok && mid && (at & LM) == 0) midm |= 1u << (q * G + u);   // the first pair of a whole line between lists it
            }
        }
        if (midm) {   // (static indices: a dynamically indexed pk[][] would live in scratch memory)
#pragma unroll
            for (uint32_t x = 0; x < PF2_TILES * G; ++x)
                if ((midm >> x) & 1u) {
                    const uint32_t q = x / G, u = x % G, b = pk[q][u] >> (32 - PF2_NB_LOG2);
                    const uint4 d4 = reinterpret_cast<const uint4*>(desc)[b];
                    const uint32_t p_rel = (d4.w & LM) + rank[q][u], at = d4.y + p_rel;
                    lga[PF2_NB + (at >> 5)] = (b * Q.n_writers + writer) * cap_lines + (((written[b] & TMASK & ~LM) + p_rel) >> 5);
                    lsrc[PF2_NB + (at >> 5)] = at;
                }
        }
        if (spilled) {   // a part that is full (degenerate inputs): its pair is tested on the spot
#pragma unroll
            for (uint32_t x = 0; x < PF2_TILES * G; ++x)
                if (((spilled >> x) & 1u) && rank[x / G][x % G] != EMPTY32) {
                    const uint32_t q = x / G, u = x % G, key = pk[q][u];
                    const uint32_t h = key >> (32 - P.bm_log2);
                    const uint32_t wd = P.bitmap[h >> 5];
                    if ((wd >> (h & 31)) & (wd >> (key & 31)) & 1u) {
                        const uint32_t k16 = key * S16_MUL_INV;
                        if (sset_walk(P, k16, hash_s16_set(k16, P.s_log2))) pf4_resolve_octet_serial(Q, octet0 + (wv * PF2_TILES + q) * 8 + (lane >> 3), key);
                    }
                }
        }
        __syncthreads();
        const uint32_t n_open = cnt[g & 1u], n_out = (n_open + (cnt[2 + (g & 1u)] >> 5)) * LN;
        if (tid < 2) cnt[2 * tid + ((g + 1) & 1u)] = 0;   // (the next group's)
        const uint32_t hz = hsel == 0 ? 2 : hsel - 1;
        for (uint32_t i = tid; i < PF2_NB; i += NT) hist3[hz * PF2_NB + i] = 0;                  // the histogram of the group after next
        // whole lines leave, eight lanes each (16 bytes per lane), two trips' LDS reads in flight.  Whole waves stay in the loop — one
        // store per wave and trip, lanes without a piece write behind the parts — so that the number of stores in flight is known.
        const uint32_t n_q = n_out >> 2;
        for (uint32_t i0 = 0; i0 < n_q; i0 += 2 * NT) {
            uint32_t ga[2], src[2];
            u32x4 e[2];
#pragma unroll
            for (uint32_t j = 0; j < 2; ++j) {
                const uint32_t i = i0 + j * NT + tid, ln = i >> 3;
                const bool valid = i < n_q;
                const uint32_t at = ln < n_open ? ln : PF2_NB + ln - n_open;
                ga[j] = valid ? lga[at] : dummy_line;
                src[j] = valid ? lsrc[at] + (tid & 7u) * 4 : 0u;
            }
#pragma unroll
            for (uint32_t j = 0; j < 2; ++j) e[j] = *reinterpret_cast<const u32x4*>(sent + src[j]);
#pragma unroll
            for (uint32_t j = 0; j < 2; ++j)
                if (i0 + j * NT < n_q) {
                    const uint32_t* dst = Q.pairs + (((size_t)ga[j] << 5) | ((tid & 7u) * 4));
                    asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(e[j]) : "memory");
                    ++stores_since;
                }
        }
        if ((g % ST) == ST - 1 || g + 1 == Q.n_groups) {   // the staged fill rows leave as 64- (32-) byte pieces
            const uint32_t g_lo = g - g % ST;
            for (uint32_t i = tid; i < PF2_NB * ST; i += NT) {
                const uint32_t b = i / ST, j = i % ST;
                if (g_lo + j <= g) fill_row(b)[g_lo + j] = stage[b * (ST + 1) + j];
            }
        }
        hsel = hsel == 2 ? 0 : hsel + 1;
        wsel ^= 1u;
      }
    }
    vm_wait<0>();   // the last prefetch (idle tiles) still targets this wave's registers
    __syncthreads();
    for (uint32_t t = tid; t < PF2_NB * LN; t += NT) {   // the open lines
        const uint32_t b = t >> 5, w = written2[wsel * PF2_NB + b], T = w & TMASK;
        if ((t & LM) < (T & LM)) part(b)[(T & ~LM) + (t & LM)] = carry[(w >> 31) * (PF2_NB * LN) + t];
    }
    for (uint32_t i = tid; i < PF2_NB; i += NT) {
        const uint32_t w = written2[wsel * PF2_NB + i] & TMASK;
        Q.count[(size_t)i * Q.n_writers + writer] = w;
        fill_row(i)[Q.n_groups] = w;
    }
}

__global__ __launch_bounds__(1024) void pf4_probe_kernel(Part4Params Q) {
    extern __shared__ uint32_t sm[];   // [slice of the level-1 bitmap: 2^(bm_log2 - 8) bits][per wave: 2 x PF4_OBUF words of list entries | 2 x PF2_PEND words of pairs | PF4_OBUF ext words]
    const FilterParams& P = Q.F;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t slice_words = 1u << (P.bm_log2 - PF2_NB_LOG2 - 5);
    const uint32_t sh_w = 45 - P.bm_log2, sh_b1 = 40 - P.bm_log2;   // entry -> word of the slice / first bit inside the word (the second: entry bits 8..12)
    unsigned long long* obuf = reinterpret_cast<unsigned long long*>(sm + slice_words + wv * (3 * PF4_OBUF + 2 * PF2_PEND));
    unsigned long long* pend = obuf + PF4_OBUF;
    uint32_t* obx = reinterpret_cast<uint32_t*>(pend + PF2_PEND);
    uint32_t obuf_n = 0, pend_n = 0;   // wave-uniform
    const unsigned long long lt = (1ull << lane) - 1;
    for (uint32_t b = blockIdx.x; b < PF2_NB; b += gridDim.x) {
        // The pair list is reserved in CHUNKS of PF4_CHUNK entries per wave and bucket (a single global counter serialises returning
        // atomics at ~11 ns each: one atomic per 32-64 entries was 0.8 ms per 112.5 M reads, the whole pass); a chunk belongs to ONE
        // bucket (chunk_b), so an entry needs no bucket bits; what is left of a wave's last chunk is filled with the invalid entry ~0.
        uint32_t ch_at = 0, ch_end = 0;   // wave-uniform: this wave's chunk
        bool list_full = false;           // wave-uniform: a reservation came back beyond the list — no more reservations (the counter
                                          // then stops at most one chunk per wave beyond the capacity: it cannot wrap)
        auto flush = [&]() {
            uint32_t done = 0;
            while (done < obuf_n) {
                if (ch_at == ch_end) {
                    uint32_t gb = Q.cap8;
                    if (lane == 0 && !list_full) {
                        gb = atomicAdd(Q.n_cand8, PF4_CHUNK);
                        if (gb < Q.cap8) Q.chunk_b[gb / PF4_CHUNK] = (uint8_t)b;
                    }
                    ch_at = __shfl(gb, 0);
                    ch_end = ch_at + PF4_CHUNK;
                    list_full = ch_at >= Q.cap8;
                }
                const uint32_t room = ch_end - ch_at, m = obuf_n - done < room ? obuf_n - done : room;
                for (uint32_t q = lane; q < m; q += 64) {
                    const unsigned long long cp = obuf[done + q];
                    if (ch_at + q < Q.cap8) { Q.cand8[ch_at + q] = cp; Q.cand8x[ch_at + q] = obx[done + q]; }
                    else pf4_resolve_octet_serial(Q, pf4_octet(Q, b, (uint32_t)(cp >> 56), (uint32_t)(cp >> 32) & 0xFFFFFFu, (uint32_t)(cp >> 24) & 255u),
                                                  (b << 24) | ((uint32_t)cp & 0xFFFFFFu));
                }
                ch_at += m;
                done += m;
            }
            obuf_n = 0;
            wave_lds_sync();
        };
        // Exact-set look-up of the last min(64, pend_n) queued pairs {writer | position | batch octet | key bits}, one per lane, in TWO
        // steps: `ask` sends for the four slots at the key's home (one request), `take` — called before the next `ask`, a batch of
        // pairs later — reads the answer, so the wave streams on while the set answers (asked and taken in one go, the look-ups were
        // 0.24 of pass B's 0.91 ms per 675 M pairs: every 6 400 pairs a wave stood still for a round trip to the set).
        uint4 asked_v = make_uint4(EMPTY32, EMPTY32, EMPTY32, EMPTY32), asked_x = make_uint4(0, 0, 0, 0);   // the key's home group: four keys, their ext words
        unsigned long long asked_pr = 0;
        bool asked = false;           // this lane has a look-up in flight
        auto take = [&]() {
            bool cand = false;
            uint32_t xw = 0;
            if (asked) {
                const uint32_t key = ((b << 24) | ((uint32_t)asked_pr & 0xFFFFFFu)) * S16_MUL_INV;
                if (asked_v.x == key) { cand = true; xw = asked_x.x; }
                else if (asked_v.y == key) { cand = true; xw = asked_x.y; }
                else if (asked_v.z == key) { cand = true; xw = asked_x.z; }
                else if (asked_v.w == key) { cand = true; xw = asked_x.w; }
                else if (asked_v.x != EMPTY32 && asked_v.y != EMPTY32 && asked_v.z != EMPTY32 && asked_v.w != EMPTY32)   // rare: a full group of foreign keys
                    cand = pf4_sgrp_walk(Q, key, (hash_s16_set(key, P.s_log2) >> 2) + 1, xw);
            }
            asked = false;
            const unsigned long long bal = __ballot(cand);
            if (bal) {
                if (obuf_n + (uint32_t)__popcll(bal) > PF4_OBUF) flush();
                if (cand) { obuf[obuf_n + __popcll(bal & lt)] = asked_pr; obx[obuf_n + __popcll(bal & lt)] = xw; }
                obuf_n += (uint32_t)__popcll(bal);
                wave_lds_sync();
                if (obuf_n >= 32) flush();
            }
        };
        auto ask = [&]() {
            take();
            const uint32_t base = pend_n > 64 ? pend_n - 64 : 0;
            if (base + lane < pend_n) {
                asked_pr = pend[base + lane];
                const uint32_t key = ((b << 24) | ((uint32_t)asked_pr & 0xFFFFFFu)) * S16_MUL_INV;
                const uint4* G = reinterpret_cast<const uint4*>(Q.sgrp + (size_t)(hash_s16_set(key, P.s_log2) >> 2) * 8);   // 32 aligned bytes: one request
                asked_v = G[0];
                asked_x = G[1];
                asked = true;
            }
            pend_n = base;
            wave_lds_sync();
        };
        __syncthreads();
        for (uint32_t i = tid * 4; i < slice_words; i += 1024 * 4)
            *reinterpret_cast<uint4*>(sm + i) = *reinterpret_cast<const uint4*>(P.bitmap + (size_t)b * slice_words + i);
        __syncthreads();
        for (uint32_t w = wv; w < Q.n_writers; w += 16) {
            const uint32_t n = Q.count[(size_t)b * Q.n_writers + w];
            const uint32_t* src = Q.pairs + ((size_t)b * Q.n_writers + w) * Q.cap;
            constexpr int PB = 16;   // entries per lane and trip: four 16-byte loads (eight: 0.86 vs 0.81 ms per 675 M pairs) (lane = four consecutive entries of each 1 024-byte row)
            uint4 nx4[PB / 4];
            auto fetch = [&](uint32_t i0) {
#pragma unroll
                for (int c = 0; c < PB / 4; ++c) {
                    const uint32_t at = i0 + (c * 64 + lane) * 4;
                    nx4[c] = at < n ? *reinterpret_cast<const uint4*>(src + at) : make_uint4(0, 0, 0, 0);   // (a part's capacity is a multiple of 64 entries)
                }
            };
            fetch(0);
            for (uint32_t i0 = 0; i0 < n; i0 += PB * 64) {
                uint32_t pr[PB];
#pragma unroll
                for (int c = 0; c < PB / 4; ++c) { pr[4 * c] = nx4[c].x; pr[4 * c + 1] = nx4[c].y; pr[4 * c + 2] = nx4[c].z; pr[4 * c + 3] = nx4[c].w; }
                if (i0 + PB * 64 < n) fetch(i0 + PB * 64);
                // both bits of every key in its word of the slice: the trip's sixteen LDS reads first (independent), then the tests.  An entry
                // is key bits << 8 | octet and the bucket's 8 bits are the same for the whole slice, so the word index is ONE shift of the
                // entry (its top bm_log2 - 13 bits: always inside the slice) and each bit index one bit-field extract — the pass is bound by
                // its vector instructions (PMC: 30 per pair), not by LDS or HBM
                uint32_t wd[PB], passm = 0;
#pragma unroll
                for (int u = 0; u < PB; ++u) wd[u] = sm[pr[u] >> sh_w];
                // (the pass is used with bitmaps of 2^27 / 2^28 bits: their keys carry a third bit, kmer_dev.hpp::hash_s16_bit3 — with it a
                //  third fewer pairs go on to the exact set, whose 128-byte lines were 6.8 of this pass's 17.6 GB at C4)
#pragma unroll
                for (int u = 0; u < PB; ++u) {
                    const uint32_t b1 = (pr[u] >> sh_b1) & 31u, b2 = (pr[u] >> 8) & 31u;
                    passm |= ((wd[u] >> b1) & (wd[u] >> b2) & (wd[u] >> s16_bit3_of(b1, b2)) & 1u) << u;
                }
                if (i0 + PB * 64 > n) {   // the part's last trip: entries behind its end do not count
#pragma unroll
                    for (int u = 0; u < PB; ++u)
                        if (i0 + ((u >> 2) * 64 + lane) * 4 + (u & 3) >= n) passm &= ~(1u << u);
                }
                // The passing entries (1.2 % at C4: a dozen per trip, nearly all lanes none or one) leave lane by lane, lowest bit first: one
                // round per entry of the lane that holds most — two on average — where a ballot per entry SLOT ran sixteen rounds, nine of
                // them with a taker.  The entry is picked from its sixteen registers by the four bits of its index.
                for (;;) {
                    const bool has = passm != 0;
                    const unsigned long long bal = __ballot(has);
                    if (!bal) break;
                    const uint32_t u = has ? (uint32_t)__builtin_ctz(passm) : 0u;
                    // (bit-field inserts under an all-ones / all-zeros mask: written as `c ? a : b` the compiler makes a dynamically indexed
                    //  array of it — 128 bytes of scratch memory per lane)
                    const uint32_t m0 = 0u - (u & 1u), m1 = 0u - ((u >> 1) & 1u), m2 = 0u - ((u >> 2) & 1u), m3 = 0u - ((u >> 3) & 1u);
                    uint32_t s8[8], s4[4], s2[2];
#pragma unroll
                    for (int q = 0; q < 8; ++q) s8[q] = (pr[2 * q + 1] & m0) | (pr[2 * q] & ~m0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) s4[q] = (s8[2 * q + 1] & m1) | (s8[2 * q] & ~m1);
#pragma unroll
                    for (int q = 0; q < 2; ++q) s2[q] = (s4[2 * q + 1] & m2) | (s4[2 * q] & ~m2);
                    const uint32_t e = (s2[1] & m3) | (s2[0] & ~m3);
                    const uint32_t pos = i0 + ((u >> 2) * 64 + lane) * 4 + (u & 3);
                    if (has) pend[pend_n + __popcll(bal & lt)] = ((unsigned long long)w << 56) | ((unsigned long long)pos << 32) | ((e & 255u) << 24) | (e >> 8);
                    passm &= passm - 1u;
                    pend_n += (uint32_t)__popcll(bal);                    // < 64 + 64 <= PF2_PEND
                    wave_lds_sync();
                    if (pend_n >= 64) ask();
                }
            }
        }
        while (pend_n) ask();
        take();
        if (obuf_n) flush();
        for (uint32_t q = ch_at + lane; q < ch_end; q += 64)
            if (q < Q.cap8) Q.cand8[q] = ~0ull;
    }
}

// A pair that is in the exact set -> the reads it can have come from.  Its batch follows from its position in the part (the
// part's fill history), its octet from the entry; every read of the octet that has an aligned 16-mer with the pair's scrambled key
// IS a candidate: its bit in `seen` is set.  Eight lanes per pair (lane = read of the octet), eight pairs per lane group in flight;
// a read's aligned 16-mers are fetched with one unaligned 8-byte load each.
__global__ __launch_bounds__(256) void pf4_resolve_kernel(Part4Params Q) {
    extern __shared__ uint32_t sm[];   // per wave: 8 octets x (octet words rounded up to 4, + 4)
    const FilterParams& P = Q.F;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6, sub = lane & 7, slot = lane >> 3;
    const uint32_t ow = 2 * P.rb;                         // words per octet (8 reads x rb bytes; the octet starts 16-byte aligned)
    const uint32_t ow4 = (ow + 3) / 4;                    // 16-byte pieces
    const uint32_t row = ow4 * 4 + 4;
    uint32_t* stg = sm + (wv * 8 + slot) * row;
    const uint32_t n = *Q.n_cand8 < Q.cap8 ? *Q.n_cand8 : Q.cap8;
    const uint64_t total_bytes = P.n_reads * P.rb;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t e0 = wave * 64; e0 < n; e0 += n_waves * 64) {
        // lane = pair: its octet (one dependent chain of look-ups per lane, 64 in flight per wave) ...
        uint32_t my_octet = 0xFFFFFFFFu, my_pk = 0, my_xw = 0;
        {
            const uint64_t e = e0 + lane;
            const unsigned long long cp = e < n ? Q.cand8[e] : ~0ull;
            if (cp != ~0ull) {   // (~0: unused tail of a wave's chunk)
                const uint32_t b = Q.chunk_b[e / PF4_CHUNK];
                my_xw = Q.cand8x[e];
                my_pk = (b << 24) | ((uint32_t)cp & 0xFFFFFFu);
                my_octet = pf4_octet(Q, b, (uint32_t)(cp >> 56), (uint32_t)(cp >> 32) & 0xFFFFFFu, (uint32_t)(cp >> 24) & 255u);
            }
        }
        // ... then eight lanes per pair, eight pairs per round: the lanes stage the octet (8 reads, contiguous) in LDS with aligned
        // 16-byte loads, lane = read scrambles its own aligned 16-mers
        for (int u = 0; u < 8; ++u) {
            const uint32_t octet = __shfl(my_octet, u * 8 + slot), pk = __shfl(my_pk, u * 8 + slot), xw = __shfl(my_xw, u * 8 + slot);
            const bool valid = octet != 0xFFFFFFFFu;
            if (!__any(valid)) continue;
            const uint64_t byte0 = (uint64_t)octet * 8 * P.rb;
            for (uint32_t c = sub; c < ow4; c += 8) {
                uint4 v = make_uint4(0, 0, 0, 0);
                const uint64_t at = byte0 + (uint64_t)c * 16;
                if (valid) {
                    if (at + 16 <= total_bytes) v = *reinterpret_cast<const uint4*>(P.reads + at);
                    else {
                        uint32_t t[4] = {0, 0, 0, 0};
                        for (uint32_t q = 0; q < 16; ++q) if (at + q < total_bytes) t[q >> 2] |= (uint32_t)P.reads[at + q] << (8 * (q & 3));
                        v = make_uint4(t[0], t[1], t[2], t[3]);
                    }
                }
                *reinterpret_cast<uint4*>(stg + c * 4) = v;
            }
            wave_lds_sync();
            const uint64_t r = (uint64_t)octet * 8 + sub;
            bool hit = false;
            if (valid && r < P.n_reads)
                for (uint32_t j = 0; j < P.np && !hit; ++j) {
                    const uint32_t bit = sub * P.rb * 8 + P.first2 + j * P.stride2, w16 = stream32(stg, bit), key = canon16(w16);
                    hit = key * S16_MUL == pk && pf4_ext_ok(stg, bit, w16, key, Q.ext, xw);
                }
            if (hit) atomicOr(&Q.seen[r >> 5], 1u << (r & 31));
            wave_lds_sync();
        }
    }
}

// the candidate list = the reads whose `seen` bit is set: every workgroup compacts one contiguous slice of the bitmap (count, one
// global atomic for the slice, then write)
__global__ __launch_bounds__(256) void pf4_list_kernel(Part4Params Q) {
    __shared__ uint32_t s_w[4], s_base;
    const FilterParams& P = Q.F;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint64_t n_words = (P.n_reads + 31) / 32;
    const uint64_t per = ((n_words + gridDim.x - 1) / gridDim.x + 255) & ~(uint64_t)255;   // whole 256-word rows per workgroup
    const uint64_t w0 = (uint64_t)blockIdx.x * per, w1 = w0 + per < n_words ? w0 + per : n_words;
    // pass 1: the slice's candidates (coalesced: thread t takes word t of every 256-word row)
    uint32_t c = 0;
    for (uint64_t i = w0 + tid; i < w1; i += 256) c += (uint32_t)__popc(Q.seen[i]);
    for (int d = 32; d; d >>= 1) c += __shfl_down(c, d);
    if (lane == 0) s_w[wv] = c;
    __syncthreads();
    if (tid == 0) {
        const uint32_t tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        s_base = tot ? atomicAdd(P.n_cand, tot) : 0u;
    }
    __syncthreads();
    uint32_t base = s_base;
    // pass 2 (the slice is in L2 now): row by row, a block scan of the words' popcounts places every read in read order
    for (uint64_t r0 = w0; r0 < w1; r0 += 256) {
        uint32_t v = r0 + tid < w1 ? Q.seen[r0 + tid] : 0u;
        const uint32_t n = (uint32_t)__popc(v);
        uint32_t inc = n;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(inc, d);
            if ((int)lane >= d) inc += y;
        }
        __syncthreads();                       // (the previous row's wave totals have been read)
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        uint32_t off = base + inc - n;
        for (uint32_t q = 0; q < wv; ++q) off += s_w[q];
        while (v) {
            const uint32_t bit = (uint32_t)__ffs(v) - 1;
            v &= v - 1;
            P.cand[off++] = (uint32_t)((r0 + tid) * 32 + bit);
        }
        base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    }
}

template <int NP>
struct PipeW {   // stage A -> B: scrambled keys, level-1 words, coarse-pass bits of one tile
    uint32_t p[NP], w[NP], pm;
    uint32_t t;  // tile index, PIPE_NONE when empty (tiles < 2^26 since reads < 2^32)
};
struct PipeS {   // stage B -> C: the (up to two) exact-set lookups of one tile
    uint32_t key0, key1;
    u32x4 v0, v1;
    uint32_t fl;  // bit 0/1: lookup 0/1 in use, bit 2: already a candidate (slow path)
    uint32_t t;
};
constexpr uint32_t PIPE_NONE = 0xFFFFFFFFu;

template <int NCH, int NP, bool EXACT>   // EXACT: the launch has exactly NP probes per read (no per-probe np test)
__global__ __launch_bounds__(512) void screen_filter_pipe_kernel(FilterParams P, uint32_t slice_words) {
    static_assert(NCH >= 1, "whole 16-byte chunks per lane");
    extern __shared__ uint32_t sm[];  // [coarse bitmap][per wave: 64 reads + pad | WOBUF candidate ids]
    const uint32_t tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x, nw = nthr >> 6;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);   // wave index, kept scalar
    const uint32_t bm_words = 1u << (P.lds_log2 - 5);
    for (uint32_t i = tid * 4; i < bm_words; i += nthr * 4)
        *reinterpret_cast<uint4*>(sm + i) = *reinterpret_cast<const uint4*>(P.bitmap_lds + i);
    uint32_t* tile = sm + bm_words + w * (slice_words + WOBUF);
    uint32_t* obuf = tile + slice_words;
    uint8_t* tb = reinterpret_cast<uint8_t*>(tile);
    uint32_t obuf_n = 0;   // wave-uniform
    const uint32_t tile_bytes = 64 * P.rb;
    const uint64_t total_bytes = P.n_reads * P.rb;
    const uint32_t n_tiles = (uint32_t)((P.n_reads + 63) / 64);
    const bool same = P.bm_log2 == P.lds_log2;
    const bool bytes_ok = ((P.stride2 | P.first2) & 7) == 0;
    const uint32_t sh_lds = 32 - P.lds_log2, sh_bm = 32 - P.bm_log2;
    const uint32_t l2mask = same ? 0u : ~3u;
    const void* dummy = P.bitmap_lds;   // >= 16 readable bytes
    u32x4 pfA[NCH], pfB[NCH];
    auto prefetch = [&](u32x4 (&pf)[NCH], uint32_t t) {   // always NCH loads
        uint32_t n16 = 0;
        const void* base = dummy;
        if (t < n_tiles) {
            const uint64_t byte0 = (uint64_t)t * tile_bytes;
            n16 = (uint32_t)((total_bytes - byte0) < tile_bytes ? (total_bytes - byte0) : tile_bytes) >> 4;
            if (n16) base = P.reads + byte0;
        }
        base = uniform_ptr(base);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint32_t i = lane + c * 64;
            const uint32_t off = i < n16 ? i * 16 : 0u;
            if (P.stream_policy == 1) vm_load128_nt(pf[c], off, base);
            else if (P.stream_policy == 2) vm_load128_sc1(pf[c], off, base);
            else if (P.stream_policy == 3) vm_load128_sc01nt(pf[c], off, base);
            else vm_load128(pf[c], off, base);
        }
    };
    const uint32_t tstride = gridDim.x * nw;   // < 2^26
    uint32_t t = blockIdx.x * nw + w;
    auto next_tile = [&](uint32_t x, uint32_t steps) {   // x + steps * tstride, saturating at n_tiles
        const uint64_t y = (uint64_t)x + (uint64_t)steps * tstride;
        return y < n_tiles ? (uint32_t)y : n_tiles;
    };
    prefetch(pfA, t);
    prefetch(pfB, next_tile(t, 1));

    PipeW<NP> W0, W1;
    PipeS S0, S1;
    W0.t = W1.t = PIPE_NONE;
    S0.t = S1.t = PIPE_NONE;
    __syncthreads();   // coarse bitmap staged

    // Loads issued per step, in order: NP (level-1 words), 2 (exact-set slots), NCH (prefetch of the tile two steps on).
    // vmcnt retires in issue order; every wait below names how many loads were issued after the one it needs.
    //   prefetch issued at the end of step s-2   -> staged at the top of step s
    //   level-1 words issued in stage A of s-1   -> tested in stage B of step s
    //   exact-set slots issued in stage B of s-1 -> tested in stage C of step s
    constexpr int PER_STEP = NP + 2 + NCH;
    constexpr int YOUNGER_THAN_PF = PER_STEP;      // all of step s-1
    constexpr int YOUNGER_THAN_S = NCH + NP;       // prefetch of step s-1, stage A of step s
    static_assert(YOUNGER_THAN_PF + PER_STEP <= 63, "vmcnt is a 6-bit counter");
    uint32_t n_steps = 0;
    const uint32_t rbase = lane * P.rb;

    // one step: stage A on tile t (fills WA), stage C on SC and stage B on WB -> SB (both filled one step ago)
    auto step = [&](u32x4 (&pf)[NCH], PipeW<NP>& WA, PipeW<NP>& WB, PipeS& SB, PipeS& SC) {
        const bool have = t < n_tiles;
        if (n_steps >= 2) vm_wait<YOUNGER_THAN_PF>(); else vm_wait<0>();
        ++n_steps;
#pragma unroll
        for (int c = 0; c < NCH; ++c) vm_ready(pf[c]);
        if (have) {
            const uint64_t byte0 = (uint64_t)t * tile_bytes;
            const uint32_t nbytes = (uint32_t)((total_bytes - byte0) < tile_bytes ? (total_bytes - byte0) : tile_bytes);
            const uint32_t n16 = nbytes >> 4;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const uint32_t i = lane + c * 64;
                if (i < n16) *reinterpret_cast<u32x4*>(tb + i * 16) = pf[c];
            }
            if (nbytes & 15u) {   // only the very last tile of the array
                const uint8_t* src = P.reads + byte0;
                for (uint32_t i = n16 * 16 + lane; i < nbytes; i += 64) tb[i] = src[i];
            }
            if (lane < 16) tb[nbytes + lane] = 0;
        }
        wave_lds_sync();
        {
            const uint32_t en = (have && (uint64_t)t * 64 + lane < P.n_reads) ? 0xFFFFFFFFu : 0u;
            uint32_t bw[NP], w32[NP];
            // the uniform byte-aligned / bit-aligned choice is made ONCE around the unrolled loop, so that all NP tile reads
            // are issued back to back (a branch per probe put an s_waitcnt lgkmcnt(0) behind every single ds_read2)
            if (bytes_ok) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const uint32_t jj = EXACT || (uint32_t)j < P.np ? (uint32_t)j : 0u;
                    w32[j] = stream32_bytes(tile, rbase + ((P.first2 + jj * P.stride2) >> 3));
                }
            } else {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const uint32_t jj = EXACT || (uint32_t)j < P.np ? (uint32_t)j : 0u;
                    w32[j] = stream32(tile, rbase * 8 + P.first2 + jj * P.stride2);
                }
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) WA.p[j] = canon16(w32[j]) * S16_MUL;
#pragma unroll
            for (int j = 0; j < NP; ++j) bw[j] = sm[WA.p[j] >> (sh_lds + 5)];
            uint32_t pm = 0;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                uint32_t pass = (uint32_t)__builtin_amdgcn_sbfe((int)bw[j], WA.p[j] >> sh_lds, 1);   // 0 or all ones
                pass &= EXACT || (uint32_t)j < P.np ? en : 0u;
                pm |= pass & (1u << j);
                vm_load32(WA.w[j], (WA.p[j] >> (sh_bm + 3)) & (pass & l2mask), P.bitmap);   // offset 0 when idle
            }
            WA.pm = pm;
            WA.t = have ? t : PIPE_NONE;
            wave_lds_sync();   // all lanes are done with the slice before the next step overwrites it
        }
        // ---- everything issued up to stage B of the previous step has arrived
        vm_wait<YOUNGER_THAN_S>();
        // ---- stage C
        if (SC.t != PIPE_NONE) {
            vm_ready(SC.v0); vm_ready(SC.v1);
            const bool hit0 = SC.v0.x == SC.key0 || SC.v0.y == SC.key0 || SC.v0.z == SC.key0 || SC.v0.w == SC.key0;
            const bool hit1 = SC.v1.x == SC.key1 || SC.v1.y == SC.key1 || SC.v1.z == SC.key1 || SC.v1.w == SC.key1;
            const bool open0 = SC.v0.x == EMPTY32 || SC.v0.y == EMPTY32 || SC.v0.z == EMPTY32 || SC.v0.w == EMPTY32;
            const bool open1 = SC.v1.x == EMPTY32 || SC.v1.y == EMPTY32 || SC.v1.z == EMPTY32 || SC.v1.w == EMPTY32;
            const bool on0 = SC.fl & 1u, on1 = SC.fl & 2u;
            bool cand = (SC.fl & 4u) || (on0 && hit0) || (on1 && hit1);
            const bool walk0 = on0 && !hit0 && !open0, walk1 = on1 && !hit1 && !open1;
            if ((walk0 || walk1) && !cand) {   // rare: four foreign keys in a row
                if (walk0) cand = sset_walk(P, SC.key0, hash_s16_set(SC.key0, P.s_log2) + 4);
                if (walk1 && !cand) cand = sset_walk(P, SC.key1, hash_s16_set(SC.key1, P.s_log2) + 4);
            }
            const unsigned long long bal = __ballot(cand);
            if (bal) {
                const uint32_t cnt = (uint32_t)__popcll(bal);
                if (cand) obuf[obuf_n + __popcll(bal & ((1ull << lane) - 1))] = SC.t * 64 + lane;
                obuf_n += cnt;   // <= 31 + 64 <= WOBUF
                wave_lds_sync();
                if (obuf_n >= 32) {
                    uint32_t gb = 0;
                    if (lane == 0) gb = atomicAdd(P.n_cand, obuf_n);
                    gb = __shfl(gb, 0);
                    for (uint32_t i = lane; i < obuf_n; i += 64) P.cand[gb + i] = obuf[i];
                    obuf_n = 0;
                    wave_lds_sync();
                }
            }
            SC.t = PIPE_NONE;
        }
        // ---- stage B (always two loads)
        uint32_t p0 = 0, p1 = 0, fl = 0;
        if (WB.t != PIPE_NONE) {
#pragma unroll
            for (int j = 0; j < NP; ++j) vm_ready(WB.w[j]);
            uint32_t mask = 0;
#pragma unroll
            for (int j = 0; j < NP; ++j)   // both bits of the key in its word (index.hip sets two per key)
                mask |= (__builtin_amdgcn_ubfe(WB.w[j], WB.p[j] >> sh_bm, 1) & __builtin_amdgcn_ubfe(WB.w[j], WB.p[j], 1)) << j;
            mask = same ? WB.pm : (mask & WB.pm);
            const uint32_t m2 = mask & (mask - 1), m3 = m2 & (m2 - 1);
            const int j0 = __ffs(mask) - 1, j1 = __ffs(m2) - 1;   // -1 when absent
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                p0 = j0 == j ? WB.p[j] : p0;
                p1 = j1 == j ? WB.p[j] : p1;
            }
            fl = (mask ? 1u : 0u) | (m2 ? 2u : 0u);
            if (m3) {   // rare: third and later passing probes of a lane, looked up on the spot
                bool slow = false;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    if (((m3 >> j) & 1u) && !slow) {
                        const uint32_t key = WB.p[j] * S16_MUL_INV;
                        slow = sset_walk(P, key, hash_s16_set(key, P.s_log2));
                    }
                }
                fl |= slow ? 4u : 0u;
            }
        }
        SB.fl = fl;
        SB.key0 = p0 * S16_MUL_INV; SB.key1 = p1 * S16_MUL_INV;
        vm_load128(SB.v0, (fl & 1u) ? hash_s16_set(SB.key0, P.s_log2) * 4 : 0u, P.sset);
        vm_load128(SB.v1, (fl & 2u) ? hash_s16_set(SB.key1, P.s_log2) * 4 : 0u, P.sset);
        SB.t = WB.t;
        WB.t = PIPE_NONE;
        prefetch(pf, next_tile(t, 2));
        t = next_tile(t, 1);
    };
    auto done = [&]() { return t >= n_tiles && W0.t == PIPE_NONE && W1.t == PIPE_NONE && S0.t == PIPE_NONE && S1.t == PIPE_NONE; };
    for (;;) {
        //   prefetch regs, A fills, B reads, B fills, C reads
        if (done()) break;
        step(pfA, W0, W1, S0, S1);
        if (done()) break;
        step(pfB, W1, W0, S1, S0);
    }
    vm_wait<0>();
    if (obuf_n) {
        uint32_t gb = 0;
        if (lane == 0) gb = atomicAdd(P.n_cand, obuf_n);
        gb = __shfl(gb, 0);
        for (uint32_t i = lane; i < obuf_n; i += 64) P.cand[gb + i] = obuf[i];
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
    // window gate: the exact canonical-16-mer set and the filter's probe geometry (sset null = gate off)
    const uint32_t* sset;
    uint32_t s_log2, stride, np, first;   // probed 16-mers: read offsets first + j * stride, j < np
    uint32_t batch;          // candidates per wave and pass (<= 64)
    // seed-and-extend kernel: occurrence lists and packed flanks (index.hip)
    const uint32_t* sval;
    const uint32_t* occ;
    const uint32_t* fpk;
    const uint32_t* foff;
    uint32_t* overflow;      // counter: candidates whose (position, gap) list exceeded list_cap
    uint32_t* overflow_list; // their read ids (re-verified by a second launch with a large list), or null
    uint32_t vlist;          // seed-and-extend kernel: gap entries per candidate (VEXT_LIST / VEXT_LIST_BIG)
    uint32_t n_occ;          // entries of the occurrence lists in all
    gf_hit* stage;           // seed-and-extend kernel: VEXT_STAGE hits per workgroup, collected before they join the hit list
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

// read r of the packed array, re-aligned to a word boundary, into row[0 .. rw) (zero behind its rb bytes).  The words are fetched
// sixteen at a time before any is used: a lane's fetches are independent, one round of global latency per sixteen words (fetched and
// stored one by one, every word of a candidate cost the wave a round trip: 60 us per 64 candidates at C4)
__device__ __forceinline__ void stage_read(const VerifyParams& P, uint32_t* row, uint32_t rw, uint32_t r, bool active) {
    const uint64_t o = (uint64_t)r * P.rb;
    const uint64_t w0 = o >> 2;
    const uint32_t sh = (uint32_t)(o & 3) * 8;
    const uint32_t nw = (P.rb + 3) / 4;
    for (uint32_t i0 = 0; i0 < rw; i0 += 16) {
        uint32_t x[17];
#pragma unroll
        for (uint32_t t = 0; t < 17; ++t) x[t] = (active && i0 + t <= nw) ? packed_word(P, w0 + i0 + t) : 0u;
#pragma unroll
        for (uint32_t t = 0; t < 16; ++t) {
            const uint32_t i = i0 + t;
            if (i >= rw) break;
            uint32_t v = sh ? (x[t] >> sh) | (x[t + 1] << (32 - sh)) : x[t];
            if (i >= nw) v = 0;
            else if (i == nw - 1 && (P.rb & 3)) v &= (1u << ((P.rb & 3) * 8)) - 1;   // drop the next read's bytes
            row[i] = v;
        }
    }
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
    const bool gate = P.sset != nullptr && P.np >= 1 && P.np <= 32;

    // 64 candidates per wave and pass.  (Measured: smaller batches on more concurrent waves are SLOWER — the pass is bound by
    // random 16-B table loads served from the Infinity Cache, not by wave count.)
    const uint32_t bsz = P.batch;
    for (uint32_t c0 = blockIdx.x * bsz; c0 < n_cand; c0 += gridDim.x * bsz) {
        const uint32_t nb = n_cand - c0 < bsz ? n_cand - c0 : bsz;
        const uint32_t my_r = lane < nb ? P.cand[c0 + lane] : 0;
        stage_read(P, sm + lane * rw, rw, my_r, lane < nb);   // my candidate's read
        __syncthreads();
        // Window gate.  A k-mer of the read can only equal a flank k-mer if the ONE probed 16-mer it contains
        // (offset first + q * stride, q = ceil((p - first) / stride) or 0; stride = k - 15) is a flank 16-mer.  Every lane looks its own
        // candidate's np aligned 16-mers up in the exact set (three lookups in flight, four slots per request), so the
        // table below is only consulted around real 16-mer hits: a chance candidate costs ~k-15 table reads, not L-k+1.
        uint32_t my_gate = 0xFFFFFFFFu;
        if (gate) {
            my_gate = 0;
            if (lane < nb) {
                const uint32_t* row = sm + lane * rw;
                for (uint32_t q0 = 0; q0 < P.np; q0 += 3) {
                    uint32_t key[3];
                    Slots4 v[3];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const uint32_t q = q0 + u < P.np ? q0 + u : P.np - 1;
                        key[u] = canon16(stream32(row, 2 * (P.first + q * P.stride)));
                        v[u] = *reinterpret_cast<const Slots4*>(P.sset + hash_s16_set(key[u], (int)P.s_log2));
                    }
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        if (q0 + u >= P.np) continue;
                        bool hit = v[u].x == key[u] || v[u].y == key[u] || v[u].z == key[u] || v[u].w == key[u];
                        const bool open = v[u].x == EMPTY32 || v[u].y == EMPTY32 || v[u].z == EMPTY32 || v[u].w == EMPTY32;
                        if (!hit && !open) {   // rare: four foreign keys in a row
                            uint32_t sl = hash_s16_set(key[u], (int)P.s_log2) + 4;
                            for (;;) {
                                const uint32_t x = P.sset[sl & ((1u << P.s_log2) - 1)];
                                if (x == key[u]) { hit = true; break; }
                                if (x == EMPTY32) break;
                                ++sl;
                            }
                        }
                        my_gate |= (uint32_t)hit << (q0 + u);
                    }
                }
            }
        }
        for (uint32_t j = 0; j < nb; ++j) {
            const uint32_t r = __shfl(my_r, j);
            const uint32_t gate_j = __shfl(my_gate, j);
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
                    if (gate && act[u]) act[u] = (gate_j >> (p <= P.first ? 0u : (p - P.first + P.stride - 1) / P.stride)) & 1u;
                    if (act[u] && P.nmask) {  // any N inside [p, p+k) ?
                        for (uint32_t q = p; q < p + P.k; ++q)
                            if ((P.nmask[(uint64_t)r * P.nmw + (q >> 5)] >> (q & 31)) & 1u) { act[u] = false; break; }
                    }
                    cn[u] = K128{0, 0};
                    slot[u] = 0;
                    if (act[u]) {
                        if (WIDE) cn[u] = canonical(stream_kmer(rwp, 2 * p, (int)P.k), (int)P.k);
                        else cn[u] = K128{canonical64(stream_kmer64(rwp, 2 * p, (int)P.k), (int)P.k), 0};   // k <= 32: one word
                        slot[u] = hash_kmer(cn[u], (int)P.t_log2);
                    }
                }
                // wave-uniform probe steps; finished lanes idle.  Two consecutive slots per step and position: the common chain
                // (one matching entry, then the EMPTY terminator) ends in ONE round trip — the kernel walks its 64 candidates
                // one after the other, so round trips per candidate set its pace
                while (__any(act[0] || act[1])) {
                    uint4 a[2][2], b[2][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            a[u][d] = make_uint4(0, 0, EMPTY32, 0);
                            b[u][d] = make_uint4(EMPTY32, 0, 0, 0);
                            if (act[u]) {
                                const uint64_t sl = (slot[u] + d) & tmask;
                                if (WIDE) { a[u][d] = P.table[2 * sl]; b[u][d] = P.table[2 * sl + 1]; }
                                else a[u][d] = P.table[sl];
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            const uint32_t g = WIDE ? b[u][d].x : a[u][d].z;
                            bool eq = false;
                            if (act[u]) {
                                if (g == EMPTY32) act[u] = false;
                                else {
                                    eq = (((uint64_t)a[u][d].y << 32) | a[u][d].x) == cn[u].hi;
                                    if (WIDE) eq = eq && (((uint64_t)a[u][d].w << 32) | a[u][d].z) == cn[u].lo;
                                }
                            }
                            const unsigned long long bal = __ballot(eq);
                            if (bal) {
                                const uint32_t o = n + __popcll(bal & ((1ull << lane) - 1));
                                if (eq && o < P.list_cap) list[o] = g;
                                n += (uint32_t)__popcll(bal);
                            }
                        }
                        slot[u] = (slot[u] + 2) & tmask;
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

// ---- seed-and-extend verification (min_hits == 1, no repeat mask) -----------------------------------------------------
// A read k-mer at offset p equals a flank k-mer (either strand) iff the ONE probed 16-mer inside it (read offset
// first + q * stride) equals the 16-mer at the corresponding flank position AND the exact match extends from that
// seed far enough to cover [p, p + k) inside the read, the flank's ACGT run and no read N.  So instead of hashing every
// k-mer of a candidate into the 16-B/slot k-mer table (tens of MB: every lookup a fabric request), each aligned 16-mer that
// is a flank 16-mer (exact set, 4 slots per request) is looked up in its occurrence list and the match is extended along
// the diagonal by XOR of 16-base words against the 2-bit packed flanks (0.15 MB at C2: L2/L1 resident):
//   hit(gap)  <=>  some seed/occurrence of that gap has  left_ext + 16 + right_ext >= k,
// extensions capped by k - 16, the read ends, the nearest read N, and the flank's room inside its ACGT run.
// Palindromic 16-mers are tried on both strands.  A candidate lists its gaps in LDS: VEXT_LIST entries in the first pass (64
// candidates per wave); the reads that hit more gaps than that — reads inside a repeat shared by many flanks — go through the
// overflow list to a second launch of this kernel with VEXT_LIST_BIG entries and a few candidates per wave, and only what outgrows
// that as well to the table kernel.  (Round 3 sent every overflow straight to the table kernel: on the planted-repeat workload,
// where 0.5 M reads hit 30-50 gaps each, that pass took 460 ms of a 475-ms step.)
constexpr uint32_t VEXT_STAGE = 1024;    // hits a wave collects in its slice of the staging buffer before one atomic appends them to the list
constexpr uint32_t VEXT_WALK_MAX = 96;   // occurrences one lane of the first pass walks for a (read, seed) before it hands the read to the long-list pass
constexpr uint32_t VEXT_LIST = 16, VEXT_LIST_BIG = 256, VEXT_BATCH_BIG = 8, VEXT_LIST_HUGE = 2048, VEXT_BATCH_HUGE = 2;   // (8 x 256 slots = 8 KiB per wave: a dozen waves per CU; the long list is a hash SET of gaps, full at 192)


__device__ __forceinline__ uint32_t fl32(const uint32_t* words, uint32_t base) {   // 16 bases from base offset `base`, MSB-first words
    const uint32_t d = base >> 4, sh = 2 * (base & 15);
    const uint64_t v = ((uint64_t)words[d] << 32) | words[d + 1];
    return (uint32_t)((v << sh) >> 32);
}

// 64 mask bits starting at bit `start` (may be negative or run past the row: those bits read 0)
__device__ __forceinline__ uint64_t nbits64(const uint32_t* m, int nmw, int start) {
    const int w0 = start >> 5;     // arithmetic shift: floor
    const uint32_t sh = (uint32_t)start & 31;
    uint32_t x[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int idx = w0 + t;
        x[t] = (idx >= 0 && idx < nmw) ? m[idx] : 0u;
    }
    const uint64_t lo = ((uint64_t)x[1] << 32) | x[0];
    return sh ? (lo >> sh) | ((uint64_t)x[2] << (64 - sh)) : lo;
}

__device__ __forceinline__ bool ext_hit(const uint32_t* row, uint32_t qs, const uint32_t* fw, uint32_t f, bool same, uint32_t capL,
                                        uint32_t capR, uint32_t k) {
    // row / fw point at base 0 of the read / flank; both have >= 4 readable words in front and zero padding behind
    uint32_t lext = 0, rext = 0;
    for (uint32_t c = 0; 16 * c < capL; ++c) {
        const uint32_t R = stream32(row - 4, 128 + 2 * (qs - 16 * (c + 1)));
        const uint32_t F = same ? fl32(fw - 4, 64 + f - 16 * (c + 1)) : revpairs32(~fl32(fw - 4, 64 + f + 16 + 16 * c));
        const uint32_t X = R ^ F;
        if (X == 0) { lext += 16; continue; }
        lext += (uint32_t)__builtin_ctz(X) >> 1;
        break;
    }
    lext = lext < capL ? lext : capL;
    for (uint32_t c = 0; 16 * c < capR; ++c) {
        const uint32_t R = stream32(row - 4, 128 + 2 * (qs + 16 + 16 * c));
        const uint32_t F = same ? fl32(fw - 4, 64 + f + 16 + 16 * c) : revpairs32(~fl32(fw - 4, 64 + f - 16 * (c + 1)));
        const uint32_t X = R ^ F;
        if (X == 0) { rext += 16; continue; }
        rext += (uint32_t)__builtin_clz(X) >> 1;
        break;
    }
    rext = rext < capR ? rext : capR;
    return lext + rext + 16 >= k;
}

__global__ __launch_bounds__(64) void screen_verify_ext_kernel(VerifyParams P) {
    extern __shared__ uint32_t sm[];   // rows [64][rwp] | nmask [64][nmw] | slots [64][np] | cnt [64] | lists [batch][vlist]
    constexpr uint32_t OBUF = 128;
    __shared__ gf_hit obuf[OBUF];
    __shared__ uint32_t obuf_n;
    const uint32_t lane = threadIdx.x;
    if (lane == 0) obuf_n = 0;
    // The hit list has ONE counter, and returning atomics on one address are served at 11-15 ns each: the LDS buffer (>= 64 hits)
    // empties into the wave's slice of a global staging buffer, and the slice joins the list a thousand hits at a time (C4: 3.8 M
    // hits per step were 47 000 atomics — half of the kernel's time on that counter's queue).
    gf_hit* const stage = P.stage + (size_t)blockIdx.x * VEXT_STAGE;
    uint32_t stage_n = 0;   // wave-uniform
    auto flush_stage = [&]() {
        uint32_t gb = 0;
        if (lane == 0) gb = atomicAdd(P.n_out, stage_n);
        gb = __shfl(gb, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wave's own stores are in L2; read them back from there (not from L1)
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(stage);
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(P.out);
        static_assert(sizeof(gf_hit) == 8, "hits move as 64-bit words");
        for (uint32_t q = lane; q < stage_n; q += 64)
            if (gb + q < P.cap) dst[gb + q] = __hip_atomic_load(src + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stage_n = 0;
    };
    auto put_out = [&](uint32_t nn) {   // obuf[0, nn) -> staging slice
        if (stage_n + nn > VEXT_STAGE) flush_stage();
        for (uint32_t q = lane; q < nn; q += 64) stage[stage_n + q] = obuf[q];
        stage_n += nn;
    };
    const uint32_t rw = (P.rb + 24) / 4 + 1, rwp = rw + 4;
    const uint32_t nmw = P.nmask ? P.nmw : 0;
    uint32_t* rows = sm;
    uint32_t* nmr = rows + 64 * rwp;
    uint32_t* slots = nmr + 64 * nmw;
    uint32_t* cnt = slots + 64 * P.np;
    uint32_t* lists = cnt + 64;
    const uint32_t VL = P.vlist;
    const bool SET = VL > VEXT_LIST;   // (then a power of two)
    const uint32_t n_cand = *P.n_cand;
    const uint32_t W = P.k - 16;
    const uint32_t bsz = P.batch;
    __syncthreads();
    for (uint32_t c0 = blockIdx.x * bsz; c0 < n_cand; c0 += gridDim.x * bsz) {
        const uint32_t nb = n_cand - c0 < bsz ? n_cand - c0 : bsz;
        const uint32_t my_r = lane < nb ? P.cand[c0 + lane] : 0;
        {   // stage my candidate's read, re-aligned to a word boundary, behind 4 zero words
            uint32_t* row = rows + lane * rwp;
            row[0] = row[1] = row[2] = row[3] = 0;
            stage_read(P, row + 4, rw, my_r, lane < nb);
            for (uint32_t i = 0; i < nmw; ++i) nmr[lane * nmw + i] = lane < nb ? P.nmask[(uint64_t)my_r * P.nmw + i] : 0;
            cnt[lane] = 0;
            if (SET) for (uint32_t i = lane; i < bsz * VL; i += 64) lists[i] = EMPTY32;
        }
        __syncthreads();
        // exact-set lookup of every aligned 16-mer of my candidate: slot of the match, or EMPTY32
        if (lane < nb) {
            const uint32_t* row = rows + lane * rwp + 4;
            for (uint32_t q0 = 0; q0 < P.np; q0 += 3) {
                uint32_t key[3], h[3];
                Slots4 v[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const uint32_t q = q0 + u < P.np ? q0 + u : P.np - 1;
                    key[u] = canon16(stream32(row, 2 * (P.first + q * P.stride)));
                    h[u] = hash_s16_set(key[u], (int)P.s_log2);
                    v[u] = *reinterpret_cast<const Slots4*>(P.sset + h[u]);
                }
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    if (q0 + u >= P.np) continue;
                    uint32_t sl = EMPTY32;
                    if (v[u].x == key[u]) sl = h[u];
                    else if (v[u].y == key[u]) sl = h[u] + 1;
                    else if (v[u].z == key[u]) sl = h[u] + 2;
                    else if (v[u].w == key[u]) sl = h[u] + 3;
                    else if (v[u].x != EMPTY32 && v[u].y != EMPTY32 && v[u].z != EMPTY32 && v[u].w != EMPTY32) {
                        uint32_t s2 = h[u] + 4;   // rare: four foreign keys in a row
                        for (;;) {
                            const uint32_t x = P.sset[s2 & ((1u << P.s_log2) - 1)];
                            if (x == key[u]) { sl = s2; break; }
                            if (x == EMPTY32) break;
                            ++s2;
                        }
                    }
                    slots[lane * P.np + q0 + u] = sl == EMPTY32 ? EMPTY32 : (sl & ((1u << P.s_log2) - 1));
                }
            }
        }
        __syncthreads();
        // one work item per (candidate, aligned 16-mer)
        const uint32_t n_items = nb * P.np;
        // what the walks below need of an item: the seed's place in the read and how far an extension may run (wave-uniform in the long-list pass)
        struct Item { const uint32_t* row; uint32_t j, qs; bool ro, pal, ok; uint32_t baseL, baseR; };
        auto item_of = [&](uint32_t i) -> Item {
            Item it;
            it.j = i / P.np;
            const uint32_t q = i - it.j * P.np;
            it.row = rows + it.j * rwp + 4;
            it.qs = P.first + q * P.stride;
            const uint32_t w16 = stream32(it.row, 2 * it.qs);
            const uint32_t key = canon16(w16);
            it.ro = key != w16;
            it.pal = revpairs32(~key) == key;
            it.ok = true;
            uint32_t nl = 64, nr = 64;
            if (nmw) {
                const uint32_t* m = nmr + it.j * nmw;
                if (nbits64(m, (int)nmw, (int)it.qs) & 0xFFFFull) it.ok = false;      // an N inside the seed: no k-mer through it counts
                const uint64_t lb = nbits64(m, (int)nmw, (int)it.qs - 64), rbits = nbits64(m, (int)nmw, (int)it.qs + 16);
                nl = lb ? (uint32_t)__builtin_clzll(lb) : 64;
                nr = rbits ? (uint32_t)__builtin_ctzll(rbits) : 64;
            }
            it.baseL = W < it.qs ? W : it.qs;
            it.baseR = P.read_len - (it.qs + 16);
            it.baseL = it.baseL < nl ? it.baseL : nl;
            it.baseR = it.baseR < W ? it.baseR : W;
            it.baseR = it.baseR < nr ? it.baseR : nr;
            return it;
        };
        // one occurrence {flank, info} of the item's 16-mer: does a k-mer of the read through the seed equal the flank's there?
        auto occ_hits = [&](const Item& it, uint32_t fid, uint32_t info) -> bool {
            const uint32_t f = info & 0xFFFFu, lroom = (info >> 18) & 63u, rroom = (info >> 24) & 63u;
            const bool fo = (info >> 16) & 1u;
            const uint32_t* fw = P.fpk + P.foff[fid];
            bool hit = false;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bool same = t == 0 ? (it.ro == fo) : (it.ro != fo);
                if (t == 1 && !it.pal) break;
                const uint32_t capL = it.baseL < (same ? lroom : rroom) ? it.baseL : (same ? lroom : rroom);
                const uint32_t capR = it.baseR < (same ? rroom : lroom) ? it.baseR : (same ? rroom : lroom);
                if (capL + capR + 16 < P.k) continue;
                hit = hit || ext_hit(it.row, it.qs, fw, f, same, capL, capR, P.k);
            }
            return hit;
        };
        if (SET) {
            // long-list pass: the reads here sit in repeats, their seeds' occurrence lists hold tens to hundreds of entries — ONE item at a
            // time, its occurrences spread over the lanes (a lane per (read, seed) walked such a list alone: 18 ms on the stress bench).  The
            // gap list is a hash set: one CAS claims a slot, so lanes cannot list a gap twice.
            for (uint32_t i = 0; i < n_items; ++i) {
                const uint32_t slot = slots[i];
                if (slot == EMPTY32) continue;
                const Item it = item_of(i);
                if (!it.ok) continue;
                uint32_t* lj = lists + it.j * VL;
                for (uint32_t oi0 = P.sval[slot];; oi0 += 64) {
                    const uint32_t oi = oi0 + lane;
                    const bool in = oi < P.n_occ;
                    const uint32_t fid = in ? P.occ[2 * (size_t)oi] : 0u, info = in ? P.occ[2 * (size_t)oi + 1] : (1u << 17);
                    const unsigned long long lastb = __ballot((info >> 17) & 1u);
                    const uint32_t n_here = lastb ? (uint32_t)__ffsll((long long)lastb) : 64u;      // lanes below belong to this 16-mer's list
                    if (lane < n_here) {
                        // a gap that is listed already needs no second proof: a low-complexity 16-mer stands at fifty offsets of the same
                        // flank, and the extension is a hundred instructions and four loads, the set look-up five
                        const uint32_t g = fid >> 1;
                        uint32_t hs = (g * 0x9E3779B1u) & (VL - 1);
                        bool listed = false;
                        for (uint32_t pr = 0; pr < VL; ++pr) {
                            const uint32_t x = lj[hs];
                            if (x == g) { listed = true; break; }
                            if (x == EMPTY32) break;
                            hs = (hs + 1) & (VL - 1);
                        }
                        if (!listed && occ_hits(it, fid, info)) {
                            for (uint32_t pr = 0;; ++pr) {      // (from the first free or foreign slot the look-up stopped at)
                                if (pr == VL) { cnt[it.j] = VL + 1; break; }
                                const uint32_t old = atomicCAS(&lj[hs], EMPTY32, g);
                                if (old == EMPTY32) { atomicAdd(&cnt[it.j], 1u); break; }
                                if (old == g) break;
                                hs = (hs + 1) & (VL - 1);
                            }
                        }
                    }
                    if (lastb) break;
                }
            }
        } else
        for (uint32_t i = lane; i < n_items; i += 64) {
            const uint32_t slot = slots[i];
            if (slot == EMPTY32) continue;
            const Item it = item_of(i);
            if (!it.ok) continue;
            const uint32_t j = it.j;
            uint32_t oi = P.sval[slot];
            for (uint32_t steps = 0;; ++steps) {
                if (steps == VEXT_WALK_MAX) { cnt[j] = VL + 1; break; }      // a long occurrence list (a 16-mer that stands in hundreds of flanks): the long-list pass walks it 64 entries at a time
                const uint32_t fid = P.occ[2 * (size_t)oi], info = P.occ[2 * (size_t)oi + 1];
                {
                    const uint32_t g = fid >> 1;
                    uint32_t* lj = lists + j * VL;
                    const uint32_t have = cnt[j] < VL ? cnt[j] : VL;
                    bool dup = false;
                    for (uint32_t e = 0; e < have; ++e) dup = dup || lj[e] == g;
                    if (!dup && occ_hits(it, fid, info)) {      // (a gap that is listed already needs no second proof)
                        const uint32_t e = atomicAdd(&cnt[j], 1u);
                        if (e < VL) lj[e] = g;
                    }
                }
                if (((info >> 17) & 1u) || cnt[j] > VL) break;      // (a list that has run over: the read goes to the long-list pass as a whole)
                ++oi;
            }
        }
        __syncthreads();
        // per candidate: distinct gaps -> hits (a list that ran over goes to the table kernel)
        {
            const uint32_t n = lane < nb ? cnt[lane] : 0;
            const bool over = SET ? n > VL - VL / 4 : n > VL;
            const unsigned long long ob = __ballot(over);
            if (ob) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(P.overflow, (uint32_t)__popcll(ob));
                base = __shfl(base, 0);
                if (over && P.overflow_list) P.overflow_list[base + __popcll(ob & ((1ull << lane) - 1))] = my_r;
            }
            const uint32_t* lj = lists + (lane < nb ? lane : 0) * VL;
            for (uint32_t d = 0; d < VL; ++d) {
                bool emit = !over && (SET ? n > 0 : d < n);
                uint32_t g = 0;
                if (emit) {
                    g = lj[d];
                    if (SET) emit = g != EMPTY32;
                    else for (uint32_t e = 0; e < d; ++e) emit = emit && lj[e] != g;
                }
                const unsigned long long bal = __ballot(emit);
                if (!bal) { if (!SET && !__any(d + 1 < n && !over)) break; else continue; }
                const uint32_t base = obuf_n;
                if (emit) obuf[base + __popcll(bal & ((1ull << lane) - 1))] = gf_hit{g, my_r};
                __syncthreads();
                if (lane == 0) obuf_n = base + (uint32_t)__popcll(bal);
                __syncthreads();
                if (obuf_n >= OBUF - 64) {
                    put_out(obuf_n);
                    __syncthreads();
                    if (lane == 0) obuf_n = 0;
                    __syncthreads();
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (obuf_n) put_out(obuf_n);
    if (stage_n) flush_stage();
}

int launch_screen(gf_ctx* ctx, const FlankIndex& ix, const void* d_reads, const void* d_nmask, size_t n_reads,
                  int read_len, int min_hits, void* d_out, size_t cap, void* d_n_out) {
    // one-shot (gf_stream_wait_after_filter): taken here, so that no exit below leaves it armed for a later, unrelated pass
    gf_ctx* const waiter = ctx->after_filter;
    ctx->after_filter = nullptr;
    if (read_len < ix.k || read_len > 1000) return GF_E_INVAL;
    if (n_reads >= 0xFFFFFFFFull || cap > 0xFFFFFFFFull) return GF_E_INVAL;
    const uint32_t rb = (uint32_t)((read_len + 3) / 4);
    if (rb > 250) return GF_E_UNSUPPORTED;
    int rc;
    if ((rc = ensure(ctx, ctx->cand, std::max<size_t>(n_reads, 1) * 4))) return rc;
    if ((rc = ensure(ctx, ctx->counters, GF_COUNTER_BYTES))) return rc;
    uint32_t* d_cnt = (uint32_t*)ctx->counters.p;  // [0] n_cand, [1] overflow
    zero_regions(ctx, ZeroList{{d_cnt, (uint32_t*)d_n_out, nullptr, nullptr}, {5, 1, 0, 0}});  // [0] n_cand [1] error overflow [2] [3] [4] the overflow lists of the verification passes
    if (n_reads == 0) return GF_OK;

    FilterParams F;
    F.reads = (const uint8_t*)d_reads;
    F.n_reads = n_reads;
    F.rb = rb;
    // Probed 16-mers of a read: offsets first + j * stride, stride = k - 15.  A k-mer at offset p in [0, L - k] covers the 16-mer
    // offsets [p, p + stride - 1], so the probes must start at first <= k - 16 and reach L - k: np = floor((L - k) / stride) + 1 of
    // them do — one fewer than probing from offset 0 to the end of the read whenever (L - 16) mod stride < k - 16 (150-base reads:
    // k = 51: 3 instead of 4, k = 41: 5 instead of 6, k = 31: 8 instead of 9).  first = the byte-aligned offset closest below k - 16
    // that still reaches (the pipelined kernel fetches byte-aligned probes faster).
    // The 256-bucket filter checks `ext` more bases behind every seed (Part4Params::ext): a k-mer must then contain the 16-mer AND
    // those bases, so the stride is k - 15 - ext and first <= k - 16 - ext; ext = the most (<= 2) that leaves np as it is
    // (150-base reads: k = 51: 2, stride 34; k = 41: 2; k = 31: 1).
    const bool use_pf4 = (ctx->screen_variant == 16 || ctx->screen_variant == 17 || (ctx->screen_variant == 0 && n_reads >= (1u << 20))) && ix.bm_log2 >= 27 &&
                         ix.bm_log2 <= 28 && rb <= 64 && ix.d_sgrp &&
                         ((size_t)PF2_WAVES * PF2_TILES * (((size_t)64 * rb + 16 + 7) / 8 * 2) * 4 + (size_t)PF2_BATCH * 5 + (size_t)PF2_NB * (PF4_STAGE + 1) * 4 +
                          (6 * PF2_NB + 8) * 4) <= 156 * 1024;
    int ext = 0;
    if (use_pf4 && ctx->screen_ext)
        for (int e = 2; e >= 1 && !ext; --e)
            if (ix.stride - e >= 1 && (read_len - ix.k) / (ix.stride - e) == (read_len - ix.k) / ix.stride) ext = e;
    const int stride_probe = ix.stride - ext;
    const uint32_t np_probe = (uint32_t)((read_len - ix.k) / stride_probe + 1);
    uint32_t first_probe = (uint32_t)(ix.k - 16 - ext);
    {
        const int lo = (read_len - ix.k) - (int)(np_probe - 1) * stride_probe;
        const uint32_t al = first_probe & ~3u;
        if ((int)al >= lo) first_probe = al;
    }
    F.stride2 = 2 * (uint32_t)stride_probe;
    F.first2 = 2 * first_probe;
    F.np = np_probe;
    F.bitmap = ix.d_bitmap;
    F.sset = ix.d_sset;
    F.bm_log2 = ix.bm_log2;
    F.s_log2 = ix.s_log2;
    F.cand = (uint32_t*)ctx->cand.p;
    F.n_cand = d_cnt;
    F.bitmap_lds = ix.d_bitmap_lds;
    F.lds_log2 = ix.lds_log2;
    F.stream_policy = (uint32_t)ctx->screen_stream_policy;
    F.bitmap_mid = ix.mid_log2 ? ix.d_bitmap_mid : nullptr;
    F.mid_log2 = (uint32_t)ix.mid_log2;
    // Kernel choice (screen_variant: 0 = automatic; 9 / 13 / 14 / 16 force the plain / pipelined / 16-bucket / 256-bucket form — the
    // parity tests run every one of them on the same inputs):
    // the LDS pre-filter pays while the coarse bitmap is sparse enough to stop most probes before L2 and a handful of waves
    // fit next to it (long reads leave too few); the pipelined form covers up to 10 probes and 64 packed bytes per read
    const size_t w_bm_bytes = ix.d_bitmap_lds ? ((size_t)1 << ix.lds_log2) / 8 : 0;
    const size_t w_slice_words = ((size_t)64 * rb + 16 + 15) / 16 * 4;
    const size_t w_per_wave = (w_slice_words + WOBUF) * 4;
    const size_t w_nw = std::min<size_t>(16, (160 * 1024 - 512 - w_bm_bytes) / w_per_wave);
    const size_t tiles64 = (n_reads + 63) / 64;
    const int nch = (rb + 15) / 16 <= 4 ? (int)((rb + 15) / 16) : 0;
    const bool lds_auto = ix.d_bitmap_lds && ix.lds_fill <= 0.6 && ctx->screen_variant == 0 && w_nw >= 6;
    const bool lds_forced = ctx->screen_variant == 13 && ix.d_bitmap_lds && w_nw >= 2;
    const bool pipe_ok = F.np >= 1 && F.np <= 10 && ix.lds_log2 >= 7 && nch >= 1;
    if ((lds_auto || lds_forced) && pipe_ok) {
        // software-pipelined wave kernel (three tiles in flight per wave)
        const size_t nw = std::min<size_t>(w_nw, 8);   // measured: 8 waves x 256 VGPRs beat 11 x 168
        const int npt = std::max(5, (int)F.np);        // probes per read, unrolled
        void (*wk)(FilterParams, uint32_t) = nullptr;
#define GF_PK(N, Q) if (nch == N && npt == Q) wk = (int)F.np == Q ? screen_filter_pipe_kernel<N, Q, true> : screen_filter_pipe_kernel<N, Q, false>;
#define GF_PKX(N, Q) if (nch == N && npt == Q) wk = screen_filter_pipe_kernel<N, Q, true>;
#define GF_PKN(Q) GF_PKX(1, Q) GF_PKX(2, Q) GF_PKX(3, Q) GF_PKX(4, Q)
        GF_PK(1, 5) GF_PK(2, 5) GF_PK(3, 5) GF_PK(4, 5)
        GF_PKN(6) GF_PKN(7) GF_PKN(8) GF_PKN(9) GF_PKN(10)
#undef GF_PKX
#undef GF_PKN
#undef GF_PK
        ctx->screen_kernels = "screen_filter_pipe_kernel<" + std::to_string(nch) + ", " + std::to_string(npt) + ", " + ((int)F.np == npt || npt > 5 ? "true" : "false") + ">";
        LaunchTimer tm(ctx, GF_KERNEL_SCREEN);
        hipLaunchKernelGGL(wk, dim3((unsigned)std::min<size_t>((tiles64 + nw - 1) / nw, ctx->n_cu)), dim3((unsigned)(nw * 64)),
                           w_bm_bytes + nw * w_per_wave, ctx->stream, F, (uint32_t)w_slice_words);
    } else if (use_pf4) {
        // partitioned filter, 256 buckets, 4-byte pairs (see Part4Params)
        Part4Params Q;
        Q.F = F;
        Q.sgrp = ix.d_sgrp;
        Q.ext = (uint32_t)ext;
        const size_t slice_words = ((size_t)64 * rb + 16 + 7) / 8 * 2;
        const size_t tiles64 = (n_reads + 63) / 64;
        const size_t tiles_wg = (size_t)PF2_WAVES * PF2_TILES;      // tiles per workgroup and iteration
        const size_t lds_a = tiles_wg * slice_words * 4 + (size_t)PF2_BATCH * 5 + (size_t)PF2_NB * (PF4_STAGE + 1) * 4 + (6 * PF2_NB + 8) * 4;
        Q.n_writers = (uint32_t)std::min<size_t>(std::min<size_t>((tiles64 + tiles_wg - 1) / tiles_wg, (size_t)ctx->n_cu), 256);   // the pair list carries the writer in 8 bits
        Q.tiles_wg = (uint32_t)tiles_wg;
        const size_t n_iter = (tiles64 + (size_t)Q.n_writers * tiles_wg - 1) / ((size_t)Q.n_writers * tiles_wg);
        const double pairs_w = (double)n_iter * tiles_wg * 64.0 * F.np;
        const double expect = pairs_w / PF2_NB;
        Q.cap = ((uint32_t)(expect * 1.05 + 6.0 * std::sqrt(expect + 1.0) + 128.0) + 63u) & ~63u;
        // whole-line stores (pf4_scatter_lines_kernel) where all probes of a read make one group, the open lines fit beside the tiles
        // and a line's index in `pairs` fits 32 bits (screen_variant 17: the unaligned form all the same)
        const uint32_t grp = F.np < PF2_GROUP ? (F.np ? F.np : 1u) : PF2_GROUP;   // probes per sorted group
        Q.n_grp = (F.np + grp - 1) / grp;
        const bool lines = Q.n_grp <= 2 && pf4_lines_lds_bytes(slice_words, grp) <= 160 * 1024 && ctx->screen_variant != 17 &&
                           (uint64_t)PF2_NB * Q.n_writers * (Q.cap >> 5) + 1 < 0xFFFFFFFFull;
        Q.n_groups = (uint32_t)(n_iter * Q.n_grp);
        Q.gs = (Q.n_groups + 1 + 15) & ~15u;
        Q.cap8 = (uint32_t)(std::min<size_t>(std::max<size_t>((size_t)1 << 22, n_reads / 2), 0x7FFFFFFFu) / PF4_CHUNK * PF4_CHUNK);
        if (ctx->screen_pf4_cap8 > 0) Q.cap8 = (uint32_t)std::max(1, ctx->screen_pf4_cap8 / (int)PF4_CHUNK) * PF4_CHUNK;   // tests: a short pair list (the serial path)
        const size_t b_pairs = (size_t)PF2_NB * Q.n_writers * Q.cap * 4, b_cnt = ((size_t)PF2_NB * Q.n_writers * 4 + 255 + 256) & ~(size_t)255,
                     b_seen = (((size_t)n_reads + 31) / 32 * 4 + 255) & ~(size_t)255, b_fill = (size_t)PF2_NB * Q.n_writers * Q.gs * 4,
                     b_c8 = ((size_t)Q.cap8 * 12 + Q.cap8 / PF4_CHUNK + 1 + 255) & ~(size_t)255;
        if (Q.cap >= (1u << 24)) return GF_E_INVAL;   // a position must fit 24 bits (2^32 reads stay far below)
        if ((rc = ensure(ctx, ctx->part_ws, b_cnt + b_seen + b_fill + b_c8 + b_pairs + 1024))) return rc;
        uint8_t* ws = (uint8_t*)ctx->part_ws.p;
        Q.count = (uint32_t*)ws;
        Q.n_cand8 = (uint32_t*)(ws + b_cnt - 256);
        Q.seen = (uint32_t*)(ws + b_cnt);
        Q.fills = (uint32_t*)(ws + b_cnt + b_seen);
        Q.cand8 = (unsigned long long*)(ws + b_cnt + b_seen + b_fill);
        Q.cand8x = (uint32_t*)(ws + b_cnt + b_seen + b_fill + (size_t)Q.cap8 * 8);
        Q.chunk_b = (uint8_t*)(ws + b_cnt + b_seen + b_fill + (size_t)Q.cap8 * 12);
        Q.pairs = (uint32_t*)(ws + b_cnt + b_seen + b_fill + b_c8);
        GF_HIP(ctx, hipMemsetAsync(ws + b_cnt - 256, 0, 256 + b_seen, ctx->stream));
        ctx->screen_kernels = std::string(lines ? "pf4_scatter_lines_kernel<" : "pf4_scatter_kernel<") + std::to_string(grp) + "u, " +
                              (((F.first2 & 7u) == 0 && (F.stride2 & 7u) == 0) ? "true" : "false") + (lines ? (Q.n_grp == 2 ? ", 2u" : ", 1u") : "") +
                              ">,pf4_probe_kernel,pf4_resolve_kernel,pf4_list_kernel";
        LaunchTimer tm(ctx, GF_KERNEL_SCREEN);
        if (lines) {
            const bool bytes = (F.first2 & 7u) == 0 && (F.stride2 & 7u) == 0;
            void (*scatter)(Part4Params, uint32_t) =
                Q.n_grp == 2 ? (bytes ? pf4_scatter_lines_kernel<4, true, 2> : pf4_scatter_lines_kernel<4, false, 2>) :      // (five to eight probes per read: grp == 4)
                bytes ? (grp == 1 ? pf4_scatter_lines_kernel<1, true> : grp == 2 ? pf4_scatter_lines_kernel<2, true> : grp == 3 ? pf4_scatter_lines_kernel<3, true> : pf4_scatter_lines_kernel<4, true>)
                      : (grp == 1 ? pf4_scatter_lines_kernel<1, false> : grp == 2 ? pf4_scatter_lines_kernel<2, false> : grp == 3 ? pf4_scatter_lines_kernel<3, false> : pf4_scatter_lines_kernel<4, false>);
            hipLaunchKernelGGL(scatter, dim3(Q.n_writers), dim3(64 * PF2_WAVES), pf4_lines_lds_bytes(slice_words, grp), ctx->stream, Q, (uint32_t)slice_words);
        } else {
            const bool bytes = (F.first2 & 7u) == 0 && (F.stride2 & 7u) == 0;
            void (*scatter)(Part4Params, uint32_t) =
                bytes ? (grp == 1 ? pf4_scatter_kernel<1, true> : grp == 2 ? pf4_scatter_kernel<2, true> : grp == 3 ? pf4_scatter_kernel<3, true> : pf4_scatter_kernel<4, true>)
                      : (grp == 1 ? pf4_scatter_kernel<1, false> : grp == 2 ? pf4_scatter_kernel<2, false> : grp == 3 ? pf4_scatter_kernel<3, false> : pf4_scatter_kernel<4, false>);
            hipLaunchKernelGGL(scatter, dim3(Q.n_writers), dim3(64 * PF2_WAVES), lds_a, ctx->stream, Q, (uint32_t)slice_words);
        }
        const size_t lds_b = (((size_t)1 << (ix.bm_log2 - PF2_NB_LOG2 - 5)) + 16 * (3 * PF4_OBUF + 2 * PF2_PEND)) * 4;
        hipLaunchKernelGGL(pf4_probe_kernel, dim3((unsigned)std::min<size_t>(PF2_NB, (size_t)ctx->n_cu)), dim3(1024), lds_b, ctx->stream, Q);
        const size_t lds_r = (size_t)4 * 8 * (((2 * (size_t)rb + 3) / 4) * 4 + 4) * 4;
        hipLaunchKernelGGL(pf4_resolve_kernel, dim3((unsigned)ctx->n_cu * 8), dim3(256), lds_r, ctx->stream, Q);
        hipLaunchKernelGGL(pf4_list_kernel, dim3((unsigned)std::min<size_t>((size_t)ctx->n_cu * 4, (n_reads + 32 * 256 - 1) / (32 * 256))), dim3(256), 0, ctx->stream, Q);
    } else {
        const size_t n_tiles = (n_reads + TILE_READS - 1) / TILE_READS;
        const unsigned grid = (unsigned)std::min<size_t>(n_tiles, (size_t)ctx->n_cu * 8);
        ctx->screen_kernels = "screen_filter_kernel<9>";
        LaunchTimer tm(ctx, GF_KERNEL_SCREEN);
        hipLaunchKernelGGL(screen_filter_kernel<9>, dim3(grid), dim3(256), TILE_READS * rb + 16, ctx->stream, F);
    }
    GF_HIP(ctx, hipGetLastError());
    if (waiter) {   // gf_stream_wait_after_filter: the peer's stream goes on once the filter pass above has finished
        hipEvent_t ev;
        GF_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ev, ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(waiter->stream, ev, 0);
        (void)hipEventDestroy(ev);
        if (e != hipSuccess) return set_hip_error(ctx, e, "gf_stream_wait_after_filter");
    }

    VerifyParams V;
    V.reads32 = (const uint32_t*)d_reads;
    V.n_words = ((uint64_t)n_reads * rb) / 4;
    V.tail_bytes = (uint32_t)(((uint64_t)n_reads * rb) & 3);
    V.nmask = (const uint32_t*)d_nmask;
    V.rb = rb;
    V.read_len = read_len;
    V.k = ix.k;
    V.nmw = (read_len + 31) / 32;
    V.cand = F.cand;
    V.n_cand = d_cnt;
    V.table = (const uint4*)ix.d_table;
    V.t_log2 = ix.t_log2;
    V.min_hits = min_hits < 1 ? 1 : min_hits;
    V.sset = (ctx->screen_verify_gate || ctx->screen_verify_ext) ? ix.d_sset : nullptr;
    V.s_log2 = ix.s_log2;
    V.stride = (uint32_t)stride_probe;
    V.np = np_probe;
    V.first = first_probe;
    V.batch = (uint32_t)std::min(64, std::max(1, ctx->screen_verify_batch));
    const uint32_t npos = read_len - ix.k + 1;
    V.out = (gf_hit*)d_out;
    V.cap = (uint32_t)cap;
    V.n_out = (uint32_t*)d_n_out;
    V.stage = nullptr;
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
    V.sval = ix.d_sval; V.occ = ix.d_occ; V.fpk = ix.d_fpk; V.foff = ix.d_foff;
    V.n_occ = (uint32_t)ix.n_occ;
    V.vlist = VEXT_LIST;
    const bool use_ext = ctx->screen_verify_ext && V.min_hits == 1 && ix.max_gaps_per_kmer == 0 && ix.ext_ok && V.np >= 1 && V.np <= 32;
    if (use_ext) {
        if ((rc = ensure(ctx, ctx->verify_stage, (size_t)grid2 * VEXT_STAGE * sizeof(gf_hit)))) return rc;
        V.stage = (gf_hit*)ctx->verify_stage.p;
        // seed-and-extend kernel instead of the k-mer table (same hits; see screen_verify_ext_kernel)
        const size_t rwp = (rb + 24) / 4 + 1 + 4, nmw = d_nmask ? V.nmw : 0;
        {
            const size_t lds2 = (64 * (rwp + nmw + V.np + 1) + (size_t)V.batch * V.vlist) * 4;
            LaunchTimer tm(ctx, GF_KERNEL_VERIFY);
            hipLaunchKernelGGL(screen_verify_ext_kernel, dim3(grid2), dim3(64), lds2, ctx->stream, V);
        }
        // pass 2: the reads that hit more than VEXT_LIST gaps (repeats shared by many flanks), a few per wave with a long list each;
        // what outgrows that too is queued for the table kernel (the old candidate list is free by now)
        // ... and a third pass for the reads that hit more gaps than THAT set holds (a homopolymer run shared by hundreds of flanks): two
        // reads per wave, 2 048 slots each.  The lists alternate between the two candidate buffers.
        const uint32_t big_vl[2] = {VEXT_LIST_BIG, VEXT_LIST_HUGE}, big_batch[2] = {VEXT_BATCH_BIG, VEXT_BATCH_HUGE};
        uint32_t* bufs[2] = {(uint32_t*)ctx->cand2.p, (uint32_t*)ctx->cand.p};
        for (int ps = 0; ps < 2; ++ps) {
            V.cand = bufs[ps & 1];
            V.n_cand = d_cnt + 2 + ps;
            V.overflow = d_cnt + 3 + ps;
            V.overflow_list = bufs[(ps + 1) & 1];
            V.vlist = big_vl[ps];
            V.batch = big_batch[ps];
            const size_t lds2 = (64 * (rwp + nmw + V.np + 1) + (size_t)V.batch * V.vlist) * 4;
            LaunchTimer tm(ctx, GF_KERNEL_VERIFY);
            hipLaunchKernelGGL(screen_verify_ext_kernel, dim3(grid2), dim3(64), lds2, ctx->stream, V);
        }
        V.cand = bufs[0];
        V.n_cand = d_cnt + 4;
        V.batch = (uint32_t)std::min(64, std::max(1, ctx->screen_verify_batch));
    } else {
        launch_verify(V);
        V.cand = (const uint32_t*)ctx->cand2.p;
        V.n_cand = d_cnt + 2;
    }
    GF_HIP(ctx, hipGetLastError());
    // last pass: the queued reads through the k-mer table with a list as large as LDS allows; overflowing that is an error (d_cnt[1])
    size_t want = ix.max_gaps_per_kmer ? (size_t)npos * ix.max_gaps_per_kmer : 15000;
    V.list_cap = (uint32_t)std::min<size_t>(std::max<size_t>(want, 1024), 15000);
    V.overflow = d_cnt + 1;
    V.overflow_list = nullptr;
    launch_verify(V);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

}  // namespace gf
