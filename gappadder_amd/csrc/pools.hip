// pools.hip — per-gap read pools on the device.
//
// The reference joins read IDs against whole FASTQ files: {readId -> set(gapKey)} from the discordant lists and ALL
// lines of the scaffold lists (run_multi_threads_discordant.py:153-185), then streams left.fq and right.fq and appends
// matching records to gap_reads/{gapKey}.fastq (:209-241, 283-316).  With reads addressed by index (read = 2*pair + mate)
// that is: the SET of (gap, read) keys, per gap ordered by (mate, pair) = left-file stream order then right-file order,
// and a gather of the packed reads.  Keys come from the k-mer screen (gf_hit), the alignment tagger and the second hop
// (gf_taghit: target read = record's read, or its mate when to_mate is set).
//
//   keys -> per-gap histogram -> exclusive scan -> scatter into gap segments -> per-gap LDS bitonic sort + unique
//        -> exclusive scan of the unique counts (= pool_off) -> gather of packed reads
#include <cstring>

#include "gf_internal.hpp"

namespace gf {

constexpr uint32_t POOL_SORT_LDS = 4096;   // keys of one gap sorted in LDS at a time (16 KiB); longer key lists: chunks of this size + passes over the segment

__global__ void keys_from_screen_kernel(const gf_hit* hits, const uint32_t* n_hits, uint32_t hit_cap, int pairs,
                                        unsigned long long* keys, uint32_t key_cap, uint32_t* n_keys) {
    const uint32_t n = *n_hits < hit_cap ? *n_hits : hit_cap;
    const uint32_t per = pairs ? 2 : 1;
    __shared__ uint32_t s_base;
    // one block-wide reservation per pass: hits are dense here, no per-element global atomics
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {
        const uint32_t cnt = (n - i0 < blockDim.x ? n - i0 : blockDim.x) * per;
        if (threadIdx.x == 0) s_base = atomicAdd(n_keys, cnt);
        __syncthreads();
        const uint32_t i = i0 + threadIdx.x;
        if (i < n) {
            const gf_hit h = hits[i];
            const uint32_t o = s_base + threadIdx.x * per;
            if (o < key_cap) keys[o] = ((unsigned long long)h.gap << 32) | h.read;
            if (pairs && o + 1 < key_cap) keys[o + 1] = ((unsigned long long)h.gap << 32) | (h.read ^ 1u);
        }
        __syncthreads();
    }
}

// tagger hits: gap = index into the gap array (row_gap == null) or via the second hop's row -> gap map
__global__ void keys_from_tags_kernel(const gf_alnrec* recs, const gf_taghit* hits, const uint32_t* n_hits, uint32_t hit_cap,
                                      const uint32_t* row_gap, unsigned long long* keys, uint32_t key_cap, uint32_t* n_keys) {
    const uint32_t n = *n_hits < hit_cap ? *n_hits : hit_cap;
    __shared__ uint32_t s_base;
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {
        const uint32_t cnt = n - i0 < blockDim.x ? n - i0 : blockDim.x;
        if (threadIdx.x == 0) s_base = atomicAdd(n_keys, cnt);
        __syncthreads();
        const uint32_t i = i0 + threadIdx.x;
        if (i < n) {
            const gf_taghit h = hits[i];
            // (a record without a read — QNAME in no FASTQ record, gf_read_join_dev — recruits nothing: its key names no gap)
            const uint32_t own = (uint32_t)recs[h.rec].read, read = own ^ (h.to_mate ? 1u : 0u);
            const uint32_t gap = own == 0xFFFFFFFFu ? 0xFFFFFFFFu : (row_gap ? row_gap[h.gap] : h.gap);
            const uint32_t o = s_base + threadIdx.x;
            if (o < key_cap) keys[o] = ((unsigned long long)gap << 32) | read;
        }
        __syncthreads();
    }
}

// all three hit lists of one batch in one launch and without atomics: the counts are on the device, so every key has a fixed
// place — screen hits (with their mates) first, then the tagger's, then the second hop's
__global__ __launch_bounds__(256) void keys_all_kernel(const gf_hit* hits, const uint32_t* n_hits, uint32_t hit_cap, int pairs,
                                                       const gf_alnrec* recs, const gf_taghit* thits, const uint32_t* n_thits, uint32_t thit_cap,
                                                       const gf_taghit* lhits, const uint32_t* n_lhits, uint32_t lhit_cap, const uint32_t* row_gap,
                                                       unsigned long long* keys, uint32_t key_cap, uint32_t* n_keys) {
    const uint32_t per = pairs ? 2 : 1;
    const uint32_t n1 = (*n_hits < hit_cap ? *n_hits : hit_cap), n2 = thits ? (*n_thits < thit_cap ? *n_thits : thit_cap) : 0,
                   n3 = lhits ? (*n_lhits < lhit_cap ? *n_lhits : lhit_cap) : 0;
    const uint64_t total = (uint64_t)n1 * per + n2 + n3;
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_keys = total < 0xFFFFFFFFull ? (uint32_t)total : 0xFFFFFFFFu;   // > key_cap: truncated
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + (uint64_t)n2 + n3; i += (uint64_t)gridDim.x * blockDim.x) {
        if (i < n1) {
            const gf_hit h = hits[i];
            const uint64_t o = i * per;
            if (o < key_cap) keys[o] = ((unsigned long long)h.gap << 32) | h.read;
            if (pairs && o + 1 < key_cap) keys[o + 1] = ((unsigned long long)h.gap << 32) | (h.read ^ 1u);
        } else {
            const bool second = i >= (uint64_t)n1 + n2;
            const gf_taghit h = second ? lhits[i - n1 - n2] : thits[i - n1];
            const uint32_t own = (uint32_t)recs[h.rec].read, read = own ^ (h.to_mate ? 1u : 0u);
            const uint32_t gap = own == 0xFFFFFFFFu ? 0xFFFFFFFFu : (second ? row_gap[h.gap] : h.gap);   // no read (gf_read_join_dev): no gap
            const uint64_t o = (uint64_t)n1 * per + (i - n1);
            if (o < key_cap) keys[o] = ((unsigned long long)gap << 32) | read;
        }
    }
}

// keys arrive in runs of equal gap (hits of neighbouring reads), so a plain atomicAdd per key hammers one address;
// each wave first groups its lanes by gap: one atomic per distinct gap per wave
__global__ void pool_hist_kernel(const unsigned long long* keys, const uint32_t* n_keys, uint32_t key_cap, uint32_t n_gaps,
                                 uint32_t* cnt) {
    const uint32_t n = *n_keys < key_cap ? *n_keys : key_cap;
    const uint32_t n_round = (n + 63) & ~63u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += gridDim.x * blockDim.x) {
        uint32_t g = i < n ? (uint32_t)(keys[i] >> 32) : 0xFFFFFFFFu;
        bool todo = g < n_gaps;
        while (true) {
            const unsigned long long act = __ballot(todo);
            if (!act) break;
            const uint32_t g0 = __shfl(g, __ffsll((long long)act) - 1);
            const unsigned long long same = __ballot(todo && g == g0);
            if ((threadIdx.x & 63) == (uint32_t)(__ffsll((long long)same) - 1)) atomicAdd(&cnt[g0], (uint32_t)__popcll(same));
            if (g == g0) todo = false;
        }
    }
}

// single-block exclusive scan of cnt[0..n) into off[0..n] (64-bit when OUT64); optionally zeroes `zero`
template <bool OUT64>
__global__ __launch_bounds__(1024) void pool_scan_kernel(const uint32_t* cnt, uint32_t n, void* off_out, uint32_t* zero) {
    __shared__ unsigned long long part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = (n + 1023) / 1024;
    const uint32_t a = tid * chunk, b = a + chunk < n ? a + chunk : n;
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
        if (OUT64) ((unsigned long long*)off_out)[i] = run; else ((uint32_t*)off_out)[i] = (uint32_t)run;
        run += cnt[i];
        if (zero) zero[i] = 0;
    }
    if (tid == 1023) {
        if (OUT64) ((unsigned long long*)off_out)[n] = part[1023]; else ((uint32_t*)off_out)[n] = (uint32_t)part[1023];
    }
}

__global__ void pool_scatter_kernel(const unsigned long long* keys, const uint32_t* n_keys, uint32_t key_cap, uint32_t n_gaps,
                                    const uint32_t* seg_off, uint32_t* cursor, uint32_t* seg) {
    const uint32_t n = *n_keys < key_cap ? *n_keys : key_cap;
    const uint32_t n_round = (n + 63) & ~63u;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += gridDim.x * blockDim.x) {
        const unsigned long long k = i < n ? keys[i] : ~0ull;
        const uint32_t g = (uint32_t)(k >> 32), read = (uint32_t)k;
        bool todo = g < n_gaps;
        while (true) {   // one cursor atomic per distinct gap per wave
            const unsigned long long act = __ballot(todo);
            if (!act) break;
            const uint32_t g0 = __shfl(g, __ffsll((long long)act) - 1);
            const bool mine = todo && g == g0;
            const unsigned long long same = __ballot(mine);
            const uint32_t leader = __ffsll((long long)same) - 1;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&cursor[g0], (uint32_t)__popcll(same));
            base = __shfl(base, leader);
            if (mine) {
                seg[seg_off[g0] + base + __popcll(same & ((1ull << lane) - 1))] = ((read & 1u) << 31) | (read >> 1);  // (mate, pair)
                todo = false;
            }
        }
    }
}

// LDS-privatised forms of the two kernels above (used while one counter per gap fits in LDS): keys of one batch come in
// random gap order, so the per-wave grouping loops run ~64 rounds; here every block counts its slice of the keys into an LDS
// histogram with LDS atomics and touches the global counters once per non-empty bin
__global__ __launch_bounds__(1024) void pool_hist_lds_kernel(const unsigned long long* keys, const uint32_t* n_keys, uint32_t key_cap,
                                                            uint32_t n_gaps, uint32_t* cnt) {
    extern __shared__ uint32_t hist[];
    const uint32_t n = *n_keys < key_cap ? *n_keys : key_cap;
    for (uint32_t i = threadIdx.x; i < n_gaps; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const uint32_t per = (n + gridDim.x - 1) / gridDim.x, a = blockIdx.x * per, b = a + per < n ? a + per : n;
    for (uint32_t i = a + threadIdx.x; i < b; i += blockDim.x) {
        const uint32_t g = (uint32_t)(keys[i] >> 32);
        if (g < n_gaps) atomicAdd(&hist[g], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_gaps; i += blockDim.x)
        if (hist[i]) atomicAdd(&cnt[i], hist[i]);
}

__global__ __launch_bounds__(1024) void pool_scatter_lds_kernel(const unsigned long long* keys, const uint32_t* n_keys, uint32_t key_cap,
                                                               uint32_t n_gaps, const uint32_t* seg_off, uint32_t* cursor, uint32_t* seg) {
    extern __shared__ uint32_t hist[];   // pass 1: this block's count per gap; pass 2: its next free position per gap
    const uint32_t n = *n_keys < key_cap ? *n_keys : key_cap;
    for (uint32_t i = threadIdx.x; i < n_gaps; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const uint32_t per = (n + gridDim.x - 1) / gridDim.x, a = blockIdx.x * per, b = a + per < n ? a + per : n;   // same slices as the histogram
    for (uint32_t i = a + threadIdx.x; i < b; i += blockDim.x) {
        const uint32_t g = (uint32_t)(keys[i] >> 32);
        if (g < n_gaps) atomicAdd(&hist[g], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_gaps; i += blockDim.x)
        if (hist[i]) hist[i] = seg_off[i] + atomicAdd(&cursor[i], hist[i]);     // reserve this block's range of the gap's segment
    __syncthreads();
    for (uint32_t i = a + threadIdx.x; i < b; i += blockDim.x) {
        const unsigned long long k = keys[i];
        const uint32_t g = (uint32_t)(k >> 32), read = (uint32_t)k;
        if (g < n_gaps) seg[atomicAdd(&hist[g], 1u)] = ((read & 1u) << 31) | (read >> 1);   // (mate, pair); order inside a gap is set by the sort
    }
}

// Ascending compare-exchange network in LDS: the classic bitonic sort with every merge's first step mirrored (i against
// size - 1 - i), so that EVERY exchange moves the larger key to the higher index — keys padded with 0xFFFFFFFF at the top then
// never move, and a list of any length sorts in the first n places.  [first_size, m]: merge sizes to run (2 = sort from scratch;
// `first_size == 0`: only the half-cleaners of strides < m, the tail of a larger merge whose wide strides ran in global memory).
__device__ __forceinline__ void lds_sort_network(uint32_t* sk, uint32_t m, uint32_t first_size) {
    auto cleaners = [&](uint32_t from_stride) {
        for (uint32_t stride = from_stride; stride > 0; stride >>= 1) {
            for (uint32_t t = threadIdx.x; t < (m >> 1); t += blockDim.x) {
                const uint32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const uint32_t x = sk[lo], y = sk[hi];
                if (x > y) { sk[lo] = y; sk[hi] = x; }
            }
            __syncthreads();
        }
    };
    if (first_size == 0) { cleaners(m >> 1); return; }
    for (uint32_t size = first_size; size <= m; size <<= 1) {
        for (uint32_t t = threadIdx.x; t < (m >> 1); t += blockDim.x) {   // mirrored step
            const uint32_t half = size >> 1, blk = t / half, j = t - blk * half;
            const uint32_t lo = blk * size + j, hi = blk * size + size - 1 - j;
            const uint32_t x = sk[lo], y = sk[hi];
            if (x > y) { sk[lo] = y; sk[hi] = x; }
        }
        __syncthreads();
        cleaners(size >> 2);
    }
}

// one workgroup per gap: sort of the gap's keys, duplicates dropped, written back in place.  Up to C keys in LDS; a
// gap with more (a flank inside a repeat, a collapsed region: the reference has no bound, run_multi_threads_discordant.py:209-241)
// is sorted in place in global memory by the same network — chunks of C keys in LDS, the strides beyond a chunk as
// passes over the segment (the workgroup's own stores are visible to it after a barrier) — slower, never dropped.
__global__ __launch_bounds__(256) void pool_sort_unique_kernel(uint32_t n_gaps, const uint32_t* seg_off, uint32_t* seg,
                                                              uint32_t* ucnt, uint32_t C /* keys the LDS buffer holds, a power of two */) {
    extern __shared__ uint32_t sk[];
    __shared__ uint32_t s_n;
    for (uint32_t g = blockIdx.x; g < n_gaps; g += gridDim.x) {
        const uint32_t a = seg_off[g], n = seg_off[g + 1] - a;
        if (n == 0) { if (threadIdx.x == 0) ucnt[g] = 0; continue; }
        uint32_t m = 1;
        while (m < n) m <<= 1;
        const bool big = n > C;
        if (threadIdx.x == 0) s_n = 0;
        if (!big) {
            for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) sk[i] = i < n ? seg[a + i] : 0xFFFFFFFFu;
            __syncthreads();
            lds_sort_network(sk, m, 2);
        } else {
            uint32_t* S = seg + a;
            auto chunks = [&](uint32_t first_size) {   // every chunk of C keys through LDS
                for (uint32_t c0 = 0; c0 < n; c0 += C) {
                    for (uint32_t i = threadIdx.x; i < C; i += blockDim.x) sk[i] = c0 + i < n ? S[c0 + i] : 0xFFFFFFFFu;
                    __syncthreads();
                    lds_sort_network(sk, C, first_size);
                    for (uint32_t i = threadIdx.x; i < C; i += blockDim.x) if (c0 + i < n) S[c0 + i] = sk[i];
                    __syncthreads();
                }
            };
            __syncthreads();
            chunks(2);                                  // sorted runs of C
            for (uint32_t size = 2 * C; size <= m; size <<= 1) {
                for (uint32_t t = threadIdx.x; t < (m >> 1); t += blockDim.x) {   // mirrored step (hi >= n: a padding key, nothing moves)
                    const uint32_t half = size >> 1, blk = t / half, j = t - blk * half;
                    const uint32_t lo = blk * size + j, hi = blk * size + size - 1 - j;
                    if (hi < n) { const uint32_t x = S[lo], y = S[hi]; if (x > y) { S[lo] = y; S[hi] = x; } }
                }
                __syncthreads();
                for (uint32_t stride = size >> 2; stride >= C; stride >>= 1) {
                    for (uint32_t t = threadIdx.x; t < (m >> 1); t += blockDim.x) {
                        const uint32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                        if (hi < n) { const uint32_t x = S[lo], y = S[hi]; if (x > y) { S[lo] = y; S[hi] = x; } }
                    }
                    __syncthreads();
                }
                chunks(0);                              // strides C/2 .. 1
            }
        }
        // unique (sorted): keep key i if it differs from key i-1; order-preserving compaction by prefix count.  (In place for the
        // big gaps: a key moves to a lower or the same index, so nothing a later round reads has been overwritten.)
        for (uint32_t i0 = 0; i0 < n; i0 += blockDim.x) {
            const uint32_t i = i0 + threadIdx.x;
            uint32_t cur = 0, prev = 0;
            if (i < n) { cur = big ? seg[a + i] : sk[i]; prev = i ? (big ? seg[a + i - 1] : sk[i - 1]) : ~cur; }
            const bool keep = i < n && (i == 0 || cur != prev);
            // block-wide exclusive count of `keep` below this thread, in wave order
            const unsigned long long bal = __ballot(keep);
            __shared__ uint32_t wcnt[4];
            const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
            if (lane == 0) wcnt[w] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t base = s_n;
            for (uint32_t q = 0; q < w; ++q) base += wcnt[q];
            if (keep) seg[a + base + __popcll(bal & ((1ull << lane) - 1))] = cur;
            __syncthreads();
            if (threadIdx.x == 0) s_n += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            __syncthreads();
        }
        if (threadIdx.x == 0) ucnt[g] = s_n;
        __syncthreads();
    }
}

// one workgroup per gap: the gap's pooled reads are copied as one contiguous run of bytes
__global__ __launch_bounds__(256) void pool_gather_kernel(uint32_t n_gaps, const uint32_t* seg_off, const uint32_t* seg,
                                                         const unsigned long long* pool_off, const uint8_t* reads, uint32_t rb,
                                                         uint64_t n_reads, uint8_t* pool, uint64_t pool_cap_reads,
                                                         uint32_t* pool_ids, uint32_t* error) {
    // the pool buffer is too small for the recruits: say so (the gather stops at the capacity, pool_off stays exact)
    if (blockIdx.x == 0 && threadIdx.x == 0 && pool_off[n_gaps] > pool_cap_reads) atomicOr(error, 0x80000000u);
    for (uint32_t g = blockIdx.x; g < n_gaps; g += gridDim.x) {
        const unsigned long long p0 = pool_off[g], p1 = pool_off[g + 1];
        const uint32_t n = (uint32_t)(p1 - p0);
        const uint32_t s0 = seg_off[g];
        // a read starts at read * rb in both arrays; (unit index) / (units per read) in 32 bits — a gap's pool is far below 4 GB
        if (rb == 38) {   // 150-base reads: ten units per read — nine dwords and the last two bytes (a read starts at a multiple of 38 bytes: the
                          // dwords are 2-byte aligned, which the memory system takes as it is) —, a division by a constant, and four units of a
                          // thread in flight (every unit is two dependent loads: the key, then the read's bytes)
            typedef uint32_t u32_a2 __attribute__((aligned(2)));
            const uint8_t* src = reads;
            uint8_t* dst = pool + p0 * 38;
            const uint64_t room = pool_cap_reads > p0 ? pool_cap_reads - p0 : 0;
            const uint64_t units10 = (uint64_t)n * 10;
            const uint32_t lim = (uint32_t)(units10 < room * 10 ? units10 : room * 10);   // (a gap's pool is far below 4 G units)
            for (uint32_t i0 = threadIdx.x; i0 < lim; i0 += 4 * blockDim.x) {
                uint32_t rd[4], bb[4], v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t i = i0 + q * blockDim.x, j = i / 10u;
                    bb[q] = i - j * 10u;
                    const uint32_t key = i < lim ? seg[s0 + j] : 0u;
                    rd[q] = ((key & 0x7FFFFFFFu) << 1) | (key >> 31);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = 0;
                    if (i0 + q * blockDim.x < lim && rd[q] < n_reads) {
                        const uint8_t* a = src + (uint64_t)rd[q] * 38 + 4 * bb[q];
                        v[q] = bb[q] < 9 ? *reinterpret_cast<const u32_a2*>(a) : (uint32_t)*reinterpret_cast<const uint16_t*>(a);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (i0 + q * blockDim.x < lim && rd[q] < n_reads) {
                        const uint32_t i = i0 + q * blockDim.x;
                        uint8_t* a = dst + (uint64_t)(i / 10u) * 38 + 4 * bb[q];
                        if (bb[q] < 9) *reinterpret_cast<u32_a2*>(a) = v[q]; else *reinterpret_cast<uint16_t*>(a) = (uint16_t)v[q];
                    }
            }
        } else {   // any other read length: dword units at whatever alignment read * rb has (the memory system takes unaligned dwords), the
                   // last unit of a read its rb & 3 bytes
            typedef uint32_t u32_a1 __attribute__((aligned(1)));
            const uint32_t ud = (rb + 3) / 4, tail = rb & 3;
            const uint64_t unitsd = (uint64_t)n * ud;
            const uint32_t limd = unitsd < 0xFFFFFFFFull ? (uint32_t)unitsd : 0xFFFFFFFFu;
            for (uint32_t i = threadIdx.x; i < limd; i += blockDim.x) {
                const uint32_t j = i / ud, b = i - j * ud;
                if (p0 + j >= pool_cap_reads) break;
                const uint32_t key = seg[s0 + j];
                const uint32_t read = ((key & 0x7FFFFFFFu) << 1) | (key >> 31);
                if (read >= n_reads) continue;
                const uint8_t* a = reads + (uint64_t)read * rb + 4 * b;
                uint8_t* d = pool + (p0 + j) * rb + 4 * b;
                if (tail == 0 || b + 1 < ud) *reinterpret_cast<u32_a1*>(d) = *reinterpret_cast<const u32_a1*>(a);
                else for (uint32_t t = 0; t < tail; ++t) d[t] = a[t];
            }
        }
        if (pool_ids)
            for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
                if (p0 + j >= pool_cap_reads) break;
                const uint32_t key = seg[s0 + j];
                pool_ids[p0 + j] = ((key & 0x7FFFFFFFu) << 1) | (key >> 31);
            }
    }
}

// ---- §8e exchange step / a-5 library concatenation: pools regrouped by owner rank, and pools of several sources merged ----
// owner of gap g: batches of `batch` consecutive gaps dealt round-robin over the ranks
__device__ __forceinline__ uint32_t owner_of(uint32_t g, uint32_t batch, uint32_t world) { return (g / batch) % world; }

// exclusive scan of one value per thread over a 1024-thread block; *total = sum (same in every thread)
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* s_w /* >= 17 words */, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t x = v;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if ((int)lane >= d) x += y;
    }
    __syncthreads();
    if (lane == 63) s_w[w] = x;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (uint32_t q = 0; q < (blockDim.x >> 6); ++q) { const uint32_t t = s_w[q]; if (q < w) base += t; tot += t; }
    *total = tot;
    return base + x - v;
}

__global__ __launch_bounds__(1024) void pool_counts_kernel(const unsigned long long* pool_off, uint32_t n_gaps, uint32_t* cnt) {
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n_gaps; g += gridDim.x * blockDim.x)
        cnt[g] = (uint32_t)(pool_off[g + 1] - pool_off[g]);
}

// block d: rows of the gaps owned by rank d, in gap order -> their row offset inside d's slot; cnt[g] for every gap
// hdr (exact-size exchange): the counts of slot (d, lib) also go INTO the send buffer, as u32[n_gaps] at byte offset hdr[d] (zero for
// the gaps d does not own), so that the all-to-all carries them and no separate all-gather of counts is needed
__global__ __launch_bounds__(1024) void xchg_send_offsets_kernel(const unsigned long long* pool_off, uint32_t n_gaps, uint32_t world,
                                                                 uint32_t batch, uint32_t* cnt, uint32_t* dst_off, uint8_t* send,
                                                                 const unsigned long long* hdr) {
    __shared__ uint32_t s_w[20];
    const uint32_t d = blockIdx.x;
    uint32_t carry = 0;
    uint32_t* hc = hdr ? reinterpret_cast<uint32_t*>(send + hdr[d]) : nullptr;
    for (uint32_t g0 = 0; g0 < n_gaps; g0 += blockDim.x) {
        const uint32_t g = g0 + threadIdx.x;
        const bool mine = g < n_gaps && owner_of(g, batch, world) == d;
        const uint32_t n = mine ? (uint32_t)(pool_off[g + 1] - pool_off[g]) : 0u;
        uint32_t tot;
        const uint32_t ex = block_scan_excl(n, s_w, &tot);
        if (hc && g < n_gaps) hc[g] = n;
        if (mine) { cnt[g] = n; dst_off[g] = carry + ex; }
        carry += tot;
        __syncthreads();
    }
}

// one workgroup per gap: its rows go to slot (owner * n_lib + lib) of the send buffer
__global__ __launch_bounds__(256) void xchg_pack_kernel(const uint8_t* pool, const unsigned long long* pool_off, uint32_t n_gaps, uint32_t rb,
                                                        uint32_t world, uint32_t batch, uint32_t lib, uint32_t n_lib, const uint32_t* dst_off,
                                                        uint8_t* send, uint64_t cap_rows_all, uint32_t* error,
                                                        const unsigned long long* slot_base, const uint32_t* slot_cap) {
    // slot_base / slot_cap (exact-size exchange): slot s starts at BYTE offset slot_base[s] of `send` (even when rb is even) and holds
    // slot_cap[s] rows; null: equal slots of cap_rows_all rows, slot s at row s * cap_rows_all
    const uint32_t ur = (rb & 1) ? rb : rb / 2, ub = (rb & 1) ? 1u : 2u;
    for (uint32_t g = blockIdx.x; g < n_gaps; g += gridDim.x) {
        const unsigned long long p0 = pool_off[g];
        uint32_t n = (uint32_t)(pool_off[g + 1] - p0);
        if (!n) continue;
        const uint32_t o = dst_off[g];
        const uint64_t slot = (uint64_t)owner_of(g, batch, world) * n_lib + lib;
        const uint64_t cap_rows = slot_cap ? slot_cap[slot] : cap_rows_all;
        if ((uint64_t)o + n > cap_rows) {   // slot too small: say so, send what fits
            if (threadIdx.x == 0) atomicOr(error, 0x40000000u);
            n = o < cap_rows ? (uint32_t)(cap_rows - o) : 0u;
        }
        const uint64_t src0 = p0 * ur, dst0 = (slot_base ? slot_base[slot] / ub : slot * cap_rows * ur) + (uint64_t)o * ur;
        const uint64_t units = (uint64_t)n * ur;
        if (rb & 1) for (uint64_t i = threadIdx.x; i < units; i += blockDim.x) send[dst0 + i] = pool[src0 + i];
        else for (uint64_t i = threadIdx.x; i < units; i += blockDim.x)
            reinterpret_cast<uint16_t*>(send)[dst0 + i] = reinterpret_cast<const uint16_t*>(pool)[src0 + i];
    }
}

// blocks 0 .. n_slots-1: row offsets of my gaps inside slot i (a slot holds MY gaps' rows of one source, in gap order);
// block n_slots: merged pool_off (u64, n_gaps + 1) = scan over my gaps of the rows of all sources
// counts of slot q: cnt + q * n_gaps, or — exact-size exchange, the counts travel inside the received buffer — u32[n_gaps] at byte
// offset cnt_base[q] of `src`
__device__ __forceinline__ const uint32_t* slot_counts(const uint32_t* cnt, const uint8_t* src, const unsigned long long* cnt_base, uint32_t q, uint32_t n_gaps) {
    return cnt_base ? reinterpret_cast<const uint32_t*>(src + cnt_base[q]) : cnt + (uint64_t)q * n_gaps;
}
__global__ __launch_bounds__(1024) void merge_offsets_kernel(const uint32_t* cnt /* [n_slots][n_gaps] */, uint32_t n_slots, uint32_t n_gaps,
                                                             uint32_t rank, uint32_t world, uint32_t batch, uint32_t* src_off /* [n_slots][n_gaps] */,
                                                             unsigned long long* moff, const uint8_t* src, const unsigned long long* cnt_base) {
    __shared__ uint32_t s_w[20];
    const uint32_t i = blockIdx.x;
    unsigned long long carry = 0;
    for (uint32_t g0 = 0; g0 < n_gaps; g0 += blockDim.x) {
        const uint32_t g = g0 + threadIdx.x;
        const bool mine = g < n_gaps && owner_of(g, batch, world) == rank;
        uint32_t n = 0;
        if (mine) {
            if (i < n_slots) n = slot_counts(cnt, src, cnt_base, i, n_gaps)[g];
            else for (uint32_t q = 0; q < n_slots; ++q) n += slot_counts(cnt, src, cnt_base, q, n_gaps)[g];
        }
        uint32_t tot;
        const uint32_t ex = block_scan_excl(n, s_w, &tot);
        if (g < n_gaps) {
            if (i < n_slots) src_off[(uint64_t)i * n_gaps + g] = (uint32_t)carry + ex;
            else moff[g] = carry + ex;
        }
        carry += tot;
        __syncthreads();
    }
    if (i == n_slots && threadIdx.x == 0) moff[n_gaps] = carry;
}

// one workgroup per gap I own: the sources' rows one after the other — libraries in order (merge_reads.py:43-51: `cat` of the
// per-library files), inside a library the source ranks in order (contiguous read shards: global read order)
__global__ __launch_bounds__(256) void merge_copy_kernel(const uint8_t* src, uint64_t cap_rows, const uint32_t* cnt, const uint32_t* src_off,
                                                         uint32_t n_lib, uint32_t n_ranks_src, uint32_t n_gaps, uint32_t rb, uint32_t rank,
                                                         uint32_t world, uint32_t batch, const unsigned long long* moff, uint8_t* merged,
                                                         uint64_t merged_cap_rows, uint32_t* error, const unsigned long long* slot_base,
                                                         const unsigned long long* cnt_base) {
    const uint32_t ur = (rb & 1) ? rb : rb / 2, ub = (rb & 1) ? 1u : 2u;
    if (blockIdx.x == 0 && threadIdx.x == 0 && moff[n_gaps] > merged_cap_rows) atomicOr(error, 0x80000000u);
    for (uint32_t g = blockIdx.x; g < n_gaps; g += gridDim.x) {
        if (owner_of(g, batch, world) != rank) continue;
        unsigned long long at = moff[g];
        for (uint32_t l = 0; l < n_lib; ++l)
            for (uint32_t r = 0; r < n_ranks_src; ++r) {
                const uint64_t slot = (uint64_t)r * n_lib + l;
                const uint32_t n_all = slot_counts(cnt, src, cnt_base, (uint32_t)slot, n_gaps)[g];
                uint32_t n = n_all;
                if (!n) continue;
                if (at + n > merged_cap_rows) n = at < merged_cap_rows ? (uint32_t)(merged_cap_rows - at) : 0u;
                const uint64_t src0 = (slot_base ? slot_base[slot] / ub : slot * cap_rows * ur) + (uint64_t)src_off[slot * n_gaps + g] * ur, dst0 = at * ur;
                const uint64_t units = (uint64_t)n * ur;
                if (rb & 1) for (uint64_t i = threadIdx.x; i < units; i += blockDim.x) merged[dst0 + i] = src[src0 + i];
                else for (uint64_t i = threadIdx.x; i < units; i += blockDim.x)
                    reinterpret_cast<uint16_t*>(merged)[dst0 + i] = reinterpret_cast<const uint16_t*>(src)[src0 + i];
                at += n_all;
            }
    }
}

}  // namespace gf

using namespace gf;

extern "C" {

int gf_pool_keys_reset(gf_ctx* ctx, void* d_n_keys) {
    if (!ctx || !d_n_keys) return GF_E_INVAL;
    GF_HIP(ctx, hipMemsetAsync(d_n_keys, 0, 4, ctx->stream));
    return GF_OK;
}

int gf_pool_keys_from_screen_dev(gf_ctx* ctx, const void* d_hits, const void* d_n_hits, size_t hit_cap, int pairs, void* d_keys,
                                 size_t key_cap, void* d_n_keys) {
    if (!ctx || !d_hits || !d_n_hits || !d_keys || !d_n_keys || hit_cap > 0xFFFFFFFFull || key_cap > 0xFFFFFFFFull) return GF_E_INVAL;
    LaunchTimer tm(ctx, GF_KERNEL_POOL);
    hipLaunchKernelGGL(keys_from_screen_kernel, dim3(ctx->n_cu * 4), dim3(256), 0, ctx->stream, (const gf_hit*)d_hits,
                       (const uint32_t*)d_n_hits, (uint32_t)hit_cap, pairs, (unsigned long long*)d_keys, (uint32_t)key_cap,
                       (uint32_t*)d_n_keys);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_pool_keys_from_tags_dev(gf_ctx* ctx, const void* d_recs, const void* d_taghits, const void* d_n_taghits, size_t hit_cap,
                               const gf_dpos* lowmapq_table_or_null, size_t n_rows, void* d_keys, size_t key_cap, void* d_n_keys) {
    if (!ctx || !d_recs || !d_taghits || !d_n_taghits || !d_keys || !d_n_keys || hit_cap > 0xFFFFFFFFull || key_cap > 0xFFFFFFFFull)
        return GF_E_INVAL;
    const uint32_t* d_row_gap = nullptr;
    const bool cached = lowmapq_table_or_null && ctx->rowgap.p && ctx->rowgap_rows.size() == n_rows * 4 &&
                        memcmp(ctx->rowgap_rows.data(), lowmapq_table_or_null, n_rows * sizeof(gf_dpos)) == 0;
    if (cached) d_row_gap = (const uint32_t*)ctx->rowgap.p;
    if (lowmapq_table_or_null && !cached) {  // second-hop rows name (src_scaffold, 1-based gap) -> index into the gap array
        std::vector<uint32_t> off(ctx->n_scaffolds + 1, 0), rg(n_rows);
        for (const gf_gap& g : ctx->gaps) off[g.scaffold + 1]++;
        for (uint32_t s = 0; s < ctx->n_scaffolds; ++s) off[s + 1] += off[s];
        for (size_t r = 0; r < n_rows; ++r) {
            const gf_dpos& t = lowmapq_table_or_null[r];
            if (t.src_scaffold >= ctx->n_scaffolds || t.src_gap == 0 || off[t.src_scaffold] + t.src_gap - 1 >= off[t.src_scaffold + 1])
                return GF_E_INVAL;
            rg[r] = off[t.src_scaffold] + t.src_gap - 1;
        }
        int rc = ensure(ctx, ctx->rowgap, n_rows * 4 + 64);
        if (rc) return rc;
        GF_HIP(ctx, hipMemcpyAsync(ctx->rowgap.p, rg.data(), n_rows * 4, hipMemcpyHostToDevice, ctx->stream));
        GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        d_row_gap = (const uint32_t*)ctx->rowgap.p;
        ctx->rowgap_rows.assign((const uint32_t*)lowmapq_table_or_null, (const uint32_t*)lowmapq_table_or_null + n_rows * 4);
    }
    LaunchTimer tm(ctx, GF_KERNEL_POOL);
    hipLaunchKernelGGL(keys_from_tags_kernel, dim3(ctx->n_cu * 4), dim3(256), 0, ctx->stream, (const gf_alnrec*)d_recs,
                       (const gf_taghit*)d_taghits, (const uint32_t*)d_n_taghits, (uint32_t)hit_cap, d_row_gap,
                       (unsigned long long*)d_keys, (uint32_t)key_cap, (uint32_t*)d_n_keys);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_pool_keys_from_second_hop_dev(gf_ctx* ctx, const void* d_recs, const void* d_taghits, const void* d_n_taghits, size_t hit_cap,
                                     const void* d_row_gap, void* d_keys, size_t key_cap, void* d_n_keys) {
    if (!ctx || !d_recs || !d_taghits || !d_n_taghits || !d_row_gap || !d_keys || !d_n_keys || hit_cap > 0xFFFFFFFFull || key_cap > 0xFFFFFFFFull)
        return GF_E_INVAL;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    LaunchTimer tm(ctx, GF_KERNEL_POOL);
    hipLaunchKernelGGL(keys_from_tags_kernel, dim3(ctx->n_cu * 4), dim3(256), 0, ctx->stream, (const gf_alnrec*)d_recs,
                       (const gf_taghit*)d_taghits, (const uint32_t*)d_n_taghits, (uint32_t)hit_cap, (const uint32_t*)d_row_gap,
                       (unsigned long long*)d_keys, (uint32_t)key_cap, (uint32_t*)d_n_keys);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_pool_keys_all_dev(gf_ctx* ctx, const void* d_hits, const void* d_n_hits, size_t hit_cap, int pairs, const void* d_recs,
                         const void* d_taghits, const void* d_n_taghits, size_t taghit_cap, const void* d_hophits, const void* d_n_hophits,
                         size_t hophit_cap, const void* d_row_gap, void* d_keys, size_t key_cap, void* d_n_keys) {
    if (!ctx || !d_hits || !d_n_hits || !d_keys || !d_n_keys || hit_cap > 0xFFFFFFFFull || taghit_cap > 0xFFFFFFFFull ||
        hophit_cap > 0xFFFFFFFFull || key_cap > 0xFFFFFFFFull || (d_taghits && (!d_n_taghits || !d_recs)) ||
        (d_hophits && (!d_n_hophits || !d_recs || !d_row_gap)))
        return GF_E_INVAL;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    LaunchTimer tm(ctx, GF_KERNEL_POOL);
    hipLaunchKernelGGL(keys_all_kernel, dim3(ctx->n_cu * 4), dim3(256), 0, ctx->stream, (const gf_hit*)d_hits, (const uint32_t*)d_n_hits,
                       (uint32_t)hit_cap, pairs, (const gf_alnrec*)d_recs, (const gf_taghit*)d_taghits, (const uint32_t*)d_n_taghits,
                       (uint32_t)taghit_cap, (const gf_taghit*)d_hophits, (const uint32_t*)d_n_hophits, (uint32_t)hophit_cap,
                       (const uint32_t*)d_row_gap, (unsigned long long*)d_keys, (uint32_t)key_cap, (uint32_t*)d_n_keys);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_build_pools_dev(gf_ctx* ctx, const void* d_packed_reads, size_t n_reads, int read_len, const void* d_keys,
                       const void* d_n_keys, size_t key_cap, void* d_pool_packed, size_t pool_cap_reads, void* d_pool_off,
                       void* d_pool_read_ids, void* d_error) {
    if (!ctx || !d_keys || !d_n_keys || !d_pool_off || !d_error || (n_reads && !d_packed_reads) ||
        (pool_cap_reads && !d_pool_packed) || key_cap > 0xFFFFFFFFull || read_len <= 0)
        return GF_E_INVAL;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t ng = (uint32_t)ctx->gaps.size();
    const uint32_t rb = (uint32_t)gf_packed_read_bytes(read_len);
    int rc;
    // workspace: cnt[ng] cursor[ng] ucnt[ng] seg_off[ng+1] | seg[key_cap]
    const size_t w_small = ((size_t)(4 * (size_t)ng + 8) * 4 + 255) & ~(size_t)255;
    if ((rc = ensure(ctx, ctx->pool_ws, w_small + key_cap * 4 + 256))) return rc;
    uint32_t* cnt = (uint32_t*)ctx->pool_ws.p;
    uint32_t* cursor = cnt + ng;
    uint32_t* ucnt = cursor + ng;
    uint32_t* seg_off = ucnt + ng;
    uint32_t* seg = (uint32_t*)((uint8_t*)ctx->pool_ws.p + w_small);
    zero_regions(ctx, ZeroList{{cnt, (uint32_t*)d_error, nullptr, nullptr}, {ng + 1, 1, 0, 0}});
    LaunchTimer tm(ctx, GF_KERNEL_POOL);
    const unsigned blocks = ctx->n_cu * 4;
    const bool lds_bins = ng && (size_t)ng * 4 <= 128 * 1024;     // one LDS counter per gap
    // slices of at least 16 384 keys, one block per CU at most (64 at least): a block streams its slice with one load per thread in flight
    // — the two kernels wait on that, not on the atomics: 64 blocks for C4's 10 M keys 0.25 ms each, 256 blocks 0.16 —, and its global
    // atomics are one per non-empty bin
    const unsigned lblocks = (unsigned)std::min<size_t>(ctx->n_cu, std::max<size_t>(64, key_cap / 16384));
                                                                  // (4-wave blocks left each of the 64 CUs waiting on its own key loads: 0.29 + 0.47 ms per 10 M keys)
    if (lds_bins) {
        hipLaunchKernelGGL(pool_hist_lds_kernel, dim3(lblocks), dim3(1024), (size_t)ng * 4, ctx->stream, (const unsigned long long*)d_keys,
                           (const uint32_t*)d_n_keys, (uint32_t)key_cap, ng, cnt);
    } else if (ng) {
        hipLaunchKernelGGL(pool_hist_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const unsigned long long*)d_keys,
                           (const uint32_t*)d_n_keys, (uint32_t)key_cap, ng, cnt);
    }
    hipLaunchKernelGGL(pool_scan_kernel<false>, dim3(1), dim3(1024), 0, ctx->stream, cnt, ng, (void*)seg_off, cursor);
    if (ng) {
        if (lds_bins)
            hipLaunchKernelGGL(pool_scatter_lds_kernel, dim3(lblocks), dim3(1024), (size_t)ng * 4, ctx->stream, (const unsigned long long*)d_keys,
                               (const uint32_t*)d_n_keys, (uint32_t)key_cap, ng, seg_off, cursor, seg);
        else
            hipLaunchKernelGGL(pool_scatter_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const unsigned long long*)d_keys,
                               (const uint32_t*)d_n_keys, (uint32_t)key_cap, ng, seg_off, cursor, seg);
        // LDS per workgroup: 4 096 keys (16 KiB, nine workgroups per CU) cover the pools of every BASELINE configuration; a longer key list is
        // sorted in chunks of that size + passes over its segment.  (64 KiB per workgroup — two per CU — for every gap: 0.68 ms at C4.)
        hipLaunchKernelGGL(pool_sort_unique_kernel, dim3(std::min<unsigned>(ng, ctx->n_cu * 8)), dim3(256), POOL_SORT_LDS * 4,
                           ctx->stream, ng, seg_off, seg, ucnt, POOL_SORT_LDS);
    }
    hipLaunchKernelGGL(pool_scan_kernel<true>, dim3(1), dim3(1024), 0, ctx->stream, ucnt, ng, d_pool_off, (uint32_t*)nullptr);
    if (ng && pool_cap_reads) {
        hipLaunchKernelGGL(pool_gather_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ng, seg_off, seg,
                           (const unsigned long long*)d_pool_off, (const uint8_t*)d_packed_reads, rb, (uint64_t)n_reads,
                           (uint8_t*)d_pool_packed, (uint64_t)pool_cap_reads, (uint32_t*)d_pool_read_ids, (uint32_t*)d_error);
    }
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_pool_counts_dev(gf_ctx* ctx, const void* d_pool_off, size_t n_gaps, void* d_cnt) {
    if (!ctx || !d_pool_off || !d_cnt || n_gaps > 0xFFFFFFF0ull) return GF_E_INVAL;
    if (!n_gaps) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    LaunchTimer tm(ctx, GF_KERNEL_POOL);
    hipLaunchKernelGGL(pool_counts_kernel, dim3((unsigned)std::min<size_t>((n_gaps + 1023) / 1024, 256)), dim3(1024), 0, ctx->stream,
                       (const unsigned long long*)d_pool_off, (uint32_t)n_gaps, (uint32_t*)d_cnt);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

static int pools_pack_for_owners(gf_ctx* ctx, const void* d_pool, const void* d_pool_off, size_t n_gaps, int read_len, int world,
                                 int batch, int lib, int n_lib, void* d_send, size_t cap_rows, const void* d_slot_base, const void* d_slot_cap,
                                 const void* d_cnt_base, void* d_cnt, void* d_error) {
    if (!ctx || !d_pool_off || !d_send || !d_cnt || !d_error || world < 1 || batch < 1 || n_lib < 1 || lib < 0 || lib >= n_lib ||
        read_len <= 0 || n_gaps > 0xFFFFFFF0ull || cap_rows > 0xFFFFFFFFull)
        return GF_E_INVAL;
    if (!n_gaps) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->xchg_ws, n_gaps * 4 + 256 + (size_t)world * 8))) return rc;
    uint32_t* dst_off = (uint32_t*)ctx->xchg_ws.p;
    const uint32_t rb = (uint32_t)gf_packed_read_bytes(read_len);
    LaunchTimer tm(ctx, GF_KERNEL_POOL);
    // (the header offsets of THIS library's slots: entries (d * n_lib + lib) of d_cnt_base, gathered into a dense [world] table on the device)
    unsigned long long* hdr = nullptr;
    if (d_cnt_base) {
        hdr = (unsigned long long*)((uint8_t*)ctx->xchg_ws.p + ((n_gaps * 4 + 255) & ~(size_t)255));
        GF_HIP(ctx, hipMemcpy2DAsync(hdr, 8, (const unsigned long long*)d_cnt_base + lib, (size_t)n_lib * 8, 8, (size_t)world, hipMemcpyDeviceToDevice,
                                     ctx->stream));
    }
    hipLaunchKernelGGL(xchg_send_offsets_kernel, dim3((unsigned)world), dim3(1024), 0, ctx->stream, (const unsigned long long*)d_pool_off,
                       (uint32_t)n_gaps, (uint32_t)world, (uint32_t)batch, (uint32_t*)d_cnt, dst_off, (uint8_t*)d_send, (const unsigned long long*)hdr);
    hipLaunchKernelGGL(xchg_pack_kernel, dim3((unsigned)std::min<size_t>(n_gaps, (size_t)ctx->n_cu * 8)), dim3(256), 0, ctx->stream,
                       (const uint8_t*)d_pool, (const unsigned long long*)d_pool_off, (uint32_t)n_gaps, rb, (uint32_t)world, (uint32_t)batch,
                       (uint32_t)lib, (uint32_t)n_lib, dst_off, (uint8_t*)d_send, (uint64_t)cap_rows, (uint32_t*)d_error,
                       (const unsigned long long*)d_slot_base, (const uint32_t*)d_slot_cap);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_pools_pack_for_owners_dev(gf_ctx* ctx, const void* d_pool, const void* d_pool_off, size_t n_gaps, int read_len, int world,
                                 int batch, int lib, int n_lib, void* d_send, size_t cap_rows, void* d_cnt, void* d_error) {
    return pools_pack_for_owners(ctx, d_pool, d_pool_off, n_gaps, read_len, world, batch, lib, n_lib, d_send, cap_rows, nullptr, nullptr, nullptr,
                                 d_cnt, d_error);
}

int gf_pools_pack_for_owners_v_dev(gf_ctx* ctx, const void* d_pool, const void* d_pool_off, size_t n_gaps, int read_len, int world,
                                   int batch, int lib, int n_lib, void* d_send, const void* d_slot_base, const void* d_slot_cap,
                                   const void* d_cnt_base, void* d_cnt, void* d_error) {
    if (!d_slot_base || !d_slot_cap) return GF_E_INVAL;
    return pools_pack_for_owners(ctx, d_pool, d_pool_off, n_gaps, read_len, world, batch, lib, n_lib, d_send, 0, d_slot_base, d_slot_cap, d_cnt_base,
                                 d_cnt, d_error);
}

static int pools_merge(gf_ctx* ctx, const void* d_src, size_t cap_rows, const void* d_slot_base, const void* d_cnt, const void* d_cnt_base,
                       int n_lib, int n_src_ranks, size_t n_gaps, int read_len, int rank, int world, int batch, void* d_merged,
                       size_t merged_cap_rows, void* d_merged_off, void* d_error) {
    if (!ctx || !d_src || (!d_cnt && !d_cnt_base) || !d_merged_off || !d_error || (merged_cap_rows && !d_merged) || n_lib < 1 || n_src_ranks < 1 ||
        world < 1 || rank < 0 || rank >= world || batch < 1 || read_len <= 0 || n_gaps > 0xFFFFFFF0ull || cap_rows > 0xFFFFFFFFull)
        return GF_E_INVAL;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    if (!n_gaps) { GF_HIP(ctx, hipMemsetAsync(d_merged_off, 0, 8, ctx->stream)); return GF_OK; }
    const uint32_t n_slots = (uint32_t)n_lib * (uint32_t)n_src_ranks;
    int rc;
    if ((rc = ensure(ctx, ctx->xchg_ws2, (size_t)n_slots * n_gaps * 4 + 256))) return rc;
    uint32_t* src_off = (uint32_t*)ctx->xchg_ws2.p;
    const uint32_t rb = (uint32_t)gf_packed_read_bytes(read_len);
    LaunchTimer tm(ctx, GF_KERNEL_POOL);
    hipLaunchKernelGGL(merge_offsets_kernel, dim3(n_slots + 1), dim3(1024), 0, ctx->stream, (const uint32_t*)d_cnt, n_slots, (uint32_t)n_gaps,
                       (uint32_t)rank, (uint32_t)world, (uint32_t)batch, src_off, (unsigned long long*)d_merged_off, (const uint8_t*)d_src,
                       (const unsigned long long*)d_cnt_base);
    hipLaunchKernelGGL(merge_copy_kernel, dim3((unsigned)std::min<size_t>(n_gaps, (size_t)ctx->n_cu * 8)), dim3(256), 0, ctx->stream,
                       (const uint8_t*)d_src, (uint64_t)cap_rows, (const uint32_t*)d_cnt, src_off, (uint32_t)n_lib, (uint32_t)n_src_ranks,
                       (uint32_t)n_gaps, rb, (uint32_t)rank, (uint32_t)world, (uint32_t)batch, (const unsigned long long*)d_merged_off,
                       (uint8_t*)d_merged, (uint64_t)merged_cap_rows, (uint32_t*)d_error, (const unsigned long long*)d_slot_base,
                       (const unsigned long long*)d_cnt_base);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_pools_merge_dev(gf_ctx* ctx, const void* d_src, size_t cap_rows, const void* d_cnt, int n_lib, int n_src_ranks, size_t n_gaps,
                       int read_len, int rank, int world, int batch, void* d_merged, size_t merged_cap_rows, void* d_merged_off,
                       void* d_error) {
    if (!d_cnt) return GF_E_INVAL;
    return pools_merge(ctx, d_src, cap_rows, nullptr, d_cnt, nullptr, n_lib, n_src_ranks, n_gaps, read_len, rank, world, batch, d_merged,
                       merged_cap_rows, d_merged_off, d_error);
}

int gf_pools_merge_v_dev(gf_ctx* ctx, const void* d_src, const void* d_slot_base, const void* d_cnt_base, int n_lib, int n_src_ranks,
                         size_t n_gaps, int read_len, int rank, int world, int batch, void* d_merged, size_t merged_cap_rows,
                         void* d_merged_off, void* d_error) {
    if (!d_slot_base || !d_cnt_base) return GF_E_INVAL;
    return pools_merge(ctx, d_src, 0, d_slot_base, nullptr, d_cnt_base, n_lib, n_src_ranks, n_gaps, read_len, rank, world, batch, d_merged,
                       merged_cap_rows, d_merged_off, d_error);
}

}  // extern "C"
