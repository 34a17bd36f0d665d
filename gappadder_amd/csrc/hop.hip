// Second-hop table on the device.  The reference inverts every `discordant` list line into "mateScaffold matePos srcScaffold
// srcGap", runs sort(1) -k1n -k2n -k3n -k4n over the file and splits it per mate scaffold (run_multi_threads_discordant.py:19-122);
// collect_discordant_low_mapq_reads.py:4-28 then looks MAPQ-0 records up in it.  Here the rows are cut from the tagger's hits
// where they are (HBM), sorted by a counting sort into buckets + one LDS sort per bucket, and the look-up arrays of the second-hop kernel
// (unique positions per scaffold + row ranges) are derived from them by three small kernels — no host copy, no host sort, no library sort,
// nothing cached between batches.
#include <cstring>

#include "gf_internal.hpp"

namespace gf {

// one row per DISCORDANT hit whose mate lies on a known scaffold: key = mate scaffold << 32 | mate position, value = gap index
__global__ __launch_bounds__(256) void hop_extract_kernel(const gf_alnrec* recs, const gf_taghit* hits, const uint32_t* n_hits, uint32_t hit_cap,
                                                          uint32_t n_scaffolds, unsigned long long* keys, uint32_t* vals, uint32_t row_cap,
                                                          uint32_t* n_rows, uint32_t* max_pos) {
    // rows are buffered per wave in LDS and leave with ONE returning atomic per 33-96 of them: a discordant hit in nearly every 64-hit step
    // made that one atomic per step on a single counter — 37 000 of them, 0.3 ms for a kernel that reads 30 MB
    constexpr uint32_t RB = 96;
    __shared__ unsigned long long s_k[4][RB];
    __shared__ uint32_t s_v[4][RB];
    const uint32_t n = *n_hits < hit_cap ? *n_hits : hit_cap;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n_round = (n + 63) & ~63u;
    uint32_t held = 0, mp = 0;       // held: wave-uniform
    auto flush = [&]() {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(n_rows, held);   // may exceed row_cap: consumers clamp, the host variant reports it
        base = __shfl(base, 0);
        for (uint32_t i = lane; i < held; i += 64)
            if (base + i < row_cap) { keys[base + i] = s_k[wv][i]; vals[base + i] = s_v[wv][i]; }
        held = 0;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += gridDim.x * blockDim.x) {
        bool take = false;
        unsigned long long key = 0;
        uint32_t gap = 0;
        if (i < n) {
            const gf_taghit h = hits[i];
            if (h.kind == GF_KIND_DISCORDANT) {
                const gf_alnrec r = recs[h.rec];
                take = r.mate_ref < n_scaffolds;
                key = ((unsigned long long)r.mate_ref << 32) | r.mate_pos;
                gap = h.gap;
            }
        }
        const unsigned long long bal = __ballot(take);
        if (!bal) continue;
        const uint32_t cnt = (uint32_t)__popcll(bal);
        if (held + cnt > RB) flush();
        if (take) {
            const uint32_t at = held + (uint32_t)__popcll(bal & ((1ull << lane) - 1));
            s_k[wv][at] = key;
            s_v[wv][at] = gap;
            mp = (uint32_t)key > mp ? (uint32_t)key : mp;
        }
        held += cnt;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (held > RB - 64) flush();
    }
    if (held) flush();
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t y = __shfl_xor(mp, d); mp = y > mp ? y : mp; }
    if (lane == 0 && mp) atomicMax(max_pos, mp);
}

// ---- the rows sorted by (mate scaffold, mate position) without a library sort.  rocPRIM's radix_sort_pairs falls back to its merge sort
// for a few hundred thousand rows — two kernels, 0.9 ms per C4 step for 3.4e5 padded rows: nearly the whole second-hop phase.  The keys are
// spread evenly (mates of chimeric pairs), so: a counting sort into <= 16 384 buckets by (scaffold, top bits of the position) — the split of
// the position follows the largest position seen —, then every bucket (tens of rows) is sorted by one wave in LDS with the all-ascending
// bitonic network on (position << 32 | gap) words; a bucket beyond the LDS buffer is sorted in place in global memory by the same
// network (chunks through LDS + passes over the bucket).  Order among equal keys: by gap index (the reference's sort breaks the tie
// by source scaffold and gap, `-k3n -k4n`: the same for gaps listed in scaffold order).
constexpr uint32_t HOP_BUCKETS_LOG2 = 14, HOP_SORT_LDS = 1024;
struct HopBuckets {
    uint32_t scaf_bits, sub_bits, pos_shift, n_buckets;
};
__device__ __forceinline__ HopBuckets hop_buckets(uint32_t n_scaffolds, uint32_t max_pos) {
    HopBuckets h;
    h.scaf_bits = 0;
    while (h.scaf_bits < 32 && (n_scaffolds >> h.scaf_bits) != 0) ++h.scaf_bits;     // the sentinel scaffold n_scaffolds included
    h.sub_bits = h.scaf_bits < HOP_BUCKETS_LOG2 ? HOP_BUCKETS_LOG2 - h.scaf_bits : 0;
    uint32_t pb = 0;
    while (pb < 32 && (max_pos >> pb) != 0) ++pb;
    h.pos_shift = pb > h.sub_bits ? pb - h.sub_bits : 0;
    h.n_buckets = (n_scaffolds + 1) << h.sub_bits;
    return h;
}
__device__ __forceinline__ uint32_t hop_bucket_of(const HopBuckets& h, unsigned long long key) {
    const uint32_t sub = (uint32_t)key >> h.pos_shift;
    return ((uint32_t)(key >> 32) << h.sub_bits) | (sub < (1u << h.sub_bits) ? sub : (1u << h.sub_bits) - 1u);
}
__global__ __launch_bounds__(256) void hop_bucket_count_kernel(const unsigned long long* keys, const uint32_t* n_rows, uint32_t row_cap, uint32_t n_scaffolds,
                                                               const uint32_t* max_pos, uint32_t* cnt) {
    const uint32_t n = *n_rows < row_cap ? *n_rows : row_cap;
    const HopBuckets h = hop_buckets(n_scaffolds, *max_pos);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) atomicAdd(&cnt[hop_bucket_of(h, keys[i])], 1u);
}
// exclusive scan of the bucket counts (one workgroup; <= 2^15 buckets); cursor = a copy of the offsets
__global__ __launch_bounds__(1024) void hop_bucket_scan_kernel(const uint32_t* cnt, uint32_t n_scaffolds, const uint32_t* max_pos, uint32_t* off, uint32_t* cursor) {
    __shared__ uint32_t part[1024];
    const uint32_t nb = hop_buckets(n_scaffolds, *max_pos).n_buckets, tid = threadIdx.x;
    const uint32_t chunk = (nb + 1023) / 1024, a = tid * chunk < nb ? tid * chunk : nb, b = a + chunk < nb ? a + chunk : nb;
    uint32_t c = 0;
    for (uint32_t i = a; i < b; ++i) c += cnt[i];
    part[tid] = c;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = tid ? part[tid - 1] : 0;
    for (uint32_t i = a; i < b; ++i) { off[i] = run; cursor[i] = run; run += cnt[i]; }
    if (tid == 1023) off[nb] = part[1023];
}
__global__ __launch_bounds__(256) void hop_bucket_scatter_kernel(const unsigned long long* keys, const uint32_t* vals, const uint32_t* n_rows, uint32_t row_cap,
                                                                 uint32_t n_scaffolds, const uint32_t* max_pos, uint32_t* cursor, unsigned long long* k_out,
                                                                 uint32_t* v_out) {
    const uint32_t n = *n_rows < row_cap ? *n_rows : row_cap;
    const HopBuckets h = hop_buckets(n_scaffolds, *max_pos);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long k = keys[i];
        const uint32_t at = atomicAdd(&cursor[hop_bucket_of(h, k)], 1u);
        k_out[at] = k;
        v_out[at] = vals[i];
    }
}
// all-ascending bitonic network on 64-bit words in LDS (see pools.hip: padding words at the top never move); one wave
__device__ __forceinline__ void hop_lds_network(unsigned long long* sk, uint32_t m, uint32_t first_size) {
    const uint32_t lane = threadIdx.x;
    auto cleaners = [&](uint32_t from_stride) {
        for (uint32_t stride = from_stride; stride > 0; stride >>= 1) {
            for (uint32_t t = lane; t < (m >> 1); t += 64) {
                const uint32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const unsigned long long x = sk[lo], y = sk[hi];
                if (x > y) { sk[lo] = y; sk[hi] = x; }
            }
            __syncthreads();
        }
    };
    if (first_size == 0) { cleaners(m >> 1); return; }
    for (uint32_t size = first_size; size <= m; size <<= 1) {
        for (uint32_t t = lane; t < (m >> 1); t += 64) {
            const uint32_t half = size >> 1, blk = t / half, j = t - blk * half;
            const uint32_t lo = blk * size + j, hi = blk * size + size - 1 - j;
            const unsigned long long x = sk[lo], y = sk[hi];
            if (x > y) { sk[lo] = y; sk[hi] = x; }
        }
        __syncthreads();
        cleaners(size >> 2);
    }
}
// one wave per bucket: its rows as (position << 32 | gap) words, sorted, written back in place
__global__ __launch_bounds__(64) void hop_bucket_sort_kernel(const uint32_t* off, uint32_t n_scaffolds, const uint32_t* max_pos, unsigned long long* k_out, uint32_t* v_out) {
    __shared__ unsigned long long sk[HOP_SORT_LDS];
    const uint32_t nb = hop_buckets(n_scaffolds, *max_pos).n_buckets, lane = threadIdx.x;
    constexpr uint32_t C = HOP_SORT_LDS;
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const uint32_t a = off[b], n = off[b + 1] - a;
        if (n < 2) continue;
        const unsigned long long hi32 = k_out[a] & 0xFFFFFFFF00000000ull;       // the bucket's scaffold
        auto get = [&](uint32_t i) -> unsigned long long { return (k_out[a + i] << 32) | v_out[a + i]; };
        auto put = [&](uint32_t i, unsigned long long w) { k_out[a + i] = hi32 | (w >> 32); v_out[a + i] = (uint32_t)w; };
        uint32_t m = 1;
        while (m < n) m <<= 1;
        __syncthreads();
        if (n <= C) {
            for (uint32_t i = lane; i < m; i += 64) sk[i] = i < n ? get(i) : ~0ull;
            __syncthreads();
            hop_lds_network(sk, m, 2);
            for (uint32_t i = lane; i < n; i += 64) put(i, sk[i]);
            __syncthreads();
            continue;
        }
        // a bucket beyond the LDS buffer (all mates at one place): chunks through LDS, the wide strides in global memory
        auto chunks = [&](uint32_t first_size) {
            for (uint32_t c0 = 0; c0 < n; c0 += C) {
                for (uint32_t i = lane; i < C; i += 64) sk[i] = c0 + i < n ? get(c0 + i) : ~0ull;
                __syncthreads();
                hop_lds_network(sk, C, first_size);
                for (uint32_t i = lane; i < C; i += 64) if (c0 + i < n) put(c0 + i, sk[i]);
                __syncthreads();
            }
        };
        auto cmpx = [&](uint32_t lo, uint32_t hi) {
            if (hi < n) { const unsigned long long x = get(lo), y = get(hi); if (x > y) { put(lo, y); put(hi, x); } }
        };
        chunks(2);
        for (uint32_t size = 2 * C; size <= m; size <<= 1) {
            for (uint32_t t = lane; t < (m >> 1); t += 64) {
                const uint32_t half = size >> 1, blk = t / half, j = t - blk * half;
                cmpx(blk * size + j, blk * size + size - 1 - j);
            }
            __syncthreads();
            for (uint32_t stride = size >> 2; stride >= C; stride >>= 1) {
                for (uint32_t t = lane; t < (m >> 1); t += 64) { const uint32_t lo = 2 * t - (t & (stride - 1)); cmpx(lo, lo + stride); }
                __syncthreads();
            }
            chunks(0);
        }
    }
}

__global__ __launch_bounds__(256) void hop_rows_kernel(const unsigned long long* keys, const uint32_t* vals, const uint32_t* n_rows, uint32_t row_cap,
                                                       const gf_gap* gaps, gf_dpos* rows, uint32_t* row_gap) {
    const uint32_t n = *n_rows < row_cap ? *n_rows : row_cap;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const unsigned long long k = keys[r];
        const uint32_t g = vals[r];
        const gf_gap gp = gaps[g];
        gf_dpos d;
        d.mate_scaffold = (uint32_t)(k >> 32); d.mate_pos = (uint32_t)k; d.src_scaffold = gp.scaffold; d.src_gap = gp.idx_in_scaffold;
        rows[r] = d;
        row_gap[r] = g;
    }
}

// Union of several SORTED tables (one per source rank; gf_second_hop_table_dev leaves each sorted by (mate scaffold, mate position)):
// a row's place in the union is its index in its own part + the rows of the parts before it that are <= its key + the rows of the
// parts behind it that are < its key — a binary search per other part, no second sort.  (Round 3 re-sorted the padded union with a
// radix sort on every rank and step: 0.9 ms at C4 that did not shrink with the number of ranks.)
__device__ __forceinline__ unsigned long long dpos_key(const gf_dpos& d) { return ((unsigned long long)d.mate_scaffold << 32) | d.mate_pos; }
__global__ __launch_bounds__(256) void hop_merge_kernel(const gf_dpos* rows_all, const uint32_t* row_gap_all, const uint32_t* n_rows_all,
                                                        uint32_t n_parts, uint32_t part_cap, gf_dpos* rows, uint32_t* row_gap, uint32_t row_cap,
                                                        uint32_t* n_rows) {
    uint32_t total = 0;
    for (uint32_t p = 0; p < n_parts; ++p) total += n_rows_all[p] < part_cap ? n_rows_all[p] : part_cap;
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_rows = total;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        uint32_t p = 0, base = 0;
        for (;; ++p) {
            const uint32_t n = n_rows_all[p] < part_cap ? n_rows_all[p] : part_cap;
            if (i < base + n) break;
            base += n;
        }
        const gf_dpos d = rows_all[(uint64_t)p * part_cap + (i - base)];
        const unsigned long long key = dpos_key(d);
        uint32_t pos = i - base;
        for (uint32_t q = 0; q < n_parts; ++q) {
            if (q == p) continue;
            const gf_dpos* R = rows_all + (uint64_t)q * part_cap;
            uint32_t lo = 0, hi = n_rows_all[q] < part_cap ? n_rows_all[q] : part_cap;
            while (lo < hi) {   // q < p: rows <= key come first; q > p: rows < key
                const uint32_t mid = (lo + hi) >> 1;
                const unsigned long long km = dpos_key(R[mid]);
                if (q < p ? km <= key : km < key) lo = mid + 1; else hi = mid;
            }
            pos += lo;
        }
        if (pos < row_cap) { rows[pos] = d; row_gap[pos] = row_gap_all[(uint64_t)p * part_cap + (i - base)]; }
    }
}

// sorted rows -> upos (unique (scaffold, position) in order), urow (first row of each, + n at the end), soff (offsets of the
// scaffolds into upos).  Three small launches: firsts per block; block offsets + the entries; the scaffold offsets.  (One 1 024-thread
// workgroup did all of it in rounds 1-3: 0.28 ms at C4, replicated on every rank.)
constexpr uint32_t HOP_TB = 256;   // blocks of the table build
__device__ __forceinline__ bool hop_first(const gf_dpos* rows, uint32_t r) {
    return r == 0 || rows[r].mate_scaffold != rows[r - 1].mate_scaffold || rows[r].mate_pos != rows[r - 1].mate_pos;
}
__global__ __launch_bounds__(256) void hop_table_count_kernel(const gf_dpos* rows, const uint32_t* n_rows, uint32_t row_cap, uint32_t* blk_cnt, uint32_t* near_bits) {
    __shared__ uint32_t s_c;
    if (threadIdx.x == 0) s_c = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < (1u << (HOP_NEAR_LOG2 - 5)); i += HOP_TB * 256) near_bits[i] = 0;   // (set by the launch that follows)
    __syncthreads();
    const uint32_t n = *n_rows < row_cap ? *n_rows : row_cap;
    const uint32_t chunk = (n + HOP_TB - 1) / HOP_TB, a = blockIdx.x * chunk < n ? blockIdx.x * chunk : n, b = a + chunk < n ? a + chunk : n;
    uint32_t c = 0;
    for (uint32_t r = a + threadIdx.x; r < b; r += 256) c += hop_first(rows, r);
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_c, c);
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s_c;
}
__global__ __launch_bounds__(256) void hop_table_write_kernel(const gf_dpos* rows, const uint32_t* n_rows, uint32_t row_cap, const uint32_t* blk_cnt,
                                                              uint32_t* upos, uint32_t* urow, uint32_t* n_unique, uint32_t* near_bits) {
    __shared__ uint32_t s_w[4], s_base, s_run;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t n = *n_rows < row_cap ? *n_rows : row_cap;
    const uint32_t chunk = (n + HOP_TB - 1) / HOP_TB, a = blockIdx.x * chunk < n ? blockIdx.x * chunk : n, b = a + chunk < n ? a + chunk : n;
    uint32_t pre = 0, tot = 0;
    for (uint32_t q = tid; q < HOP_TB; q += 256) { const uint32_t v = blk_cnt[q]; tot += v; if (q < blockIdx.x) pre += v; }
    for (int d = 32; d >= 1; d >>= 1) { pre += __shfl_xor(pre, d); tot += __shfl_xor(tot, d); }
    if (tid == 0) { s_base = 0; s_run = 0; }
    __syncthreads();
    if (lane == 0) { atomicAdd(&s_base, pre); atomicAdd(&s_run, tot); }
    __syncthreads();
    const uint32_t nu = s_run;
    uint32_t run = s_base;
    __syncthreads();
    for (uint32_t r0 = a; r0 < b; r0 += 256) {
        const uint32_t r = r0 + tid;
        const bool f = r < b && hop_first(rows, r);
        const unsigned long long bal = __ballot(f);
        if (lane == 0) s_w[w] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t u = run + (uint32_t)__popcll(bal & ((1ull << lane) - 1));
        for (uint32_t q = 0; q < w; ++q) u += s_w[q];
        if (f) {
            const uint32_t q = rows[r].mate_pos, sc = rows[r].mate_scaffold;
            upos[u] = q; urow[u] = r;
            // a record at pos matches this row iff q - 199 <= pos <= q + 299 (collect_discordant_low_mapq_reads.py:62-71): its bins
            const uint32_t b0 = (q > 199 ? q - 199 : 0) >> HOP_NEAR_SHIFT, b1 = (uint32_t)(((uint64_t)q + 299 < 0xFFFFFFFFull ? (uint64_t)q + 299 : 0xFFFFFFFFull) >> HOP_NEAR_SHIFT);
            for (uint32_t bn = b0; bn <= b1; ++bn) { const uint32_t bit = hop_near_bit(sc, bn); atomicOr(&near_bits[bit >> 5], 1u << (bit & 31)); }
        }
        run += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    if (blockIdx.x == 0 && tid == 0) { urow[nu] = n; *n_unique = nu; }
}
__global__ __launch_bounds__(256) void hop_table_soff_kernel(const gf_dpos* rows, const uint32_t* urow, const uint32_t* n_unique, uint32_t n_scaffolds,
                                                             uint32_t* soff, uint32_t* n_out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = 0;   // the hit counter of the look-up that follows
    const uint32_t nu = *n_unique;
    // soff[s] = number of unique entries on scaffolds < s: lower bound over the rows' scaffold through urow
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s <= n_scaffolds; s += gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = nu;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (rows[urow[mid]].mate_scaffold < s) lo = mid + 1; else hi = mid;
        }
        soff[s] = lo;
    }
}

// tagger.hip; *d_n_out must be zero already
int launch_low_mapq_devtable(gf_ctx* ctx, const void* d_low, const void* d_n_low, size_t low_cap, const uint32_t* upos, const uint32_t* urow,
                             const uint32_t* soff, const uint32_t* near_bits, void* d_out, size_t cap, void* d_n_out);

}  // namespace gf

using namespace gf;

extern "C" {

int gf_second_hop_table_dev(gf_ctx* ctx, const void* d_recs, const void* d_taghits, const void* d_n_taghits, size_t hit_cap, void* d_rows,
                            void* d_row_gap, size_t row_cap, void* d_n_rows) {
    if (!ctx || !d_recs || !d_taghits || !d_n_taghits || !d_rows || !d_row_gap || !d_n_rows || row_cap == 0 || row_cap >= 0x7FFFFFFFull ||
        hit_cap > 0xFFFFFFFFull)
        return GF_E_INVAL;
    if (ctx->n_scaffolds == 0 || !ctx->d_gaps) return GF_E_STATE;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t scaf_bits = 0;   // (as hop_buckets)
    while (scaf_bits < 32 && (ctx->n_scaffolds >> scaf_bits) != 0) ++scaf_bits;
    const uint32_t sub_bits = scaf_bits < HOP_BUCKETS_LOG2 ? HOP_BUCKETS_LOG2 - scaf_bits : 0;
    const size_t nb_max = ((size_t)ctx->n_scaffolds + 1) << sub_bits;
    const size_t b_k = (row_cap * 8 + 255) & ~(size_t)255, b_v = (row_cap * 4 + 255) & ~(size_t)255, b_b = ((nb_max + 2) * 4 + 255) & ~(size_t)255;
    int rc;
    if ((rc = ensure(ctx, ctx->rowgap, 2 * b_k + 2 * b_v + 3 * b_b + 256))) return rc;
    ctx->rowgap_rows.clear();   // the buffer is sort scratch now: gf_pool_keys_from_tags_dev's cached row -> gap map is gone
    uint8_t* w = (uint8_t*)ctx->rowgap.p;
    unsigned long long* k_in = (unsigned long long*)w;
    unsigned long long* k_out = (unsigned long long*)(w + b_k);
    uint32_t* v_in = (uint32_t*)(w + 2 * b_k);
    uint32_t* v_out = (uint32_t*)(w + 2 * b_k + b_v);
    uint32_t* cnt = (uint32_t*)(w + 2 * b_k + 2 * b_v);          // [nb_max + 1] bucket counts, then one word: the largest mate position
    uint32_t* off = (uint32_t*)(w + 2 * b_k + 2 * b_v + b_b);
    uint32_t* cursor = (uint32_t*)(w + 2 * b_k + 2 * b_v + 2 * b_b);
    uint32_t* max_pos = cnt + nb_max + 1;
    LaunchTimer tm(ctx, GF_KERNEL_LOWMAPQ);
    const unsigned grid = (unsigned)std::min<size_t>((row_cap + 255) / 256, (size_t)ctx->n_cu * 4);
    zero_regions(ctx, ZeroList{{cnt, (uint32_t*)d_n_rows, nullptr, nullptr}, {(uint32_t)(nb_max + 2), 1, 0, 0}});
    hipLaunchKernelGGL(hop_extract_kernel, dim3(ctx->n_cu * 4), dim3(256), 0, ctx->stream, (const gf_alnrec*)d_recs, (const gf_taghit*)d_taghits,
                       (const uint32_t*)d_n_taghits, (uint32_t)hit_cap, ctx->n_scaffolds, k_in, v_in, (uint32_t)row_cap, (uint32_t*)d_n_rows, max_pos);
    hipLaunchKernelGGL(hop_bucket_count_kernel, dim3(grid), dim3(256), 0, ctx->stream, k_in, (const uint32_t*)d_n_rows, (uint32_t)row_cap, ctx->n_scaffolds,
                       max_pos, cnt);
    hipLaunchKernelGGL(hop_bucket_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, cnt, ctx->n_scaffolds, max_pos, off, cursor);
    hipLaunchKernelGGL(hop_bucket_scatter_kernel, dim3(grid), dim3(256), 0, ctx->stream, k_in, v_in, (const uint32_t*)d_n_rows, (uint32_t)row_cap,
                       ctx->n_scaffolds, max_pos, cursor, k_out, v_out);
    hipLaunchKernelGGL(hop_bucket_sort_kernel, dim3((unsigned)std::min<size_t>(nb_max, (size_t)ctx->n_cu * 16)), dim3(64), 0, ctx->stream, off, ctx->n_scaffolds,
                       max_pos, k_out, v_out);
    hipLaunchKernelGGL(hop_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream, k_out, v_out, (const uint32_t*)d_n_rows, (uint32_t)row_cap,
                       (const gf_gap*)ctx->d_gaps, (gf_dpos*)d_rows, (uint32_t*)d_row_gap);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_second_hop_table_merge_dev(gf_ctx* ctx, const void* d_rows_all, const void* d_row_gap_all, const void* d_n_rows_all, int n_parts,
                                  size_t part_cap, void* d_rows, void* d_row_gap, size_t row_cap, void* d_n_rows) {
    if (!ctx || !d_rows_all || !d_row_gap_all || !d_n_rows_all || !d_rows || !d_row_gap || !d_n_rows || n_parts < 1 || n_parts > 1024 ||
        part_cap == 0 || row_cap == 0 || row_cap >= 0x7FFFFFFFull || part_cap >= 0x7FFFFFFFull)
        return GF_E_INVAL;
    if (ctx->n_scaffolds == 0 || !ctx->d_gaps) return GF_E_STATE;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    LaunchTimer tm(ctx, GF_KERNEL_LOWMAPQ);
    const unsigned grid = (unsigned)std::min<size_t>(((size_t)n_parts * part_cap + 255) / 256, (size_t)ctx->n_cu * 8);
    hipLaunchKernelGGL(hop_merge_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const gf_dpos*)d_rows_all, (const uint32_t*)d_row_gap_all,
                       (const uint32_t*)d_n_rows_all, (uint32_t)n_parts, (uint32_t)part_cap, (gf_dpos*)d_rows, (uint32_t*)d_row_gap, (uint32_t)row_cap,
                       (uint32_t*)d_n_rows);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_tag_low_mapq_table_dev(gf_ctx* ctx, const void* d_low, const void* d_n_low, size_t low_cap, const void* d_rows, const void* d_n_rows,
                              size_t row_cap, void* d_out, size_t cap, void* d_n_out) {
    if (!ctx || !d_low || !d_n_low || !d_rows || !d_n_rows || !d_n_out || (cap && !d_out) || row_cap == 0 || row_cap >= 0x7FFFFFFFull ||
        low_cap > 0xFFFFFFFFull || cap > 0xFFFFFFFFull)
        return GF_E_INVAL;
    if (ctx->n_scaffolds == 0) return GF_E_STATE;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    const size_t b1 = (row_cap * 4 + 255) & ~(size_t)255, b2 = ((row_cap + 1) * 4 + 255) & ~(size_t)255, b3 = ((size_t)ctx->n_scaffolds + 1) * 4;
    int rc;
    const size_t b3a = (b3 + 255) & ~(size_t)255;
    const size_t b4 = (((size_t)HOP_TB + 1) * 4 + 255) & ~(size_t)255, b5 = (size_t)1 << (HOP_NEAR_LOG2 - 3);
    if ((rc = ensure(ctx, ctx->table, b1 + b2 + b3a + b4 + b5 + 64))) return rc;
    ctx->low_rows.clear();   // the cached host-table copy in ctx->table is gone
    uint8_t* base = (uint8_t*)ctx->table.p;
    uint32_t* upos = (uint32_t*)base;
    uint32_t* urow = (uint32_t*)(base + b1);
    uint32_t* soff = (uint32_t*)(base + b1 + b2);
    uint32_t* blk_cnt = (uint32_t*)(base + b1 + b2 + b3a);   // [HOP_TB] firsts per block, [HOP_TB] unique positions in all
    uint32_t* near_bits = (uint32_t*)(base + b1 + b2 + b3a + b4);   // 2^HOP_NEAR_LOG2 bits: the positions a row lies near, hashed
    {
        LaunchTimer tm(ctx, GF_KERNEL_LOWMAPQ);
        hipLaunchKernelGGL(hop_table_count_kernel, dim3(HOP_TB), dim3(256), 0, ctx->stream, (const gf_dpos*)d_rows, (const uint32_t*)d_n_rows, (uint32_t)row_cap, blk_cnt, near_bits);
        hipLaunchKernelGGL(hop_table_write_kernel, dim3(HOP_TB), dim3(256), 0, ctx->stream, (const gf_dpos*)d_rows, (const uint32_t*)d_n_rows, (uint32_t)row_cap,
                           blk_cnt, upos, urow, blk_cnt + HOP_TB, near_bits);
        hipLaunchKernelGGL(hop_table_soff_kernel, dim3((ctx->n_scaffolds + 256) / 256), dim3(256), 0, ctx->stream, (const gf_dpos*)d_rows, urow, blk_cnt + HOP_TB,
                           ctx->n_scaffolds, soff, (uint32_t*)d_n_out);
    }
    return launch_low_mapq_devtable(ctx, d_low, d_n_low, low_cap, upos, urow, soff, near_bits, d_out, cap, d_n_out);
}

}  // extern "C"
