// Second-hop table on the device.  The reference inverts every `discordant` list line into "mateScaffold matePos srcScaffold
// srcGap", runs sort(1) -k1n -k2n -k3n -k4n over the file and splits it per mate scaffold (run_multi_threads_discordant.py:19-122);
// collect_discordant_low_mapq_reads.py:4-28 then looks MAPQ-0 records up in it.  Here the rows are cut from the tagger's hits
// where they are (HBM), sorted with one rocPRIM radix sort, and the look-up arrays of the second-hop kernel (unique positions per
// scaffold + row ranges) are derived from them by one small kernel — no host copy, no host sort, nothing cached between batches.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "gf_internal.hpp"

namespace gf {

__global__ __launch_bounds__(256) void hop_fill_kernel(unsigned long long* keys, uint32_t n, unsigned long long v, uint32_t* n_rows) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_rows = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) keys[i] = v;
}

// one row per DISCORDANT hit whose mate lies on a known scaffold: key = mate scaffold << 32 | mate position, value = gap index
__global__ __launch_bounds__(256) void hop_extract_kernel(const gf_alnrec* recs, const gf_taghit* hits, const uint32_t* n_hits, uint32_t hit_cap,
                                                          uint32_t n_scaffolds, unsigned long long* keys, uint32_t* vals, uint32_t row_cap,
                                                          uint32_t* n_rows) {
    const uint32_t n = *n_hits < hit_cap ? *n_hits : hit_cap;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n_round = (n + 63) & ~63u;
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
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(n_rows, (uint32_t)__popcll(bal));   // may exceed row_cap: consumers clamp, the host variant reports it
        base = __shfl(base, 0);
        const uint32_t o = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1));
        if (take && o < row_cap) { keys[o] = key; vals[o] = gap; }
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
__global__ __launch_bounds__(256) void hop_table_count_kernel(const gf_dpos* rows, const uint32_t* n_rows, uint32_t row_cap, uint32_t* blk_cnt) {
    __shared__ uint32_t s_c;
    if (threadIdx.x == 0) s_c = 0;
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
                                                              uint32_t* upos, uint32_t* urow, uint32_t* n_unique) {
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
        if (f) { upos[u] = rows[r].mate_pos; urow[u] = r; }
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
                             const uint32_t* soff, void* d_out, size_t cap, void* d_n_out);

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
    unsigned end_bit = 33;   // position bits + enough scaffold bits to order the sentinel n_scaffolds << 32 last
    while (end_bit < 64 && (ctx->n_scaffolds >> (end_bit - 32)) != 0) ++end_bit;
    size_t temp_bytes = 0;
    unsigned long long* nullk = nullptr;
    uint32_t* nullv = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, temp_bytes, nullk, nullk, nullv, nullv, row_cap, 0, end_bit, ctx->stream) != hipSuccess) return GF_E_NODEV;
    const size_t b_k = (row_cap * 8 + 255) & ~(size_t)255, b_v = (row_cap * 4 + 255) & ~(size_t)255;
    int rc;
    if ((rc = ensure(ctx, ctx->rowgap, 2 * b_k + 2 * b_v + temp_bytes + 256))) return rc;
    ctx->rowgap_rows.clear();   // the buffer is sort scratch now: gf_pool_keys_from_tags_dev's cached row -> gap map is gone
    uint8_t* w = (uint8_t*)ctx->rowgap.p;
    unsigned long long* k_in = (unsigned long long*)w;
    unsigned long long* k_out = (unsigned long long*)(w + b_k);
    uint32_t* v_in = (uint32_t*)(w + 2 * b_k);
    uint32_t* v_out = (uint32_t*)(w + 2 * b_k + b_v);
    void* temp = w + 2 * b_k + 2 * b_v;
    LaunchTimer tm(ctx, GF_KERNEL_LOWMAPQ);
    const unsigned grid = (unsigned)std::min<size_t>((row_cap + 255) / 256, (size_t)ctx->n_cu * 4);
    hipLaunchKernelGGL(hop_fill_kernel, dim3(grid), dim3(256), 0, ctx->stream, k_in, (uint32_t)row_cap, (unsigned long long)ctx->n_scaffolds << 32,
                       (uint32_t*)d_n_rows);
    hipLaunchKernelGGL(hop_extract_kernel, dim3(ctx->n_cu * 4), dim3(256), 0, ctx->stream, (const gf_alnrec*)d_recs, (const gf_taghit*)d_taghits,
                       (const uint32_t*)d_n_taghits, (uint32_t)hit_cap, ctx->n_scaffolds, k_in, v_in, (uint32_t)row_cap, (uint32_t*)d_n_rows);
    if (rocprim::radix_sort_pairs(temp, temp_bytes, k_in, k_out, v_in, v_out, row_cap, 0, end_bit, ctx->stream) != hipSuccess) return GF_E_NODEV;
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
    if ((rc = ensure(ctx, ctx->table, b1 + b2 + b3a + (HOP_TB + 1) * 4 + 64))) return rc;
    ctx->low_rows.clear();   // the cached host-table copy in ctx->table is gone
    uint8_t* base = (uint8_t*)ctx->table.p;
    uint32_t* upos = (uint32_t*)base;
    uint32_t* urow = (uint32_t*)(base + b1);
    uint32_t* soff = (uint32_t*)(base + b1 + b2);
    uint32_t* blk_cnt = (uint32_t*)(base + b1 + b2 + b3a);   // [HOP_TB] firsts per block, [HOP_TB] unique positions in all
    {
        LaunchTimer tm(ctx, GF_KERNEL_LOWMAPQ);
        hipLaunchKernelGGL(hop_table_count_kernel, dim3(HOP_TB), dim3(256), 0, ctx->stream, (const gf_dpos*)d_rows, (const uint32_t*)d_n_rows, (uint32_t)row_cap, blk_cnt);
        hipLaunchKernelGGL(hop_table_write_kernel, dim3(HOP_TB), dim3(256), 0, ctx->stream, (const gf_dpos*)d_rows, (const uint32_t*)d_n_rows, (uint32_t)row_cap,
                           blk_cnt, upos, urow, blk_cnt + HOP_TB);
        hipLaunchKernelGGL(hop_table_soff_kernel, dim3((ctx->n_scaffolds + 256) / 256), dim3(256), 0, ctx->stream, (const gf_dpos*)d_rows, urow, blk_cnt + HOP_TB,
                           ctx->n_scaffolds, soff, (uint32_t*)d_n_out);
    }
    return launch_low_mapq_devtable(ctx, d_low, d_n_low, low_cap, upos, urow, soff, d_out, cap, d_n_out);
}

}  // extern "C"
