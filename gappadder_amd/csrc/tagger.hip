// tagger.hip — alignment-record tagging.
//
// tag_kernel restates GapReadsCollector.parse_reads_fall_in_gaps_one_scaffold / _short_is
// (collect_reads_for_gaps.py:68-163 / :166-263) on decoded 32-byte records; the reference's per-position dict
// `focal_region` (get_focal_region_of_scaffold_v2, :31-65) becomes a closed-form window test per gap:
//   left  window: 0 <= start-POS < dist2  (keys start-i, i in range(dist2), start-i >= 0; 'c' if i <= clip_dist)
//   right window: 0 <= POS-end   < dist2  (keys end+i)
// low_mapq_kernel restates collect_discordant_low_mapq_reads.py:4-84: MAPQ==0 records, focal_region[p] = the
// LAST discordant mate position q (sorted file order) with q-199 <= p <= q+299, one hit per row of q.
#include "gf_internal.hpp"

namespace gf {

struct TagParams {
    const gf_alnrec* recs;
    uint64_t n;
    const gf_gap* gaps;
    const uint32_t* scaf_off;
    uint32_t n_scaffolds;
    int32_t dist1, dist2, clip_dist, anchor_mapq;
    int32_t short_is;
    gf_taghit* out;
    uint32_t cap;
    uint32_t* n_out;
};

__device__ __forceinline__ void emit_hit(bool want, const gf_taghit& h, gf_taghit* out, uint32_t cap, uint32_t* n_out) {
    const unsigned long long bal = __ballot(want);
    if (!bal) return;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t base = 0;
    const uint32_t leader = __ffsll((long long)bal) - 1;
    if (lane == leader) base = atomicAdd(n_out, (uint32_t)__popcll(bal));
    base = __shfl(base, leader);
    if (want) {
        const uint32_t o = base + __popcll(bal & ((1ull << lane) - 1));
        if (o < cap) out[o] = h;
    }
}

__global__ __launch_bounds__(256) void tag_kernel(TagParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    // every lane of a wave must reach the ballots together: iterate whole waves
    const uint64_t n_round = (P.n + 63) & ~63ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        bool live = i < P.n;
        gf_alnrec r = {};
        uint32_t g = 0, g_end = 0;
        if (live) {
            const uint4* p = reinterpret_cast<const uint4*>(P.recs + i);
            const uint4 a = p[0], b = p[1];
            r.pos = a.x; r.mate_pos = a.y; r.tlen = (int32_t)a.z; r.ref = a.w;
            r.mate_ref = b.x; r.flag = (uint16_t)(b.y & 0xFFFF); r.mapq = (uint8_t)((b.y >> 16) & 0xFF);
            r.clipflag = (uint8_t)(b.y >> 24);
            live = r.ref < P.n_scaffolds;
            if (live) {
                g = P.scaf_off[r.ref];
                g_end = P.scaf_off[r.ref + 1];
                // first gap whose right window can still reach POS: end + dist2 > pos  (ends ascend)
                uint32_t lo = g, hi = g_end;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if ((int64_t)P.gaps[mid].end + P.dist2 > (int64_t)r.pos) hi = mid; else lo = mid + 1;
                }
                g = lo;
            }
        }
        const int64_t pos = r.pos;
        // walk the (few) gaps whose windows can contain POS; lanes without work idle through the ballots
        while (true) {
            bool more = live && g < g_end && (int64_t)P.gaps[g].start - P.dist2 < pos;
            if (!__any(more)) break;
            bool clip = false, pair = false, unmap = false;
            if (more) {
                const gf_gap gp = P.gaps[g];
                const int64_t il = (int64_t)gp.start - pos, ir = pos - (int64_t)gp.end;
                int tag = -1;  // 0: 0c, 1: 0d, 2: 1c, 3: 1d
                if (il >= 0 && il < P.dist2) tag = il <= P.clip_dist ? 0 : 1;
                else if (ir >= 0 && ir < P.dist2) tag = ir <= P.clip_dist ? 2 : 3;
                if (tag >= 0) {
                    clip = (tag == 0 && r.clipflag >= 2) || (tag == 2 && (r.clipflag == 1 || r.clipflag == 3));
                    const bool mapped = (r.flag & 0x4) == 0, mate_mapped = (r.flag & 0x8) == 0;
                    if (mapped && mate_mapped && (int)r.mapq >= P.anchor_mapq) {
                        if (r.mate_ref != r.ref) pair = true;
                        else {
                            const int64_t t = r.tlen < 0 ? -(int64_t)r.tlen : (int64_t)r.tlen;
                            pair = t >= P.dist2 || (P.short_is && t <= P.dist1);
                        }
                    } else if (mapped && !mate_mapped) {
                        unmap = true;
                    }
                }
            }
            gf_taghit h;
            h.rec = (uint32_t)i; h.gap = g;
            h.kind = GF_KIND_CLIP; h.to_mate = 0;
            emit_hit(clip, h, P.out, P.cap, P.n_out);
            h.kind = pair ? GF_KIND_DISCORDANT : GF_KIND_UNMAP; h.to_mate = 1;
            emit_hit(pair || unmap, h, P.out, P.cap, P.n_out);
            if (more) ++g;
        }
    }
}

struct LowParams {
    const gf_alnrec* recs;
    uint64_t n;
    const uint32_t* upos;      // unique (scaffold-grouped) mate positions, ascending inside a scaffold
    const uint32_t* urow;      // n_unique+1 offsets into the row table
    const uint32_t* scaf_off;  // n_scaffolds+1 offsets into upos
    uint32_t n_scaffolds;
    gf_taghit* out;
    uint32_t cap;
    uint32_t* n_out;
};

__global__ __launch_bounds__(256) void low_mapq_kernel(LowParams P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n_round = (P.n + 63) & ~63ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        uint32_t row = 0, row_end = 0;
        if (i < P.n) {
            const uint4* p = reinterpret_cast<const uint4*>(P.recs + i);
            const uint4 a = p[0], b = p[1];
            const uint32_t pos = a.x, ref = a.w, mapq = (b.y >> 16) & 0xFF;
            if (mapq == 0 && ref < P.n_scaffolds) {
                uint32_t lo = P.scaf_off[ref], hi = P.scaf_off[ref + 1];
                const uint32_t first = lo;
                const uint64_t lim = (uint64_t)pos + 199;  // largest q <= pos+199
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if ((uint64_t)P.upos[mid] <= lim) lo = mid + 1; else hi = mid;
                }
                if (lo > first) {
                    const uint32_t u = lo - 1;
                    if ((uint64_t)P.upos[u] + 299 >= pos) {
                        row = P.urow[u];
                        row_end = P.urow[u + 1];
                    }
                }
            }
        }
        while (true) {
            const bool more = row < row_end;
            if (!__any(more)) break;
            gf_taghit h;
            h.rec = (uint32_t)i; h.gap = row; h.kind = GF_KIND_LOWMAPQ; h.to_mate = 0;
            emit_hit(more, h, P.out, P.cap, P.n_out);
            if (more) ++row;
        }
    }
}

static unsigned stream_grid(gf_ctx* ctx, size_t n) {
    size_t blocks = (n + 255) / 256;
    return (unsigned)std::max<size_t>(1, std::min<size_t>(blocks, (size_t)ctx->n_cu * 8));
}

int launch_tag(gf_ctx* ctx, const void* d_recs, size_t n, int insert_size, int sd, int clip_dist, int anchor_mapq,
               void* d_out, size_t cap, void* d_n_out) {
    if (!ctx->d_gaps) return GF_E_STATE;
    if (n >= 0xFFFFFFFFull || cap > 0xFFFFFFFFull) return GF_E_INVAL;
    GF_HIP(ctx, hipMemsetAsync(d_n_out, 0, 4, ctx->stream));
    if (n == 0) return GF_OK;
    TagParams P;
    P.recs = (const gf_alnrec*)d_recs;
    P.n = n;
    P.gaps = ctx->d_gaps;
    P.scaf_off = ctx->d_scaf_off;
    P.n_scaffolds = ctx->n_scaffolds;
    P.dist1 = insert_size - 3 * sd;
    P.dist2 = insert_size + 3 * sd;
    P.clip_dist = clip_dist;
    P.anchor_mapq = anchor_mapq;
    P.short_is = insert_size < 750;  // collect_reads_for_gaps.py:275
    P.out = (gf_taghit*)d_out;
    P.cap = (uint32_t)cap;
    P.n_out = (uint32_t*)d_n_out;
    {
        LaunchTimer tm(ctx, GF_KERNEL_TAG);
        hipLaunchKernelGGL(tag_kernel, dim3(stream_grid(ctx, n)), dim3(256), 0, ctx->stream, P);
    }
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int launch_low_mapq(gf_ctx* ctx, const void* d_recs, size_t n, const gf_dpos* table, size_t n_rows, void* d_out,
                    size_t cap, void* d_n_out) {
    if (n >= 0xFFFFFFFFull || cap > 0xFFFFFFFFull || n_rows >= 0xFFFFFFFFull) return GF_E_INVAL;
    if (ctx->n_scaffolds == 0) return GF_E_STATE;
    GF_HIP(ctx, hipMemsetAsync(d_n_out, 0, 4, ctx->stream));
    if (n == 0 || n_rows == 0) return GF_OK;
    // host: unique positions per scaffold + row offsets (the table is tiny next to the record stream)
    std::vector<uint32_t> upos, urow, soff(ctx->n_scaffolds + 1, 0);
    uint32_t cur_s = 0;
    for (size_t r = 0; r < n_rows; ++r) {
        if (r) {
            const gf_dpos &a = table[r - 1], &b = table[r];
            if (a.mate_scaffold > b.mate_scaffold || (a.mate_scaffold == b.mate_scaffold && a.mate_pos > b.mate_pos))
                return GF_E_INVAL;  // must be sorted (run_multi_threads_discordant.py:103)
        }
        if (table[r].mate_scaffold >= ctx->n_scaffolds) return GF_E_INVAL;
        if (r == 0 || table[r].mate_scaffold != table[r - 1].mate_scaffold || table[r].mate_pos != table[r - 1].mate_pos) {
            while (cur_s < table[r].mate_scaffold) soff[++cur_s] = (uint32_t)upos.size();
            upos.push_back(table[r].mate_pos);
            urow.push_back((uint32_t)r);
        }
    }
    urow.push_back((uint32_t)n_rows);
    while (cur_s < ctx->n_scaffolds) soff[++cur_s] = (uint32_t)upos.size();
    const size_t b1 = upos.size() * 4, b2 = urow.size() * 4, b3 = soff.size() * 4;
    int rc;
    if ((rc = ensure(ctx, ctx->table, b1 + b2 + b3 + 64))) return rc;
    uint8_t* base = (uint8_t*)ctx->table.p;
    GF_HIP(ctx, hipMemcpyAsync(base, upos.data(), b1, hipMemcpyHostToDevice, ctx->stream));
    GF_HIP(ctx, hipMemcpyAsync(base + b1, urow.data(), b2, hipMemcpyHostToDevice, ctx->stream));
    GF_HIP(ctx, hipMemcpyAsync(base + b1 + b2, soff.data(), b3, hipMemcpyHostToDevice, ctx->stream));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));  // host vectors go out of scope
    LowParams P;
    P.recs = (const gf_alnrec*)d_recs;
    P.n = n;
    P.upos = (const uint32_t*)base;
    P.urow = (const uint32_t*)(base + b1);
    P.scaf_off = (const uint32_t*)(base + b1 + b2);
    P.n_scaffolds = ctx->n_scaffolds;
    P.out = (gf_taghit*)d_out;
    P.cap = (uint32_t)cap;
    P.n_out = (uint32_t*)d_n_out;
    {
        LaunchTimer tm(ctx, GF_KERNEL_LOWMAPQ);
        hipLaunchKernelGGL(low_mapq_kernel, dim3(stream_grid(ctx, n)), dim3(256), 0, ctx->stream, P);
    }
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

}  // namespace gf
