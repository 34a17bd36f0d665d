// synth.hip — device generator of the seeded synthetic workload defined in include/gf_synth.h (bench/test
// utility: fills HBM with packed reads + alignment records so that the timed region starts with inputs resident).
#include <cstring>

#include "../../include/gf_synth.h"
#include "gf_internal.hpp"

namespace gf {

__global__ __launch_bounds__(256) void synth_kernel(gf_synth_cfg c, uint64_t first_pair, uint64_t n_pairs, uint8_t* packed,
                                                    uint32_t* recs) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n_pairs) return;
    const uint64_t lp = t >> 1;  // local pair
    const int end = (int)(t & 1);
    gfs_pair p;
    gfs_make_pair(&c, first_pair + lp, &p);
    gfs_errs er;
    gfs_make_errs(&c, p.err[end], &er);
    const uint32_t mate_no = ((end == 0) != (p.flip != 0)) ? 0u : 1u;
    const uint32_t rb = (c.read_len + 3) / 4;
    uint8_t* o = packed + (2 * lp + mate_no) * (uint64_t)rb;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < c.read_len; ++i) {
        acc = (acc << 2) | gfs_read_base(&c, &p, &er, end, i);
        if ((i & 3) == 3) { o[i >> 2] = (uint8_t)acc; acc = 0; }
    }
    if (c.read_len & 3) o[rb - 1] = (uint8_t)(acc << (2 * (4 - (c.read_len & 3))));
    if (recs) {
        uint32_t r[2][8];
        gfs_make_records(&c, lp, &p, r);   // read id = index in THIS batch (what the pool builder gathers by)
        uint4* dst = reinterpret_cast<uint4*>(recs + (2 * lp + end) * 8);
        dst[0] = make_uint4(r[end][0], r[end][1], r[end][2], r[end][3]);
        dst[1] = make_uint4(r[end][4], r[end][5], r[end][6], r[end][7]);
    }
}

}  // namespace gf

using namespace gf;

extern "C" {

static int synth_check(const gf_synth_cfg* c) {
    if (!c || c->read_len < 16 || c->read_len > 256 || c->n_scaffolds == 0 || c->scaffold_len < 4ull * c->read_len) return GF_E_INVAL;
    if (c->gaps_per_scaffold && (uint64_t)c->gap_len + 4ull * c->flank_len + 2ull * c->read_len >= c->scaffold_len / (c->gaps_per_scaffold + 1))
        return GF_E_INVAL;
    if ((uint64_t)c->insert_mean + 8ull * c->insert_sd + c->read_len >= c->scaffold_len) return GF_E_INVAL;
    if (c->repeats) {   // planted repeats: period >= 4, a copy count, and gaps far enough apart that two gaps' planted intervals never meet
        if ((c->repeats & 0xFFu) < 4 || ((c->repeats >> 8) & 0xFFu) < 1 || (c->repeats >> 16) || c->gaps_per_scaffold == 0) return GF_E_INVAL;
        if ((uint64_t)c->gap_len + 2ull * (GFS_REP_MAXLEN + 400) + 2ull * c->read_len >= c->scaffold_len / (c->gaps_per_scaffold + 1)) return GF_E_INVAL;
    }
    return GF_OK;
}

int gf_synth_pairs_dev(gf_ctx* ctx, const void* cfg_, uint64_t first_pair, size_t n_pairs, void* d_packed,
                       void* d_recs_or_null) {
    if (!ctx || !d_packed) return GF_E_INVAL;
    const gf_synth_cfg* cfg = (const gf_synth_cfg*)cfg_;
    int rc = synth_check(cfg);
    if (rc) return rc;
    if (n_pairs == 0) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    const size_t threads = 2 * n_pairs;
    {
        LaunchTimer tm(ctx, GF_KERNEL_SYNTH);
        hipLaunchKernelGGL(synth_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, *cfg, first_pair,
                           (uint64_t)n_pairs, (uint8_t*)d_packed, (uint32_t*)d_recs_or_null);
    }
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

// host: gap table + flank sequences of the synthetic draft (gnrt_pos_true_seqs.py:94-99 flank rule)
int gf_synth_layout(const void* c_, gf_gap* gaps, char* flank_ascii, uint64_t* flank_off) {
    const gf_synth_cfg* c = (const gf_synth_cfg*)c_;
    int rc = synth_check(c);
    if (rc) return rc;
    if (!gaps || !flank_ascii || !flank_off) return GF_E_INVAL;
    uint64_t off = 0;
    size_t g = 0;
    for (uint32_t s = 0; s < c->n_scaffolds; ++s)
        for (uint32_t j = 0; j < c->gaps_per_scaffold; ++j, ++g) {
            const uint64_t st = gfs_gap_start(c, j), en = st + c->gap_len;
            gaps[g].scaffold = s; gaps[g].start = (uint32_t)st; gaps[g].end = (uint32_t)en; gaps[g].idx_in_scaffold = j + 1;
            flank_off[2 * g] = off;
            for (uint64_t x = st - c->flank_len; x < st - 5; ++x) flank_ascii[off++] = "ACGT"[gfs_base(c, s, x)];
            flank_off[2 * g + 1] = off;
            for (uint64_t x = en + 5; x < en + c->flank_len; ++x) flank_ascii[off++] = "ACGT"[gfs_base(c, s, x)];
        }
    flank_off[2 * g] = off;
    return GF_OK;
}

// host: the TRUE bases (what lies behind the planted N-runs too) of [start, start + n) of one scaffold — ground truth for the
// closed-gap check (the reference evaluates filled gaps against the true sequences the same way, validate_gap_seqs.py:5-75)
int gf_synth_truth(const void* c_, uint32_t scaffold, uint64_t start, size_t n, char* out_ascii) {
    const gf_synth_cfg* c = (const gf_synth_cfg*)c_;
    int rc = synth_check(c);
    if (rc) return rc;
    if (!out_ascii || scaffold >= c->n_scaffolds || start + n > c->scaffold_len) return GF_E_INVAL;
    for (size_t i = 0; i < n; ++i) out_ascii[i] = "ACGT"[gfs_base(c, scaffold, start + i)];
    return GF_OK;
}

}  // extern "C"
