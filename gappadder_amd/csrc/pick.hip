// pick.hip — flank anchoring on the device: which gaps have a contig that both flanks anchor on the same
// strand, and how long the sequence between the anchors is.  This is the test `ContigsSelection` applies after every assembly
// round (pick_contigs.py:64-358: `bwa mem -T {score} -a` of the two flanks against the gap's contigs; per contig the best
// same-strand pair of a left and a right hit :149-297, over the contigs the longest span :300-321) with bwa replaced by EXACT
// anchors: the last `anchor_len` bases of the left flank and the first `anchor_len` bases of the right flank (anchor_len = the
// reference's bwa_min_score, 30 then 15: assemble_gaps.py:336, 365).  It makes "gaps closed" a quantity of the step instead of a
// host loop over all contigs.  Definition and oracle: oracle/gp_oracle.py::pick_gap (the reference's selection, pinned on its own
// answers, applied to the stand-in's hits); host twin: gappadder_amd/pick_contigs.py.
//
// Per contig: forward hits = leftmost occurrence of the left anchor, rightmost occurrence of the right anchor; reverse hits = the
// same in the reverse-complemented contig, searched as the reverse-complemented anchors in the forward contig.  The reference
// keeps one hit per (side, clip type), the first one on equal match lengths: a flank LONGER than the anchor is clipped in front
// on one strand and behind on the other, so both strands' hits survive; a flank exactly as long as the anchor gives unclipped
// hits on both strands and only the forward one (the first reported) survives.  Of the surviving pairs the forward pair is tried
// first and a later pair must match MORE bases to replace it (:173-291) — so: the forward pair when it exists, else the reverse
// pair; span = bases between the anchors, >= 0 or the contig does not count (:313-321).  Per gap: the longest span, the
// earlier contig on ties; a pick at a longer anchor outranks any pick at a shorter one (the pipeline only picks at 15 what
// 30 left open, assemble_gaps.py:336-366).  gap_best[g] = anchor_len << 56 | (span + 1) << 32 | (0x7FFFFFFF - contig) << 1 |
// reverse; 0 = no contig anchored = gap not closed.
#include <cstring>

#include "gf_internal.hpp"

namespace gf {

constexpr int ANCHOR_MAX = 32;

constexpr int ANCHOR_ROW = 5 * ANCHOR_MAX;

// anchors: per gap 5 x 32 bytes: left, right, revcomp(left), revcomp(right), each `a` bases from byte 0 (byte 0 == 0: none), and a
// flags row: byte 0 bit 0 = the left flank is exactly `a` bases long (unclipped hits), bit 1 = the right flank is
__global__ __launch_bounds__(256) void pick_anchor_kernel(const gf_contig* contigs, const uint32_t* n_contigs, uint32_t contig_cap,
                                                          const char* seq, const uint8_t* anchors, uint32_t n_gaps, uint32_t a,
                                                          unsigned long long* gap_best, uint32_t* n_closed) {
    const uint32_t n = *n_contigs < contig_cap ? *n_contigs : contig_cap;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t ci = wave; ci < n; ci += n_waves) {
        const gf_contig c = contigs[ci];
        if (c.gap >= n_gaps || c.length < 2 * a) continue;
        const uint8_t* an = anchors + (uint64_t)c.gap * ANCHOR_ROW;
        if (an[0] == 0 || an[ANCHOR_MAX] == 0) continue;
        const char* s = seq + c.seq_off;
        // positions where each of the four patterns occurs: min and max per pattern
        uint32_t mn[4] = {EMPTY32, EMPTY32, EMPTY32, EMPTY32}, mx[4] = {0, 0, 0, 0};
        bool any[4] = {false, false, false, false};
        const uint32_t last = c.length - a;
        for (uint32_t p = lane; p <= last; p += 64) {
            const uint8_t b0 = (uint8_t)s[p];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint8_t* pat = an + q * ANCHOR_MAX;
                if (pat[0] != b0) continue;
                uint32_t i = 1;
                while (i < a && (uint8_t)s[p + i] == pat[i]) ++i;
                if (i == a) {
                    any[q] = true;
                    mn[q] = p < mn[q] ? p : mn[q];
                    mx[q] = p > mx[q] ? p : mx[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // wave reductions
            for (int d = 32; d >= 1; d >>= 1) {
                const uint32_t m2 = __shfl_xor(mn[q], d), x2 = __shfl_xor(mx[q], d);
                mn[q] = m2 < mn[q] ? m2 : mn[q];
                mx[q] = x2 > mx[q] ? x2 : mx[q];
            }
            any[q] = __ballot(any[q]) != 0;
        }
        if (lane != 0) continue;
        // forward: left anchor at mn[0], right anchor at mx[1];  reverse: revcomp(right) at mn[3], revcomp(left) at mx[2] (the oriented
        // contig is the reverse complement).  An unclipped flank's reverse hit is hidden by its forward hit.
        const uint32_t fl = an[4 * ANCHOR_MAX];
        const bool lf = any[0], rf = any[1], lr = any[2] && !((fl & 1) && any[0]), rr = any[3] && !((fl & 2) && any[1]);
        uint32_t best = 0, orient = 0;
        if (lf && rf) {
            if (mx[1] >= mn[0] + a) best = mx[1] - (mn[0] + a) + 1;
        } else if (lr && rr) {
            if (mx[2] >= mn[3] + a) { best = mx[2] - (mn[3] + a) + 1; orient = 1; }
        }
        if (!best) continue;
        const unsigned long long val = ((unsigned long long)a << 56) | ((unsigned long long)(best & 0xFFFFFFu) << 32) |
                                       ((unsigned long long)(0x7FFFFFFFu - ci) << 1) | orient;
        const unsigned long long old = atomicMax(gap_best + c.gap, val);
        if (old == 0) atomicAdd(n_closed, 1u);
    }
}

}  // namespace gf

using namespace gf;

extern "C" {

int gf_pick_anchored_dev(gf_ctx* ctx, const void* d_contigs, const void* d_n_contigs, size_t contig_cap, const void* d_seq,
                         int anchor_len, void* d_gap_best, void* d_n_closed) {
    if (!ctx || !d_contigs || !d_n_contigs || !d_seq || !d_gap_best || !d_n_closed || anchor_len < 8 || anchor_len > ANCHOR_MAX ||
        contig_cap > 0xFFFFFFFFull)
        return GF_E_INVAL;
    const size_t ng = ctx->gaps.size();
    if (ctx->flank_left.size() != ng || ctx->flank_right.size() != ng) return GF_E_STATE;
    if (!ng) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf& tab = ctx->anchor_tabs[anchor_len];
    if (!tab.p) {   // built once per anchor length (gf_set_gaps drops the tables)
        std::vector<uint8_t> h(ng * ANCHOR_ROW, 0);
        auto acgt = [](char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; };
        auto comp = [](char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; };
        for (size_t g = 0; g < ng; ++g) {
            const std::string &l = ctx->flank_left[g], &r = ctx->flank_right[g];
            if ((int)l.size() < anchor_len || (int)r.size() < anchor_len) continue;
            const char* la = l.data() + l.size() - anchor_len;
            const char* ra = r.data();
            bool ok = true;
            for (int i = 0; i < anchor_len; ++i) ok = ok && acgt(la[i]) && acgt(ra[i]);
            if (!ok) continue;
            uint8_t* o = h.data() + g * ANCHOR_ROW;
            o[4 * ANCHOR_MAX] = (uint8_t)(((int)l.size() == anchor_len ? 1 : 0) | ((int)r.size() == anchor_len ? 2 : 0));
            for (int i = 0; i < anchor_len; ++i) {
                o[i] = (uint8_t)la[i];
                o[ANCHOR_MAX + i] = (uint8_t)ra[i];
                o[2 * ANCHOR_MAX + i] = (uint8_t)comp(la[anchor_len - 1 - i]);
                o[3 * ANCHOR_MAX + i] = (uint8_t)comp(ra[anchor_len - 1 - i]);
            }
        }
        int rc = ensure(ctx, tab, h.size() + 64);
        if (rc) return rc;
        GF_HIP(ctx, hipMemcpy(tab.p, h.data(), h.size(), hipMemcpyHostToDevice));
    }
    LaunchTimer tm(ctx, GF_KERNEL_PICK);
    hipLaunchKernelGGL(pick_anchor_kernel, dim3(ctx->n_cu * 4), dim3(256), 0, ctx->stream, (const gf_contig*)d_contigs,
                       (const uint32_t*)d_n_contigs, (uint32_t)contig_cap, (const char*)d_seq, (const uint8_t*)tab.p,
                       (uint32_t)ng, (uint32_t)anchor_len, (unsigned long long*)d_gap_best, (uint32_t*)d_n_closed);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

}  // extern "C"
