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
// flags row: byte 0 bit 0 = the left flank is exactly `a` bases long (unclipped hits), bit 1 = the right flank is.
//
// One wave per contig, ONE pass over it for up to two anchor lengths (a_long > a_short: the short anchors are the inner ends of the
// long ones, so a long hit is a short hit that extends): a lane loads four bases at its position as one dword and compares them
// with the heads of the four SHORT patterns held in registers (1.5 % of the positions go on to the byte loop); a short hit updates
// the short pattern's leftmost / rightmost position and, when the long pattern's other a_long - a_short bases match as well, the
// long one's.  (Before: one launch per anchor length, four pattern-byte loads per position.)
struct PickParams {
    const gf_contig* contigs;
    const uint32_t* n_contigs;
    uint32_t contig_cap;
    const char* seq;
    const uint8_t* anc_s;      // table of the short (or only) anchor length
    const uint8_t* anc_l;      // table of the long anchor length, or null
    uint32_t n_gaps, a_s, a_l;
    unsigned long long* gap_best;
    uint32_t* n_closed;
    const uint32_t* first;     // or null: only the contigs from *first on (the merged contigs a merge round appended)
};

__device__ __forceinline__ uint32_t pick_span(const bool* any, const uint32_t* mn, const uint32_t* mx, uint32_t fl, uint32_t a, uint32_t* orient) {
    // forward: left anchor at mn[0], right anchor at mx[1];  reverse: revcomp(right) at mn[3], revcomp(left) at mx[2] (the oriented
    // contig is the reverse complement).  An unclipped flank's reverse hit is hidden by its forward hit.
    const bool lf = any[0], rf = any[1], lr = any[2] && !((fl & 1) && any[0]), rr = any[3] && !((fl & 2) && any[1]);
    *orient = 0;
    if (lf && rf) return mx[1] >= mn[0] + a ? mx[1] - (mn[0] + a) + 1 : 0;
    if (lr && rr) { *orient = 1; return mx[2] >= mn[3] + a ? mx[2] - (mn[3] + a) + 1 : 0; }
    return 0;
}

__global__ __launch_bounds__(256) void pick_anchor_kernel(PickParams P) {
    const uint32_t n = *P.n_contigs < P.contig_cap ? *P.n_contigs : P.contig_cap;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t a = P.a_s, al = P.a_l, ext = P.anc_l ? al - a : 0;
    for (uint32_t ci = (P.first ? *P.first : 0u) + wave; ci < n; ci += n_waves) {
        const gf_contig c = P.contigs[ci];
        if (c.gap >= P.n_gaps || c.length < 2 * a) continue;
        const uint8_t* as = P.anc_s + (uint64_t)c.gap * ANCHOR_ROW;
        if (as[0] == 0 || as[ANCHOR_MAX] == 0) continue;         // (no short anchors: no long ones either)
        const uint8_t* alp = P.anc_l ? P.anc_l + (uint64_t)c.gap * ANCHOR_ROW : nullptr;
        const bool has_long = alp && alp[0] != 0 && alp[ANCHOR_MAX] != 0 && c.length >= 2 * al;
        const char* s = P.seq + c.seq_off;
        uint32_t head[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint8_t* pat = as + q * ANCHOR_MAX;
            head[q] = (uint32_t)pat[0] | ((uint32_t)pat[1] << 8) | ((uint32_t)pat[2] << 16) | ((uint32_t)pat[3] << 24);
        }
        // positions where each pattern occurs: min and max per pattern, short [0..3] and long [4..7]
        uint32_t mn[8], mx[8];
        bool any[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { mn[q] = EMPTY32; mx[q] = 0; any[q] = false; }
        const uint32_t last = c.length - a;
        for (uint32_t p = lane; p <= last; p += 64) {
            uint32_t w = 0;                                       // four bases at p (a >= 8: they exist)
#pragma unroll
            for (int b = 0; b < 4; ++b) w |= (uint32_t)(uint8_t)s[p + b] << (8 * b);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (w != head[q]) continue;
                const uint8_t* pat = as + q * ANCHOR_MAX;
                uint32_t i = 4;
                while (i < a && (uint8_t)s[p + i] == pat[i]) ++i;
                if (i != a) continue;
                any[q] = true;
                mn[q] = p < mn[q] ? p : mn[q];
                mx[q] = p > mx[q] ? p : mx[q];
                if (!has_long) continue;
                // the long pattern around this short hit: the short one is its END for the left anchor and for revcomp(right), its
                // START for the right anchor and for revcomp(left)
                const uint8_t* lp = alp + q * ANCHOR_MAX;
                const bool at_end = q == 0 || q == 3;
                if (at_end ? p < ext : p + al > c.length) continue;
                const uint32_t p0 = at_end ? p - ext : p;
                bool ok = true;
                for (uint32_t j = 0; j < ext && ok; ++j) {
                    const uint32_t o = at_end ? j : a + j;
                    ok = (uint8_t)s[p0 + o] == lp[o];
                }
                if (!ok) continue;
                any[4 + q] = true;
                mn[4 + q] = p0 < mn[4 + q] ? p0 : mn[4 + q];
                mx[4 + q] = p0 > mx[4 + q] ? p0 : mx[4 + q];
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {   // wave reductions
            for (int d = 32; d >= 1; d >>= 1) {
                const uint32_t m2 = __shfl_xor(mn[q], d), x2 = __shfl_xor(mx[q], d);
                mn[q] = m2 < mn[q] ? m2 : mn[q];
                mx[q] = x2 > mx[q] ? x2 : mx[q];
            }
            any[q] = __ballot(any[q]) != 0;
        }
        if (lane != 0) continue;
        uint32_t orient = 0, best = 0, alen = al;
        if (has_long) best = pick_span(any + 4, mn + 4, mx + 4, alp[4 * ANCHOR_MAX], al, &orient);
        if (!best) { best = pick_span(any, mn, mx, as[4 * ANCHOR_MAX], a, &orient); alen = a; }
        if (!best) continue;
        if (best > 0xFFFFFFu) best = 0xFFFFFFu;   // the span field has 24 bits: a longer span saturates (it still outranks every shorter one), it never wraps
        const unsigned long long val = ((unsigned long long)alen << 56) | ((unsigned long long)best << 32) |
                                       ((unsigned long long)(0x7FFFFFFFu - ci) << 1) | orient;
        const unsigned long long old = atomicMax(P.gap_best + c.gap, val);
        if (old == 0) atomicAdd(P.n_closed, 1u);
    }
}

}  // namespace gf

using namespace gf;

extern "C" {

static int anchor_table(gf_ctx* ctx, int anchor_len, const uint8_t** out) {
    const size_t ng = ctx->gaps.size();
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
    *out = (const uint8_t*)tab.p;
    return GF_OK;
}

static int pick_anchored2(gf_ctx* ctx, const void* d_contigs, const void* d_n_contigs, size_t contig_cap, const void* d_seq,
                          int anchor_len, int anchor_len_short, const void* d_first, void* d_gap_best, void* d_n_closed) {
    if (!ctx || !d_contigs || !d_n_contigs || !d_seq || !d_gap_best || !d_n_closed || anchor_len < 8 || anchor_len > ANCHOR_MAX ||
        contig_cap > 0xFFFFFFFFull || (anchor_len_short && (anchor_len_short < 8 || anchor_len_short >= anchor_len)))
        return GF_E_INVAL;
    const size_t ng = ctx->gaps.size();
    if (ctx->flank_left.size() != ng || ctx->flank_right.size() != ng) return GF_E_STATE;
    if (!ng) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    PickParams P;
    int rc;
    P.anc_l = nullptr;
    if (anchor_len_short) {
        if ((rc = anchor_table(ctx, anchor_len, &P.anc_l))) return rc;
        if ((rc = anchor_table(ctx, anchor_len_short, &P.anc_s))) return rc;
    } else if ((rc = anchor_table(ctx, anchor_len, &P.anc_s))) return rc;
    P.a_s = (uint32_t)(anchor_len_short ? anchor_len_short : anchor_len);
    P.a_l = (uint32_t)anchor_len;
    P.contigs = (const gf_contig*)d_contigs;
    P.n_contigs = (const uint32_t*)d_n_contigs;
    P.contig_cap = (uint32_t)contig_cap;
    P.seq = (const char*)d_seq;
    P.n_gaps = (uint32_t)ng;
    P.gap_best = (unsigned long long*)d_gap_best;
    P.n_closed = (uint32_t*)d_n_closed;
    P.first = (const uint32_t*)d_first;
    LaunchTimer tm(ctx, GF_KERNEL_PICK);
    hipLaunchKernelGGL(pick_anchor_kernel, dim3(ctx->n_cu * 8), dim3(256), 0, ctx->stream, P);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_pick_anchored2_dev(gf_ctx* ctx, const void* d_contigs, const void* d_n_contigs, size_t contig_cap, const void* d_seq,
                          int anchor_len, int anchor_len_short, void* d_gap_best, void* d_n_closed) {
    return pick_anchored2(ctx, d_contigs, d_n_contigs, contig_cap, d_seq, anchor_len, anchor_len_short, nullptr, d_gap_best, d_n_closed);
}

int gf_pick_anchored2_from_dev(gf_ctx* ctx, const void* d_contigs, const void* d_n_contigs, size_t contig_cap, const void* d_seq,
                               int anchor_len, int anchor_len_short, const void* d_first, void* d_gap_best, void* d_n_closed) {
    if (!d_first) return GF_E_INVAL;
    return pick_anchored2(ctx, d_contigs, d_n_contigs, contig_cap, d_seq, anchor_len, anchor_len_short, d_first, d_gap_best, d_n_closed);
}

int gf_pick_anchored_dev(gf_ctx* ctx, const void* d_contigs, const void* d_n_contigs, size_t contig_cap, const void* d_seq,
                         int anchor_len, void* d_gap_best, void* d_n_closed) {
    return gf_pick_anchored2_dev(ctx, d_contigs, d_n_contigs, contig_cap, d_seq, anchor_len, 0, d_gap_best, d_n_closed);
}

}  // extern "C"
