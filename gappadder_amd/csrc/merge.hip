// merge.hip — first GPU piece of the contig dedup/merge step: the all-pairs k-mer prefilter of the reference's
// ContigsMerger.  CompactVer3 (ContigsCompactor.cpp:773-983) makes a node of every contig and of its reverse complement
// (:782-800), builds one QuickCheckerContigsMatch per node (the set of ALL 10-mers of the node, :2041-2056) and asks for every
// pair i <= j whether some 10-mer of the first or last 30 bases of node j is in node i's set (IsMatchFeasibleV2 :2020-2039,
// threadQuickCheck :1073-1098); only the surviving pairs get the O(n m) overlap DP (:1572-1976, not built).  The reference
// does this with std::map probes from a pthread pool; a per-gap contig set has tens to hundreds of contigs, and a run has one
// set per gap.
//
// Here: one workgroup per contig set.  The END k-mers of a chunk of nodes go into an LDS hash table keyed by the k-mer (entries
// (k-mer, j), 2 x (30 - k + 1) per node); then every position of every node i is probed against it — a hit on (k-mer, j) with
// j >= i sets bit (i, j) of the set's pair matrix; the matrix is compacted into (set, i, j) triples.  k-mers use KmerUtils'
// code (anything but C/G/T is A, KmerUtils.cpp:25-41; nothing is canonical); the reverse-complement node keeps non-ACGT symbols
// as they are (FastaSequence::RevsereComplement), i.e. as A.
#include <algorithm>
#include <cstring>

#include "gf_internal.hpp"

namespace gf {

constexpr uint32_t QC_SLOTS = 16384;            // 8-byte slots: 128 KiB of LDS
constexpr uint32_t QC_END = 30;                 // lenContigLen (ContigsCompactor.cpp:2024)
constexpr unsigned long long QC_EMPTY = ~0ull;

struct QcParams {
    const char* seq;
    const unsigned long long* contig_off;   // n_contigs + 1
    const unsigned long long* set_off;      // n_sets + 1 (contig indices)
    uint32_t n_sets;
    uint32_t k;
    uint32_t max_nodes;                     // nodes of the largest set (matrix slice = max_nodes^2 bits per workgroup)
    uint32_t* matrix;                       // [gridDim.x][words]
    uint32_t words;                         // words per slice
    gf_qcpair* out;
    uint32_t cap;
    uint32_t* n_out;
    uint32_t* next_set;
    uint32_t* error;                        // bit 0: a contig shorter than 30 bases (its set is skipped)
};

// base code at position p of node `node` (node = 2 * contig + strand) of a set whose first contig is c0
__device__ __forceinline__ uint32_t qc_code(const QcParams& P, unsigned long long c0, uint32_t node, uint32_t p, uint32_t len) {
    const unsigned long long o = P.contig_off[c0 + (node >> 1)];
    char ch = P.seq[o + ((node & 1) ? len - 1 - p : p)];
    if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
    const uint32_t f = ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 0u;
    if (!(node & 1)) return f;
    return (ch == 'A' || f) ? 3u - f : 0u;    // complement of A/C/G/T; any other symbol stays itself = A
}

__device__ __forceinline__ uint32_t qc_kmer(const QcParams& P, unsigned long long c0, uint32_t node, uint32_t p, uint32_t len) {
    uint32_t v = 0;
    for (uint32_t i = 0; i < P.k; ++i) v = (v << 2) | qc_code(P, c0, node, p + i, len);
    return v;
}

__device__ __forceinline__ uint32_t qc_slot(uint32_t kmer) { return ((kmer * 0x9E3779B1u) >> 7) & (QC_SLOTS - 1); }

__global__ __launch_bounds__(1024) void quick_check_kernel(QcParams P) {
    extern __shared__ unsigned long long qtab[];   // QC_SLOTS entries: k-mer << 32 | node j
    __shared__ uint32_t s_set, s_bad;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    uint32_t* mat = P.matrix + (size_t)blockIdx.x * P.words;
    const uint32_t per_node = 2 * (QC_END - P.k + 1);
    const uint32_t chunk = (QC_SLOTS / 2) / per_node;      // nodes whose end k-mers fill the table to at most one half
    for (;;) {
        __syncthreads();
        if (tid == 0) { s_set = atomicAdd(P.next_set, 1u); s_bad = 0; }
        __syncthreads();
        const uint32_t st = s_set;
        if (st >= P.n_sets) break;
        const unsigned long long c0 = P.set_off[st], c1 = P.set_off[st + 1];
        const uint32_t n = (uint32_t)(c1 - c0), M = 2 * n;
        if (n == 0) continue;
        if (M > P.max_nodes) { if (tid == 0) atomicOr(P.error, 2u); continue; }
        for (uint32_t c = tid; c < n; c += blockDim.x)
            if (P.contig_off[c0 + c + 1] - P.contig_off[c0 + c] < QC_END) s_bad = 1;
        const uint32_t mwords = (M * M + 31) / 32;
        for (uint32_t i = tid; i < mwords; i += blockDim.x) mat[i] = 0;
        __syncthreads();
        if (s_bad) { if (tid == 0) atomicOr(P.error, 1u); continue; }
        for (uint32_t j0 = 0; j0 < M; j0 += chunk) {
            const uint32_t j1 = j0 + chunk < M ? j0 + chunk : M;
            for (uint32_t i = tid; i < QC_SLOTS; i += blockDim.x) qtab[i] = QC_EMPTY;
            __syncthreads();
            // end k-mers of nodes j0 .. j1-1
            for (uint32_t w = tid; w < (j1 - j0) * per_node; w += blockDim.x) {
                const uint32_t j = j0 + w / per_node, e = w % per_node, half = per_node / 2;
                const uint32_t len = (uint32_t)(P.contig_off[c0 + (j >> 1) + 1] - P.contig_off[c0 + (j >> 1)]);
                const uint32_t p = e < half ? e : len - QC_END + (e - half);
                const uint32_t km = qc_kmer(P, c0, j, p, len);
                const unsigned long long mine = ((unsigned long long)km << 32) | j;
                uint32_t s = qc_slot(km);
                for (;;) {
                    unsigned long long v = qtab[s];
                    if (v == QC_EMPTY) {
                        v = atomicCAS(&qtab[s], QC_EMPTY, mine);
                        if (v == QC_EMPTY) break;
                    }
                    if (v == mine) break;       // the same k-mer twice in this node's ends
                    s = (s + 1) & (QC_SLOTS - 1);
                }
            }
            __syncthreads();
            // every position of every node i <= j1-1 asks for its k-mer; entries with j >= i make the pair feasible
            // work item = (node i, position): nodes are walked by the waves, positions by the lanes
            for (uint32_t i = tid >> 6; i < j1; i += blockDim.x >> 6) {
                const uint32_t len = (uint32_t)(P.contig_off[c0 + (i >> 1) + 1] - P.contig_off[c0 + (i >> 1)]);
                const uint32_t npos = len - P.k + 1;
                for (uint32_t p = lane; p < npos; p += 64) {
                    const uint32_t km = qc_kmer(P, c0, i, p, len);
                    uint32_t s = qc_slot(km);
                    for (;;) {
                        const unsigned long long v = qtab[s];
                        if (v == QC_EMPTY) break;
                        if ((uint32_t)(v >> 32) == km) {
                            const uint32_t j = (uint32_t)v;
                            if (j >= i) {
                                const uint32_t bit = i * M + j;
                                // (checked through L2, where the atomics land: this CU's L1 may still hold the slice of the previous set)
                                if (!((__hip_atomic_load(&mat[bit >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (bit & 31)) & 1u))
                                    atomicOr(&mat[bit >> 5], 1u << (bit & 31));
                            }
                        }
                        s = (s + 1) & (QC_SLOTS - 1);
                    }
                }
            }
            __syncthreads();
        }
        // the matrix slice went through L2 atomics and plain loads of this CU: make the loads below see the atomics
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        for (uint32_t w0 = 0; w0 < mwords; w0 += blockDim.x) {
            const uint32_t w = w0 + tid;
            uint32_t bits = w < mwords ? __hip_atomic_load(&mat[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            const uint32_t cnt = (uint32_t)__popc(bits);
            // wave-level reservation: prefix over the lanes' counts
            uint32_t pre = cnt;
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(pre, d);
                if ((int)lane >= d) pre += y;
            }
            const uint32_t total = __shfl(pre, 63);
            if (!total) continue;
            uint32_t base = 0;
            if (lane == 63) base = atomicAdd(P.n_out, total);
            base = __shfl(base, 63) + pre - cnt;
            while (bits) {
                const uint32_t b = (uint32_t)__ffs(bits) - 1;
                bits &= bits - 1;
                const uint32_t bit = w * 32 + b;
                if (base < P.cap) { gf_qcpair q; q.set = st; q.i = bit / M; q.j = bit % M; P.out[base] = q; }
                ++base;
            }
        }
    }
}

// ---- f-3, second stage: the overlap evaluation of a node pair (ContigsCompactor::Evaluate + IsScoreSignificant +
// ContigsCompactorAction::SetMergedStringConcat / IsContainment, ContigsCompactor-v0.2.0/ContigsMerger/ContigsCompactor.cpp:1572-1976,
// :108-159; semantics restated in oracle/gp_oracle.c::or_overlap_evaluate, pinned on the reference's own answers).  An overlap
// alignment (first row / column 0, match +1, mismatch, indel; predecessor order diagonal, up, left, each only on a strictly larger
// score); the end cell is the maximum over the last column / row shifted in by c = 0 .. max_clip in the reference's scan order.
// One workgroup per pair sweeps the anti-diagonals (three rolling diagonals of integer scores in LDS — the scores are integers
// whenever the indel score is, which the entry point requires); the reference's trace-back table is not needed: all its caller
// uses of the trace back is whether the path starts on row 0 or on column 0, two bits that travel with the scores; the end cell is
// kept per thread as (score, rank in the reference's scan order) and reduced at the end.
constexpr uint32_t OV_MAXLEN = 8190;
struct OvParams {
    const char* seq;
    const unsigned long long* contig_off;
    const unsigned long long* set_off;
    const gf_qcpair* pairs;
    uint32_t n_pairs;
    gf_ovl_params pr;
    gf_ovl_result* out;
    uint32_t* next;
};

__global__ __launch_bounds__(256) void overlap_eval_kernel(OvParams P) {
    extern __shared__ uint32_t sm[];   // [3 x (OV_MAXLEN + 2) scores][3 x (OV_MAXLEN + 2) start flags (bytes)][node 1][node 2]
    __shared__ uint32_t s_pair;
    __shared__ long long s_best_sc[256];
    __shared__ unsigned long long s_best_rk[256];
    constexpr uint32_t ROW = OV_MAXLEN + 2;
    int32_t* sc = reinterpret_cast<int32_t*>(sm);
    uint8_t* fl = reinterpret_cast<uint8_t*>(sm + 3 * ROW);
    char* a = reinterpret_cast<char*>(fl + 3 * ROW);
    char* b = a + ROW;
    const uint32_t tid = threadIdx.x;
    const int mis = (int)P.pr.mismatch, ind = (int)P.pr.indel, clip = (int)P.pr.max_clip;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_pair = atomicAdd(P.next, 1u);
        __syncthreads();
        const uint32_t pi = s_pair;
        if (pi >= P.n_pairs) break;
        const gf_qcpair q = P.pairs[pi];
        const unsigned long long c0 = P.set_off[q.set];
        const unsigned long long o1 = P.contig_off[c0 + q.i / 2], e1 = P.contig_off[c0 + q.i / 2 + 1];
        const unsigned long long o2 = P.contig_off[c0 + q.j / 2], e2 = P.contig_off[c0 + q.j / 2 + 1];
        const int n1 = (int)(e1 - o1), n2 = (int)(e2 - o2);
        gf_ovl_result r;
        memset(&r, 0, sizeof r);
        if (n1 > (int)OV_MAXLEN || n2 > (int)OV_MAXLEN || n1 < 1 || n2 < 1) {
            if (tid == 0) { r.res = -1; P.out[pi] = r; }
            continue;
        }
        auto load_node = [&](char* dst, unsigned long long o, int n, bool rc) {
            for (int i = (int)tid; i < n; i += 256) {
                char ch = P.seq[o + (rc ? n - 1 - i : i)];
                if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
                if (rc) ch = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : ch;
                dst[i] = ch;
            }
        };
        load_node(a, o1, n1, q.i & 1);
        load_node(b, o2, n2, q.j & 1);
        long long best_sc = -1000000000ll;
        unsigned long long best_rk = ~0ull;   // c << 40 | (row candidate) << 39 | index << 2 | start flags
        __syncthreads();
        for (int d = 0; d <= n1 + n2; ++d) {
            int32_t* cur = sc + (d % 3) * ROW;
            const int32_t* p1 = sc + ((d + 2) % 3) * ROW;   // diagonal d - 1
            const int32_t* p2 = sc + ((d + 1) % 3) * ROW;   // diagonal d - 2
            uint8_t* fcur = fl + (d % 3) * ROW;
            const uint8_t* f1 = fl + ((d + 2) % 3) * ROW;
            const uint8_t* f2 = fl + ((d + 1) % 3) * ROW;
            const int ilo = d > n2 ? d - n2 : 0, ihi = d < n1 ? d : n1;
            for (int i = ilo + (int)tid; i <= ihi; i += 256) {
                const int j = d - i;
                int s;
                uint32_t f;
                if (i == 0) { s = 0; f = 1u | (j == 0 ? 2u : 0u); }
                else if (j == 0) { s = 0; f = 2u; }
                else {
                    s = p2[i - 1] + (a[i - 1] == b[j - 1] ? 1 : mis);
                    f = f2[i - 1];
                    const int up = p1[i - 1] + ind, lf = p1[i] + ind;
                    if (s < up) { s = up; f = f1[i - 1]; }
                    if (s < lf) { s = lf; f = f1[i]; }
                }
                cur[i] = s;
                fcur[i] = (uint8_t)f;
                // end-cell candidates: column n2 - c (scanned over i) before row n1 - c (scanned over j), c ascending
                unsigned long long rk = ~0ull;
                if (n2 - j <= clip) rk = ((unsigned long long)(n2 - j) << 40) | ((unsigned long long)i << 2);
                if (n1 - i <= clip) {
                    const unsigned long long rr = ((unsigned long long)(n1 - i) << 40) | (1ull << 39) | ((unsigned long long)j << 2);
                    if (rr < rk) rk = rr;
                }
                if (rk != ~0ull && (s > best_sc || (s == best_sc && rk < (best_rk & ~3ull)))) { best_sc = s; best_rk = rk | f; }
            }
            __syncthreads();
        }
        s_best_sc[tid] = best_sc;
        s_best_rk[tid] = best_rk;
        __syncthreads();
        if (tid == 0) {
            for (uint32_t t = 1; t < 256; ++t)
                if (s_best_sc[t] > best_sc || (s_best_sc[t] == best_sc && (s_best_rk[t] & ~3ull) < (best_rk & ~3ull))) { best_sc = s_best_sc[t]; best_rk = s_best_rk[t]; }
            const int nclip = (int)(best_rk >> 40);
            const bool rowc = (best_rk >> 39) & 1u;
            const int idx = (int)((best_rk >> 2) & 0x1FFFFFFFFull), fend = (int)(best_rk & 3u);
            const int row_end = rowc ? n1 - nclip : idx, col_end = rowc ? idx : n2 - nclip;
            const int score = (int)best_sc;
            int ov0 = n1 < n2 ? n1 : n2, ov1 = ov0, ov2 = ov0;
            if (row_end + nclip == n1) ov1 = col_end;
            if (col_end + nclip == n2) ov2 = row_end;
            int ov = ov1 < ov2 ? ov1 : ov2;
            if (ov0 < ov) ov = ov0;
            int res = 2;
            if (ov < n1 * P.pr.frac_min_overlap && ov < n2 * P.pr.frac_min_overlap) res = 0;
            else if (row_end + nclip == n1 && col_end + 5 - 1 >= n2) res = 0;
            else if (col_end + nclip == n2 && row_end + 5 - 1 >= n1) res = 0;
            else if (score < ov * (1 - P.pr.frac_loss)) res = 0;
            else if (ov < P.pr.min_overlap_scaffold) res = 0;
            else if (ov < P.pr.min_overlap) res = 1;
            if (P.pr.relax != 0.0) res = 2;   // Evaluate's fRelax mode (ContigsCompactor.cpp:1712-1725): no significance test
            r.res = res; r.row_end = row_end; r.col_end = col_end; r.nclip = nclip; r.score = score;
            if (res) {
                const int contained = (row_end + nclip == n1 && (fend & 1)) || (col_end + nclip == n2 && (fend & 2));
                int merged;
                if (contained && row_end + nclip == n1 && n1 < n2) merged = n2;
                else if (contained && col_end + nclip == n2 && n2 < n1) merged = n1;
                else if (row_end + nclip == n1) merged = (n1 - nclip) + (n2 - col_end);
                else merged = (n2 - nclip) + (n1 - row_end);
                r.contained = contained;
                r.merged_len = merged;
                r.overlap = n1 + n2 - nclip - merged;
                r.containment = contained && ((row_end + nclip == n1 && n1 < col_end) || (col_end + nclip == n2 && n2 < row_end));
                r.first_goes_first = (row_end + nclip) == n1;
            }
            P.out[pi] = r;
        }
    }
}

}  // namespace gf

using namespace gf;

extern "C" {

int gf_quick_check_dev(gf_ctx* ctx, const void* d_seq, const void* d_contig_off, const void* d_set_off, size_t n_sets, size_t max_set_contigs,
                       int k, void* d_out, size_t cap, void* d_n_out) {
    if (!ctx || !d_contig_off || !d_set_off || !d_n_out || (cap && !d_out) || k < 4 || k > 16 || n_sets >= 0xFFFFFFFFull || cap > 0xFFFFFFFFull ||
        max_set_contigs > 16384)
        return GF_E_INVAL;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->counters, GF_COUNTER_BYTES))) return rc;
    uint32_t* d_next = (uint32_t*)ctx->counters.p + 9;
    uint32_t* d_err = (uint32_t*)d_n_out + 1;   // the caller's second word: skipped sets (bit 0: a contig shorter than 30 bases, bit 1: more contigs than max_set_contigs)
    zero_regions(ctx, ZeroList{{(uint32_t*)d_n_out, d_next, nullptr, nullptr}, {2, 1, 0, 0}});
    if (!n_sets) return GF_OK;
    QcParams P;
    P.seq = (const char*)d_seq;
    P.contig_off = (const unsigned long long*)d_contig_off;
    P.set_off = (const unsigned long long*)d_set_off;
    P.n_sets = (uint32_t)n_sets;
    P.k = (uint32_t)k;
    P.max_nodes = (uint32_t)(2 * std::max<size_t>(1, max_set_contigs));
    P.words = (uint32_t)(((uint64_t)P.max_nodes * P.max_nodes + 31) / 32 + 1);
    const unsigned grid = (unsigned)std::min<size_t>(n_sets, (size_t)ctx->n_cu);
    if ((rc = ensure(ctx, ctx->xchg_ws2, (size_t)grid * P.words * 4 + 256))) return rc;
    P.matrix = (uint32_t*)ctx->xchg_ws2.p;
    P.out = (gf_qcpair*)d_out;
    P.cap = (uint32_t)cap;
    P.n_out = (uint32_t*)d_n_out;
    P.next_set = d_next;
    P.error = d_err;
    {
        LaunchTimer tm(ctx, GF_KERNEL_MERGE);
        hipLaunchKernelGGL(quick_check_kernel, dim3(grid), dim3(1024), (size_t)QC_SLOTS * 8, ctx->stream, P);
    }
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_quick_check(gf_ctx* ctx, const char* seq, const uint64_t* contig_off, const uint64_t* set_off, size_t n_sets, int k, gf_qcpair* out,
                   size_t cap, size_t* n_out) {
    if (!ctx || !n_out || (n_sets && (!contig_off || !set_off)) || (cap && !out)) return GF_E_INVAL;
    *n_out = 0;
    if (!n_sets) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n_contigs = (size_t)set_off[n_sets];
    size_t max_set = 0;
    for (size_t s = 0; s < n_sets; ++s) {
        if (set_off[s] > set_off[s + 1]) return GF_E_INVAL;
        max_set = std::max<size_t>(max_set, (size_t)(set_off[s + 1] - set_off[s]));
    }
    const size_t n_bytes = (size_t)contig_off[n_contigs];
    for (size_t c = 0; c < n_contigs; ++c)
        if (contig_off[c] > contig_off[c + 1] || contig_off[c + 1] - contig_off[c] < QC_END) {
            ctx->last_error = "gf_quick_check: contig " + std::to_string(c) + " has fewer than 30 bases";
            return GF_E_INVAL;
        }
    int rc;
    const size_t b_seq = (n_bytes + 63) & ~(size_t)63, b_co = ((n_contigs + 1) * 8 + 63) & ~(size_t)63, b_so = ((n_sets + 1) * 8 + 63) & ~(size_t)63;
    if ((rc = ensure(ctx, ctx->stage_in, b_seq + b_co + b_so + 64))) return rc;
    if ((rc = ensure(ctx, ctx->stage_out, std::max<size_t>(cap, 1) * sizeof(gf_qcpair) + 64))) return rc;
    uint8_t* d_seq = (uint8_t*)ctx->stage_in.p;
    uint8_t* d_co = d_seq + b_seq;
    uint8_t* d_so = d_co + b_co;
    uint8_t* d_cnt = d_so + b_so;
    if (n_bytes) GF_HIP(ctx, hipMemcpyAsync(d_seq, seq, n_bytes, hipMemcpyHostToDevice, ctx->stream));
    GF_HIP(ctx, hipMemcpyAsync(d_co, contig_off, (n_contigs + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    GF_HIP(ctx, hipMemcpyAsync(d_so, set_off, (n_sets + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = gf_quick_check_dev(ctx, d_seq, d_co, d_so, n_sets, max_set, k, ctx->stage_out.p, cap, d_cnt))) return rc;
    uint32_t nn[2] = {0, 0};
    GF_HIP(ctx, hipMemcpyAsync(nn, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t n = nn[0];
    if (nn[1]) {   // (cannot happen after the checks above; a device-side caller reads the same word)
        ctx->last_error = "gf_quick_check: contig sets were skipped (flags " + std::to_string(nn[1]) + ")";
        return GF_E_INVAL;
    }
    *n_out = n;
    if (n > cap) return GF_E_NOSPACE;
    if (n) GF_HIP(ctx, hipMemcpy(out, ctx->stage_out.p, n * sizeof(gf_qcpair), hipMemcpyDeviceToHost));
    std::sort(out, out + n, [](const gf_qcpair& a, const gf_qcpair& b) {
        if (a.set != b.set) return a.set < b.set;
        if (a.i != b.i) return a.i < b.i;
        return a.j < b.j;
    });
    return GF_OK;
}

int gf_overlap_evaluate_dev(gf_ctx* ctx, const void* d_seq, const void* d_contig_off, const void* d_set_off, const void* d_pairs, size_t n_pairs,
                            const gf_ovl_params* params, void* d_out) {
    if (!ctx || !params || (n_pairs && (!d_seq || !d_contig_off || !d_set_off || !d_pairs || !d_out)) || n_pairs >= 0xFFFFFFFFull) return GF_E_INVAL;
    // integer scores: the reference assigns the mismatch score to an int (ContigsCompactor.cpp:1640); with an integral indel score
    // every cell is an integer and the kernel's int32 sweep is the reference's double table exactly
    if (params->indel != (double)(int)params->indel || params->max_clip < 0 || params->max_clip > 1e6) {
        ctx->last_error = "gf_overlap_evaluate: the indel score must be integral";
        return GF_E_UNSUPPORTED;
    }
    GF_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->counters, GF_COUNTER_BYTES))) return rc;
    uint32_t* d_next = (uint32_t*)ctx->counters.p + 11;
    zero_regions(ctx, ZeroList{{d_next, nullptr, nullptr, nullptr}, {1, 0, 0, 0}});
    if (!n_pairs) return GF_OK;
    OvParams P;
    P.seq = (const char*)d_seq;
    P.contig_off = (const unsigned long long*)d_contig_off;
    P.set_off = (const unsigned long long*)d_set_off;
    P.pairs = (const gf_qcpair*)d_pairs;
    P.n_pairs = (uint32_t)n_pairs;
    P.pr = *params;
    P.out = (gf_ovl_result*)d_out;
    P.next = d_next;
    const size_t lds = (size_t)3 * (OV_MAXLEN + 2) * 4 + (size_t)3 * (OV_MAXLEN + 2) + (size_t)2 * (OV_MAXLEN + 2) + 16;
    {
        LaunchTimer tm(ctx, GF_KERNEL_MERGE);
        hipLaunchKernelGGL(overlap_eval_kernel, dim3((unsigned)std::min<size_t>(n_pairs, (size_t)ctx->n_cu)), dim3(256), lds, ctx->stream, P);
    }
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_overlap_evaluate(gf_ctx* ctx, const char* seq, const uint64_t* contig_off, const uint64_t* set_off, size_t n_sets, const gf_qcpair* pairs,
                        size_t n_pairs, const gf_ovl_params* params, gf_ovl_result* out) {
    if (!ctx || !params || (n_pairs && (!seq || !contig_off || !set_off || !pairs || !out || !n_sets))) return GF_E_INVAL;
    if (!n_pairs) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n_contigs = (size_t)set_off[n_sets];
    const size_t n_bytes = (size_t)contig_off[n_contigs];
    for (size_t p = 0; p < n_pairs; ++p) {
        if (pairs[p].set >= n_sets) return GF_E_INVAL;
        const size_t nc = (size_t)(set_off[pairs[p].set + 1] - set_off[pairs[p].set]);
        if (pairs[p].i >= 2 * nc || pairs[p].j >= 2 * nc) return GF_E_INVAL;
        for (uint32_t nd : {pairs[p].i, pairs[p].j}) {
            const size_t c = (size_t)set_off[pairs[p].set] + nd / 2;
            const uint64_t len = contig_off[c + 1] - contig_off[c];
            if (contig_off[c] > contig_off[c + 1] || len < 1 || len > OV_MAXLEN) {
                ctx->last_error = "gf_overlap_evaluate: contig " + std::to_string(c) + " is empty or longer than " + std::to_string(OV_MAXLEN) + " bases";
                return GF_E_INVAL;
            }
        }
    }
    int rc;
    const size_t b_seq = (n_bytes + 63) & ~(size_t)63, b_co = ((n_contigs + 1) * 8 + 63) & ~(size_t)63, b_so = ((n_sets + 1) * 8 + 63) & ~(size_t)63,
                 b_pr = (n_pairs * sizeof(gf_qcpair) + 63) & ~(size_t)63;
    if ((rc = ensure(ctx, ctx->stage_in, b_seq + b_co + b_so + b_pr + 64))) return rc;
    if ((rc = ensure(ctx, ctx->stage_out, n_pairs * sizeof(gf_ovl_result) + 64))) return rc;
    uint8_t* d_seq = (uint8_t*)ctx->stage_in.p;
    uint8_t* d_co = d_seq + b_seq;
    uint8_t* d_so = d_co + b_co;
    uint8_t* d_pr = d_so + b_so;
    GF_HIP(ctx, hipMemcpyAsync(d_seq, seq, n_bytes, hipMemcpyHostToDevice, ctx->stream));
    GF_HIP(ctx, hipMemcpyAsync(d_co, contig_off, (n_contigs + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    GF_HIP(ctx, hipMemcpyAsync(d_so, set_off, (n_sets + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    GF_HIP(ctx, hipMemcpyAsync(d_pr, pairs, n_pairs * sizeof(gf_qcpair), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = gf_overlap_evaluate_dev(ctx, d_seq, d_co, d_so, d_pr, n_pairs, params, ctx->stage_out.p))) return rc;
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GF_HIP(ctx, hipMemcpy(out, ctx->stage_out.p, n_pairs * sizeof(gf_ovl_result), hipMemcpyDeviceToHost));
    return GF_OK;
}

}  // extern "C"
