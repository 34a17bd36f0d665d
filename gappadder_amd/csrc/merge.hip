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
    const uint32_t* d_n_sets;               // or null: the number of sets is this device word (n_sets is then an upper bound)
    uint32_t* set_range;                    // or null: [2 * set] = first triple of the set, [2 * set + 1] = its triples (a set's triples are contiguous)
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
    __shared__ uint32_t s_scan[20], s_base;
    const uint32_t n_sets = P.d_n_sets ? (*P.d_n_sets < P.n_sets ? *P.d_n_sets : P.n_sets) : P.n_sets;
    for (;;) {
        __syncthreads();
        if (tid == 0) { s_set = atomicAdd(P.next_set, 1u); s_bad = 0; }
        __syncthreads();
        const uint32_t st = s_set;
        if (st >= n_sets) break;
        if (P.set_range && tid < 2) P.set_range[2 * st + tid] = 0;
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
        // the set's triples leave as ONE contiguous range (one reservation per set): a caller that works per set (the merge round of
        // the step) finds them at set_range; order inside the range = (i, j) ascending
        uint32_t my = 0;
        for (uint32_t w = tid; w < mwords; w += blockDim.x) my += (uint32_t)__popc(__hip_atomic_load(&mat[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        for (int d = 32; d >= 1; d >>= 1) my += __shfl_xor(my, d);
        if (lane == 0) s_scan[tid >> 6] = my;
        __syncthreads();
        if (tid == 0) {
            uint32_t tot = 0;
            for (uint32_t q = 0; q < (blockDim.x >> 6); ++q) tot += s_scan[q];
            s_base = tot ? atomicAdd(P.n_out, tot) : 0u;
            if (P.set_range) { P.set_range[2 * st] = s_base; P.set_range[2 * st + 1] = tot; }
        }
        __syncthreads();
        uint32_t run = s_base;
        for (uint32_t w0 = 0; w0 < mwords; w0 += blockDim.x) {
            const uint32_t w = w0 + tid;
            uint32_t bits = w < mwords ? __hip_atomic_load(&mat[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            const uint32_t cnt = (uint32_t)__popc(bits);
            uint32_t pre = cnt;
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(pre, d);
                if ((int)lane >= d) pre += y;
            }
            __syncthreads();
            if (lane == 63) s_scan[tid >> 6] = pre;
            __syncthreads();
            uint32_t before = 0, chunk_total = 0;
            for (uint32_t q = 0; q < (blockDim.x >> 6); ++q) { const uint32_t t = s_scan[q]; if (q < (tid >> 6)) before += t; chunk_total += t; }
            uint32_t base = run + before + pre - cnt;
            while (bits) {
                const uint32_t b = (uint32_t)__ffs(bits) - 1;
                bits &= bits - 1;
                const uint32_t bit = w * 32 + b;
                if (base < P.cap) { gf_qcpair q; q.set = st; q.i = bit / M; q.j = bit % M; P.out[base] = q; }
                ++base;
            }
            run += chunk_total;
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
    const uint32_t* d_n_pairs;   // or null: the number of pairs is this device word (n_pairs is then an upper bound = the capacity of `pairs`)
};

// LDS of one evaluation: three rolling diagonals of cells, the two nodes.  A cell is ONE word, score << 2 | start flags (bit 0: the path
// starts on row 0, bit 1: on column 0): three 4-byte reads + two base reads and one write per cell (score and flags in separate arrays were
// eight LDS reads and two writes per cell, and the sweep is bound by them: DESIGN.md §4)
constexpr uint32_t OV_ROW = OV_MAXLEN + 2;
constexpr size_t OV_LDS_BYTES = (size_t)3 * OV_ROW * 4 + (size_t)2 * OV_ROW + 16;
struct OvLds {
    int32_t* sc;
    char* a;
    char* b;
    __device__ explicit OvLds(uint32_t* sm) {
        sc = reinterpret_cast<int32_t*>(sm);
        a = reinterpret_cast<char*>(sm + 3 * OV_ROW);
        b = a + OV_ROW;
    }
};
// node n1 (S.a) against node n2 (S.b), both staged in LDS; all NT threads of the workgroup; the result is in *res (shared memory) after
// the call.  NT = 1 024: a diagonal of a 2-kb pair is 2-3 cells per thread and sixteen waves hide each other's LDS latency — the sweep
// is a chain of ~(n1 + n2) barriers, 19.3 ms per 6 600 pairs with 256 threads (C5's merge round), measured again in DESIGN.md
constexpr int OV_NT = 1024;
template <int NT>
__device__ void ov_evaluate(const OvLds& S, int n1, int n2, const gf_ovl_params& pr, long long* s_best_sc, unsigned long long* s_best_rk,
                            gf_ovl_result* res) {
    constexpr uint32_t ROW = OV_ROW;
    int32_t* sc = S.sc;
    const char* a = S.a;
    const char* b = S.b;
    const uint32_t tid = threadIdx.x;
    const int mis4 = 4 * (int)pr.mismatch, ind4 = 4 * (int)pr.indel, clip = (int)pr.max_clip;
    long long best_sc = -1000000000ll;
    unsigned long long best_rk = ~0ull;   // c << 40 | (row candidate) << 39 | index << 2 | start flags
    __syncthreads();
    for (int d = 0; d <= n1 + n2; ++d) {
        int32_t* cur = sc + (d % 3) * ROW;
        const int32_t* p1 = sc + ((d + 2) % 3) * ROW;   // diagonal d - 1
        const int32_t* p2 = sc + ((d + 1) % 3) * ROW;   // diagonal d - 2
        const int ilo = d > n2 ? d - n2 : 0, ihi = d < n1 ? d : n1;
        for (int i = ilo + (int)tid; i <= ihi; i += NT) {
            const int j = d - i;
            int v;                                      // score << 2 | flags
            if (i == 0) v = 1 | (j == 0 ? 2 : 0);
            else if (j == 0) v = 2;
            else {
                // predecessor order diagonal, up, left, each only on a strictly larger SCORE: x.score < y.score <=> (x | 3) < (y & ~3)
                v = p2[i - 1] + (a[i - 1] == b[j - 1] ? 4 : mis4);
                const int up = p1[i - 1] + ind4, lf = p1[i] + ind4;
                if ((v | 3) < (up & ~3)) v = up;
                if ((v | 3) < (lf & ~3)) v = lf;
            }
            cur[i] = v;
            // end-cell candidates: column n2 - c (scanned over i) before row n1 - c (scanned over j), c ascending
            if (n2 - j <= clip || n1 - i <= clip) {
                const int s = v >> 2;
                unsigned long long rk = ~0ull;
                if (n2 - j <= clip) rk = ((unsigned long long)(n2 - j) << 40) | ((unsigned long long)i << 2);
                if (n1 - i <= clip) {
                    const unsigned long long rr = ((unsigned long long)(n1 - i) << 40) | (1ull << 39) | ((unsigned long long)j << 2);
                    if (rr < rk) rk = rr;
                }
                if (s > best_sc || (s == best_sc && rk < (best_rk & ~3ull))) { best_sc = s; best_rk = rk | (unsigned)(v & 3); }
            }
        }
        __syncthreads();
    }
    // the best end cell: (score descending, rank ascending) — first inside every wave, then over the waves
    for (int d = 32; d >= 1; d >>= 1) {
        const long long osc = __shfl_xor(best_sc, d);
        const unsigned long long ork = __shfl_xor(best_rk, d);
        if (osc > best_sc || (osc == best_sc && (ork & ~3ull) < (best_rk & ~3ull))) { best_sc = osc; best_rk = ork; }
    }
    if ((tid & 63) == 0) { s_best_sc[tid >> 6] = best_sc; s_best_rk[tid >> 6] = best_rk; }
    __syncthreads();
    if (tid == 0) {
        gf_ovl_result r;
        memset(&r, 0, sizeof r);
        for (uint32_t t = 1; t < NT / 64; ++t)
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
        int rs = 2;
        if (ov < n1 * pr.frac_min_overlap && ov < n2 * pr.frac_min_overlap) rs = 0;
        else if (row_end + nclip == n1 && col_end + 5 - 1 >= n2) rs = 0;
        else if (col_end + nclip == n2 && row_end + 5 - 1 >= n1) rs = 0;
        else if (score < ov * (1 - pr.frac_loss)) rs = 0;
        else if (ov < pr.min_overlap_scaffold) rs = 0;
        else if (ov < pr.min_overlap) rs = 1;
        if (pr.relax != 0.0) rs = 2;   // Evaluate's fRelax mode (ContigsCompactor.cpp:1712-1725): no significance test
        r.res = rs; r.row_end = row_end; r.col_end = col_end; r.nclip = nclip; r.score = score;
        if (rs) {
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
        *res = r;
    }
    __syncthreads();
}
// stages `n` bases from global memory into LDS: upper-cased, reverse-complemented when rc (other symbols stay themselves)
__device__ __forceinline__ void ov_load_node(char* dst, const char* src, int n, bool rc) {
    for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) {
        char ch = src[rc ? n - 1 - i : i];
        if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
        if (rc) ch = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : ch;
        dst[i] = ch;
    }
}

__global__ __launch_bounds__(OV_NT) void overlap_eval_kernel(OvParams P) {
    extern __shared__ uint32_t sm[];   // [3 x (OV_MAXLEN + 2) cells: score << 2 | start flags][node 1][node 2]
    __shared__ uint32_t s_pair;
    __shared__ long long s_best_sc[OV_NT / 64];
    __shared__ unsigned long long s_best_rk[OV_NT / 64];
    __shared__ gf_ovl_result s_res;
    const OvLds S(sm);
    const uint32_t tid = threadIdx.x;
    const uint32_t n_pairs = P.d_n_pairs ? (*P.d_n_pairs < P.n_pairs ? *P.d_n_pairs : P.n_pairs) : P.n_pairs;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_pair = atomicAdd(P.next, 1u);
        __syncthreads();
        const uint32_t pi = s_pair;
        if (pi >= n_pairs) break;
        const gf_qcpair q = P.pairs[pi];
        const unsigned long long c0 = P.set_off[q.set];
        const unsigned long long o1 = P.contig_off[c0 + q.i / 2], e1 = P.contig_off[c0 + q.i / 2 + 1];
        const unsigned long long o2 = P.contig_off[c0 + q.j / 2], e2 = P.contig_off[c0 + q.j / 2 + 1];
        const int n1 = (int)(e1 - o1), n2 = (int)(e2 - o2);
        if (n1 > (int)OV_MAXLEN || n2 > (int)OV_MAXLEN || n1 < 1 || n2 < 1) {
            if (tid == 0) { gf_ovl_result r; memset(&r, 0, sizeof r); r.res = -1; P.out[pi] = r; }
            continue;
        }
        ov_load_node(S.a, P.seq + o1, n1, q.i & 1);
        ov_load_node(S.b, P.seq + o2, n2, q.j & 1);
        ov_evaluate<OV_NT>(S, n1, n2, P.pr, s_best_sc, s_best_rk, &s_res);
        if (tid == 0) P.out[pi] = s_res;
    }
}

// ---- the contig-merge ROUND of the step, on the device (assemble_gaps.py:301-306, 335-339: run_contigs_merge before the pick; here for the
// gaps the first pick left open, whose contig sets alone can gain from merging).  What gappadder_amd/MergeContigs.py::merge_sets and
// Pipeline.merge_open_gaps do on the host between batched GPU calls runs as a chain of launches without a host synchronisation:
//   mg_count / mg_scan_gaps / mg_fill     the open gaps' contig lists (ascending contig index)
//   mg_dedup                              exact-containment dedup (MergeContigs.drop_contained) per gap, one workgroup per gap
//   mg_scan_sets / mg_copy                the sets that take part (2 .. max_set contigs left) as the node arrays of the two kernels above
//   quick_check_kernel, overlap_eval_kernel   the merger's prefilter and overlap evaluation = the EDGES of its graph (device counts)
//   mg_paths                              per set: adjacency in the reference's edge order, strongly connected components (Tarjan from
//                                         node 0 up, result reversed), roots and ends, shortest-path DP per root, twin removal
//                                         (GraphUtils.cpp:625-859, 1028-1178, 1258-1344, 1422-1454 as restated in MergeContigs.find_paths)
//   mg_scan_jobs / mg_strings             FormMergedSeqFromPath per path: the running string against the next node in Evaluate's relaxed
//                                         mode (ov_evaluate above), merged strings appended to the contig list as records with k = kv = 0
// Everything order-dependent follows the host twin exactly (tests/test_gpu_merge.py compares the two on the reference's own KAT sets and
// on random sets); capacities that overflow set bits of stats[MG_ERR].
constexpr uint32_t MG_MAX_IN = 1024;      // contigs of an open gap that the dedup takes (more: the gap is left alone, counted in stats[MG_SKIPPED])
constexpr uint32_t MG_MAX_NODES = 256;    // nodes of a set (2 x max_set)
constexpr uint32_t MG_MAX_EDGES = 4096;   // edges of one set's graph
constexpr uint32_t MG_MAX_PATHS = 2048;   // paths of one set before the twin removal
constexpr uint32_t MG_PATH_BYTES = 1u << 17;   // their nodes (bytes) per workgroup
constexpr uint32_t MG_MIN_NODE = 30, MG_MAX_NODE = OV_MAXLEN;
constexpr uint32_t MG_PER_ROOT = 21;      // MAX_CONTIG_IN_PATH_COUNT + 1 (ContigsCompactor.cpp:34; MergeContigs.find_paths)
enum { MG_N_PRE = 0, MG_N_SETS = 1, MG_SKIPPED = 2, MG_N_PAIRS = 3, MG_QC_FLAGS = 4, MG_N_JOBS = 5, MG_ERR = 6, MG_N0 = 7, MG_N_EDGES = 8,
       MG_SETS_WITH_JOBS = 9, MG_Q_JOBS = 10, MG_Q_SETS = 11, MG_JOB_NODES = 12, MG_Q_DEDUP = 13, MG_Q_COPY = 14, MG_N_NODES = 15, MG_SKIPPED_GRAPH = 16,
       MG_WORDS = 32 };
// error bits: capacities of this call (the caller sizes them: raise).  A set whose GRAPH outgrows the round's own limits — more than
// MG_MAX_EDGES edges, MG_MAX_PATHS paths or MG_PATH_BYTES path nodes (the contig graph of a repeat-bearing gap has thousands of paths) —
// is left alone and counted in stats[MG_SKIPPED_GRAPH], like the sets of more than max_set contigs in stats[MG_SKIPPED]
constexpr uint32_t MG_E_SEQ = 1, MG_E_PAIRS = 2, MG_E_CONTIGS = 32, MG_E_OUTSEQ = 64;

struct MgJob { uint32_t set, off, len; };   // path = job_nodes[off .. off + len)

struct MgParams {
    gf_contig* contigs;
    uint32_t* n_contigs;
    uint32_t contig_cap;
    char* seq;
    unsigned long long* seq_len;
    unsigned long long seq_cap;
    const unsigned long long* gap_best;
    uint32_t n_gaps, max_set;
    uint32_t* stats;
    // workspace
    uint32_t* cnt;          // [n_gaps] contigs of an open gap, later the fill cursor
    uint32_t* pre_of_gap;   // [n_gaps]
    uint32_t* pre_gap;      // [n_gaps]
    uint32_t* pre_off;      // [n_gaps + 1]
    uint32_t* ids;          // [contig_cap]
    uint32_t* kept_n;       // [n_gaps] contigs left by the dedup
    uint32_t* node_n;       // [n_gaps] ... of node length
    unsigned long long* node_bytes;   // [n_gaps]
    uint32_t* set_pre;      // [n_gaps]
    unsigned long long* set_base;     // [n_gaps] first byte of the set in mseq
    unsigned long long* contig_off;   // [node_cap + 1]
    unsigned long long* set_off;      // [n_gaps + 1]
    uint32_t node_cap;
    char* mseq;
    unsigned long long mseq_cap;
    // graph + paths
    const gf_qcpair* pairs;
    const gf_ovl_result* res;
    uint32_t pair_cap;
    const uint32_t* set_range;        // [2 * set]
    uint8_t* path_ws;                 // per workgroup: MG_PATH_BYTES of path nodes
    int32_t* dp_dist;                 // per workgroup: [MG_MAX_NODES roots][MG_MAX_NODES]
    uint8_t* dp_pred;                 // ... pred, and 1 byte of flags (bit 0 reached, bit 1 ends with its node twice)
    uint8_t* dp_flag;
    MgJob* jobs;
    uint32_t job_cap;
    uint8_t* job_nodes;
    uint32_t job_node_cap;
    uint32_t* set_jobs;               // [2 * set]: first job, jobs
    uint32_t* set_rec;                // [set]: contig record of the set's first job
    char* cur_ws;                     // per workgroup: 2 x 16 384 bytes (the running string and its successor)
    gf_ovl_params pr;
    // order of a gap's contigs = the order of its contigs.fa (assemble_gaps.py:124-135): the (k, kv) pairs in list order, inside a pair by
    // (length descending, sequence) — n_k > 0; n_k == 0: record order (contig index)
    uint32_t n_k;
    uint16_t k_list[16], kv_list[16];
};

__device__ __forceinline__ uint32_t mg_block_scan_excl(uint32_t v, uint32_t* s_w, uint32_t* total) {   // blockDim.x a multiple of 64, <= 1024
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

__global__ __launch_bounds__(256) void mg_count_kernel(MgParams P) {
    const uint32_t n = *P.n_contigs < P.contig_cap ? *P.n_contigs : P.contig_cap;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
        const uint32_t g = P.contigs[c].gap;
        if (g < P.n_gaps && P.gap_best[g] == 0) atomicAdd(&P.cnt[g], 1u);
    }
}

__global__ __launch_bounds__(1024) void mg_scan_gaps_kernel(MgParams P) {
    __shared__ uint32_t s_w[20];
    uint32_t carry_set = 0, carry_off = 0, skipped = 0;
    for (uint32_t g0 = 0; g0 < P.n_gaps; g0 += blockDim.x) {
        const uint32_t g = g0 + threadIdx.x;
        const uint32_t c = g < P.n_gaps ? P.cnt[g] : 0u;
        const bool el = c >= 2 && c <= MG_MAX_IN;
        if (c > MG_MAX_IN) ++skipped;
        uint32_t t1, t2;
        const uint32_t e1 = mg_block_scan_excl(el ? 1u : 0u, s_w, &t1);
        const uint32_t e2 = mg_block_scan_excl(el ? c : 0u, s_w, &t2);
        if (g < P.n_gaps) {
            P.pre_of_gap[g] = el ? carry_set + e1 : EMPTY32;
            if (el) { P.pre_gap[carry_set + e1] = g; P.pre_off[carry_set + e1] = carry_off + e2; }
            P.cnt[g] = 0;     // (from here on the fill cursor)
        }
        carry_set += t1;
        carry_off += t2;
        __syncthreads();
    }
    if (skipped) atomicAdd(&P.stats[MG_SKIPPED], skipped);
    if (threadIdx.x == 0) {
        P.pre_off[carry_set] = carry_off;
        P.stats[MG_N_PRE] = carry_set;
        P.stats[MG_N0] = *P.n_contigs < P.contig_cap ? *P.n_contigs : P.contig_cap;
    }
}

__global__ __launch_bounds__(256) void mg_fill_kernel(MgParams P) {
    const uint32_t n = P.stats[MG_N0];
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
        const uint32_t g = P.contigs[c].gap;
        if (g >= P.n_gaps) continue;
        const uint32_t pre = P.pre_of_gap[g];
        if (pre == EMPTY32) continue;
        P.ids[P.pre_off[pre] + atomicAdd(&P.cnt[g], 1u)] = c;
    }
}

__device__ __forceinline__ char mg_comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c; }

// one workgroup per open gap: q is dropped iff it occurs, on either strand, inside a contig that precedes it in (length descending,
// index ascending) order — containment is transitive, so this is drop_contained's "inside a KEPT contig" without its serial order
__global__ __launch_bounds__(1024) void mg_dedup_kernel(MgParams P) {
    __shared__ uint32_t s_id[MG_MAX_IN], s_len[MG_MAX_IN];
    __shared__ unsigned long long s_off[MG_MAX_IN], s_key[MG_MAX_IN];
    __shared__ uint8_t s_drop[MG_MAX_IN];
    __shared__ uint32_t s_w[20], s_set;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_set = atomicAdd(&P.stats[MG_Q_DEDUP], 1u);
        __syncthreads();
        const uint32_t pre = s_set;
        if (pre >= P.stats[MG_N_PRE]) break;
        const uint32_t o = P.pre_off[pre], n = P.pre_off[pre + 1] - o;
        uint32_t np2 = 2;
        while (np2 < n) np2 <<= 1;
        s_id[tid] = tid < n ? P.ids[o + tid] : EMPTY32;
        // sort key: (pair index, length descending) — or, without a pair list, the contig index; ties by sequence, then index
        {
            unsigned long long key = ~0ull;
            if (tid < n) {
                const gf_contig c = P.contigs[s_id[tid]];
                uint32_t pi = 0xFFFFu;
                for (uint32_t q = 0; q < P.n_k; ++q) if (P.k_list[q] == c.k && P.kv_list[q] == c.kv) { pi = q; break; }
                key = P.n_k ? ((unsigned long long)pi << 32) | (0xFFFFFFFFu - c.length) : (unsigned long long)s_id[tid];
            }
            s_key[tid] = key;
        }
        __syncthreads();
        auto before = [&](uint32_t ia, unsigned long long ka, uint32_t ib, unsigned long long kb) {   // does (ia, ka) stand before (ib, kb)?
            if (ka != kb) return ka < kb;
            if (ia == EMPTY32 || ib == EMPTY32 || !P.n_k) return ia < ib;
            const gf_contig ca = P.contigs[ia], cb = P.contigs[ib];
            const char* sa = P.seq + ca.seq_off;
            const char* sb = P.seq + cb.seq_off;
            for (uint32_t t = 0; t < ca.length; ++t) if (sa[t] != sb[t]) return sa[t] < sb[t];      // (equal keys: equal lengths)
            return ia < ib;
        };
        for (uint32_t k = 2; k <= np2; k <<= 1)         // bitonic sort
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                if (tid < np2) {
                    const uint32_t x = tid ^ j;
                    if (x > tid) {
                        const uint32_t a = s_id[tid], b = s_id[x];
                        const unsigned long long ka = s_key[tid], kb = s_key[x];
                        const bool up = (tid & k) == 0;
                        if (before(b, kb, a, ka) == up) { s_id[tid] = b; s_id[x] = a; s_key[tid] = kb; s_key[x] = ka; }
                    }
                }
                __syncthreads();
            }
        if (tid < n) {
            const gf_contig c = P.contigs[s_id[tid]];
            s_len[tid] = c.length;
            s_off[tid] = c.seq_off;
            s_drop[tid] = 0;
        }
        __syncthreads();
        for (uint32_t q = wv; q < n; q += blockDim.x >> 6) {
            const uint32_t lq = s_len[q];
            const char* sq = P.seq + s_off[q];
            // the first min(8, lq) bases of q and of its reverse complement as one word each: ONE (unaligned) 8-byte load per offset of p
            // decides both strands; the byte loops run only behind a matching head
            const uint32_t hb = lq < 8 ? lq : 8;
            unsigned long long hf = 0, hr = 0;
            for (uint32_t t = 0; t < hb; ++t) {
                hf |= (unsigned long long)(uint8_t)sq[t] << (8 * t);
                hr |= (unsigned long long)(uint8_t)mg_comp(sq[lq - 1 - t]) << (8 * t);
            }
            const unsigned long long hmask = hb == 8 ? ~0ull : ((1ull << (8 * hb)) - 1);
            bool gone = false;
            for (uint32_t p = 0; p < n && !gone; ++p) {
                const uint32_t lp = s_len[p];
                if (p == q || lp < lq || (lp == lq && p > q) || lq == 0) continue;
                const char* sp = P.seq + s_off[p];
                for (uint32_t o0 = 0; o0 + lq <= lp && !gone; o0 += 64) {
                    const uint32_t at = o0 + lane;
                    bool hit = false;
                    if (at + lq <= lp) {
                        unsigned long long w = 0;
                        if (at + 8 <= lp) memcpy(&w, sp + at, 8);
                        else for (uint32_t t = 0; at + t < lp; ++t) w |= (unsigned long long)(uint8_t)sp[at + t] << (8 * t);
                        w &= hmask;
                        if (w == hf) {
                            uint32_t t = hb;
                            while (t < lq && sq[t] == sp[at + t]) ++t;
                            hit = t == lq;
                        }
                        if (!hit && w == hr) {
                            uint32_t t = hb;
                            while (t < lq && mg_comp(sq[lq - 1 - t]) == sp[at + t]) ++t;
                            hit = t == lq;
                        }
                    }
                    gone = __ballot(hit) != 0;
                }
            }
            if (gone && lane == 0) s_drop[q] = 1;
        }
        __syncthreads();
        const bool keep = tid < n && !s_drop[tid];
        const bool node = keep && s_len[tid] >= MG_MIN_NODE && s_len[tid] <= MG_MAX_NODE;
        uint32_t tk, tn;
        const uint32_t ek = mg_block_scan_excl(keep ? 1u : 0u, s_w, &tk);
        mg_block_scan_excl(node ? 1u : 0u, s_w, &tn);
        unsigned long long bytes = node ? s_len[tid] : 0;
        for (int d = 32; d >= 1; d >>= 1) bytes += __shfl_xor(bytes, d);
        __syncthreads();
        if (tid == 0) P.node_bytes[pre] = 0;
        __syncthreads();
        if (lane == 0 && bytes) atomicAdd(&P.node_bytes[pre], bytes);
        const uint32_t my_id = tid < n ? s_id[tid] : 0;
        __syncthreads();
        if (keep) P.ids[o + ek] = my_id;       // kept contigs, in the gap's contig order, at the front of the gap's list
        if (tid == 0) { P.kept_n[pre] = tk; P.node_n[pre] = tn; }
    }
}

__global__ __launch_bounds__(1024) void mg_scan_sets_kernel(MgParams P) {
    __shared__ uint32_t s_w[20];
    __shared__ unsigned long long s_b[20];
    const uint32_t n_pre = P.stats[MG_N_PRE];
    uint32_t carry_set = 0, carry_node = 0, skipped = 0;
    unsigned long long carry_bytes = 0;
    for (uint32_t p0 = 0; p0 < n_pre; p0 += blockDim.x) {
        const uint32_t pre = p0 + threadIdx.x;
        const uint32_t kn = pre < n_pre ? P.kept_n[pre] : 0u;
        const bool el = kn >= 2 && kn <= P.max_set;
        if (kn > P.max_set) ++skipped;
        const uint32_t nn = el ? P.node_n[pre] : 0u;
        const unsigned long long nb = el ? P.node_bytes[pre] : 0ull;
        uint32_t t1, t2;
        const uint32_t e1 = mg_block_scan_excl(el ? 1u : 0u, s_w, &t1);
        const uint32_t e2 = mg_block_scan_excl(nn, s_w, &t2);
        // 64-bit scan of the bytes
        const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        unsigned long long x = nb;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long y = __shfl_up(x, d);
            if ((int)lane >= d) x += y;
        }
        __syncthreads();
        if (lane == 63) s_b[w] = x;
        __syncthreads();
        unsigned long long bbase = 0, btot = 0;
        for (uint32_t q = 0; q < (blockDim.x >> 6); ++q) { const unsigned long long t = s_b[q]; if (q < w) bbase += t; btot += t; }
        if (el) {
            const uint32_t st = carry_set + e1;
            P.set_pre[st] = pre;
            P.set_off[st] = carry_node + e2;
            P.set_base[st] = carry_bytes + bbase + x - nb;
        }
        carry_set += t1;
        carry_node += t2;
        carry_bytes += btot;
        __syncthreads();
    }
    if (skipped) atomicAdd(&P.stats[MG_SKIPPED], skipped);
    if (threadIdx.x == 0) {
        const bool fits = carry_bytes <= P.mseq_cap && carry_node <= P.node_cap;
        if (!fits) atomicOr(&P.stats[MG_ERR], MG_E_SEQ);
        P.stats[MG_N_SETS] = fits ? carry_set : 0u;
        P.stats[MG_N_NODES] = fits ? carry_node : 0u;
        P.set_off[fits ? carry_set : 0u] = fits ? carry_node : 0u;
        P.contig_off[fits ? carry_node : 0u] = fits ? carry_bytes : 0ull;
    }
}

// one workgroup per set: the node contigs (kept contigs of node length, ascending index) back to back in mseq
__global__ __launch_bounds__(256) void mg_copy_kernel(MgParams P) {
    __shared__ uint32_t s_w[20], s_set;
    __shared__ unsigned long long s_src[128], s_dst[128];
    __shared__ uint32_t s_ln[128];
    const uint32_t tid = threadIdx.x;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_set = atomicAdd(&P.stats[MG_Q_COPY], 1u);
        __syncthreads();
        const uint32_t st = s_set;
        if (st >= P.stats[MG_N_SETS]) break;
        const uint32_t pre = P.set_pre[st], o = P.pre_off[pre], kn = P.kept_n[pre];
        const unsigned long long c0 = P.set_off[st];
        uint32_t done_nodes = 0;
        unsigned long long done_bytes = 0;
        for (uint32_t i0 = 0; i0 < kn; i0 += 128) {      // (max_set <= 128: one trip)
            const uint32_t i = i0 + tid;
            gf_contig c;
            c.length = 0; c.seq_off = 0;
            if (tid < 128 && i < kn) c = P.contigs[P.ids[o + i]];
            const bool node = tid < 128 && i < kn && c.length >= MG_MIN_NODE && c.length <= MG_MAX_NODE;
            uint32_t tn, tb;
            const uint32_t en = mg_block_scan_excl(node ? 1u : 0u, s_w, &tn);
            const uint32_t eb = mg_block_scan_excl(node ? c.length : 0u, s_w, &tb);
            if (node) {
                s_src[en] = c.seq_off; s_ln[en] = c.length;
                s_dst[en] = P.set_base[st] + done_bytes + eb;
                P.contig_off[c0 + done_nodes + en] = s_dst[en];
            }
            __syncthreads();
            for (uint32_t q = 0; q < tn; ++q) {
                const char* src = P.seq + s_src[q];
                char* dst = P.mseq + s_dst[q];
                for (uint32_t t = tid; t < s_ln[q]; t += blockDim.x) dst[t] = src[t];
            }
            done_nodes += tn;
            done_bytes += tb;
            __syncthreads();
        }
    }
}

// ---- per set: graph, components, paths
struct MgGraph {
    uint32_t M, E;
    const uint16_t* adj_off;   // [M + 1]
    const uint8_t* adj_dst;    // [E]
    const int16_t* adj_len;    // [E]  -overlap
};

__global__ __launch_bounds__(256) void mg_paths_kernel(MgParams P) {
    extern __shared__ unsigned long long s_edge[];            // [MG_MAX_EDGES]: src << 40 | i << 32 | j << 24 | dst << 16 | overlap
    __shared__ uint16_t s_adj_off[MG_MAX_NODES + 1];
    __shared__ uint8_t s_adj_dst[MG_MAX_EDGES];
    __shared__ int16_t s_adj_len[MG_MAX_EDGES];
    __shared__ int32_t s_index[MG_MAX_NODES], s_low[MG_MAX_NODES];
    __shared__ uint8_t s_on[MG_MAX_NODES], s_stack[MG_MAX_NODES], s_work_v[MG_MAX_NODES];
    __shared__ uint16_t s_work_e[MG_MAX_NODES];
    __shared__ uint16_t s_comp[MG_MAX_NODES], s_pos[MG_MAX_NODES];   // component of a node (in SCC's order), position of a node in `order`
    __shared__ uint8_t s_order[MG_MAX_NODES];
    __shared__ uint16_t s_comp_first[MG_MAX_NODES + 1];              // order[s_comp_first[c] .. s_comp_first[c + 1]) = component c, ascending
    __shared__ uint8_t s_root[MG_MAX_NODES], s_end[MG_MAX_NODES];
    __shared__ uint32_t s_n_edges, s_n_comp, s_n_roots, s_n_ends, s_n_paths, s_path_bytes, s_set, s_bad;
    __shared__ uint32_t s_poff[MG_MAX_PATHS];                        // path p = path_ws[s_poff[p] .. + s_plen[p])
    __shared__ uint16_t s_plen[MG_MAX_PATHS], s_rank[MG_MAX_PATHS];
    __shared__ uint8_t s_keep[MG_MAX_PATHS];
    const uint32_t tid = threadIdx.x;
    uint8_t* paths = P.path_ws + (size_t)blockIdx.x * MG_PATH_BYTES;
    int32_t* dist_all = P.dp_dist + (size_t)blockIdx.x * MG_MAX_NODES * MG_MAX_NODES;
    uint8_t* pred_all = P.dp_pred + (size_t)blockIdx.x * MG_MAX_NODES * MG_MAX_NODES;
    uint8_t* flag_all = P.dp_flag + (size_t)blockIdx.x * MG_MAX_NODES * MG_MAX_NODES;
    for (;;) {
        __syncthreads();
        if (tid == 0) { s_set = atomicAdd(&P.stats[MG_Q_SETS], 1u); s_n_edges = 0; s_bad = 0; s_n_paths = 0; s_path_bytes = 0; }
        __syncthreads();
        const uint32_t st = s_set;
        if (st >= P.stats[MG_N_SETS]) break;
        if (tid < 2) P.set_jobs[2 * st + tid] = 0;
        const uint32_t M = 2 * (uint32_t)(P.set_off[st + 1] - P.set_off[st]);
        const uint32_t pb = P.set_range[2 * st], pn = P.set_range[2 * st + 1];
        if (M < 4 || pn == 0 || pb + pn > P.pair_cap) { if (pn && pb + pn > P.pair_cap && tid == 0) atomicOr(&P.stats[MG_ERR], MG_E_PAIRS); continue; }
        // edges: a pair with a significant overlap that is no containment (threadMergeContigV2, ContigsCompactor.cpp:656-674)
        for (uint32_t x = tid; x < pn; x += blockDim.x) {
            const gf_ovl_result r = P.res[pb + x];
            if (r.res != 2 || r.containment) continue;
            const gf_qcpair q = P.pairs[pb + x];
            const uint32_t src = r.first_goes_first ? q.i : q.j, dst = r.first_goes_first ? q.j : q.i;
            const uint32_t at = atomicAdd(&s_n_edges, 1u);
            if (at < MG_MAX_EDGES)
                s_edge[at] = ((unsigned long long)src << 40) | ((unsigned long long)q.i << 32) | ((unsigned long long)q.j << 24) |
                             ((unsigned long long)dst << 16) | (unsigned long long)(uint32_t)(r.overlap & 0xFFFF);
        }
        __syncthreads();
        const uint32_t E = s_n_edges;
        if (E > MG_MAX_EDGES) { if (tid == 0) atomicAdd(&P.stats[MG_SKIPPED_GRAPH], 1u); continue; }
        if (E == 0) continue;
        if (tid == 0) atomicAdd(&P.stats[MG_N_EDGES], E);
        // adjacency lists in the order the reference adds the edges: pairs (i, j) ascending -> sort by (src, i, j)
        uint32_t np2 = 2;
        while (np2 < E) np2 <<= 1;
        for (uint32_t x = E + tid; x < np2; x += blockDim.x) s_edge[x] = ~0ull;
        __syncthreads();
        for (uint32_t k = 2; k <= np2; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t t = tid; t < np2; t += blockDim.x) {
                    const uint32_t x = t ^ j;
                    if (x > t) {
                        const unsigned long long a = s_edge[t], b = s_edge[x];
                        const bool up = (t & k) == 0;
                        if ((a > b) == up) { s_edge[t] = b; s_edge[x] = a; }
                    }
                }
                __syncthreads();
            }
        for (uint32_t v = tid; v <= M; v += blockDim.x) s_adj_off[v] = 0;
        __syncthreads();
        for (uint32_t x = tid; x < E; x += blockDim.x) {
            const unsigned long long e = s_edge[x];
            s_adj_dst[x] = (uint8_t)(e >> 16);
            s_adj_len[x] = (int16_t)(-(int)(e & 0xFFFF));
            const uint32_t src = (uint32_t)(e >> 40);
            if (x == 0 || (uint32_t)(s_edge[x - 1] >> 40) != src) s_adj_off[src] = (uint16_t)x;     // first edge of src (filled in below for nodes without edges)
        }
        __syncthreads();
        if (tid == 0) {
            // offsets: a node without edges takes the offset of the next node that has some
            s_adj_off[M] = (uint16_t)E;
            // mark nodes that have edges
            for (uint32_t v = 0; v < M; ++v) s_on[v] = 0;
            for (uint32_t x = 0; x < E; ++x) s_on[(uint32_t)(s_edge[x] >> 40)] = 1;
            uint16_t nxt = (uint16_t)E;
            for (int v = (int)M - 1; v >= 0; --v) { if (s_on[v]) nxt = s_adj_off[v]; else s_adj_off[v] = nxt; }
            // ---- strongly connected components: Tarjan from node 0 up, neighbours in edge order; AbstractGraph::SCC returns them reversed
            // (= topological), every component sorted (MergeContigs._components)
            for (uint32_t v = 0; v < M; ++v) { s_index[v] = -1; s_low[v] = 0; s_on[v] = 0; }
            uint32_t sp = 0, n_comp = 0;
            int counter = 1;
            // components are emitted in completion order into s_order from the BACK (so that the reversed list reads front to back)
            uint32_t back = M;
            for (uint32_t root = 0; root < M; ++root) {
                if (s_index[root] >= 0) continue;
                uint32_t wp = 0;
                s_work_v[0] = (uint8_t)root; s_work_e[0] = s_adj_off[root]; wp = 1;
                s_index[root] = s_low[root] = counter++;
                s_stack[sp++] = (uint8_t)root; s_on[root] = 1;
                while (wp) {
                    const uint32_t v = s_work_v[wp - 1];
                    const uint32_t ei = s_work_e[wp - 1];
                    if (ei < s_adj_off[v + 1]) {
                        s_work_e[wp - 1] = (uint16_t)(ei + 1);
                        const uint32_t w = s_adj_dst[ei];
                        if (s_index[w] < 0) {
                            s_index[w] = s_low[w] = counter++;
                            s_stack[sp++] = (uint8_t)w; s_on[w] = 1;
                            s_work_v[wp] = (uint8_t)w; s_work_e[wp] = s_adj_off[w]; ++wp;
                        } else if (s_on[w]) {
                            if (s_index[w] < s_low[v]) s_low[v] = s_index[w];
                        }
                        continue;
                    }
                    --wp;
                    if (wp) { const uint32_t u = s_work_v[wp - 1]; if (s_low[v] < s_low[u]) s_low[u] = s_low[v]; }
                    if (s_low[v] == s_index[v]) {
                        uint32_t cn = 0;
                        for (;;) {
                            const uint32_t w = s_stack[--sp];
                            s_on[w] = 0;
                            s_order[--back] = (uint8_t)w;
                            s_comp[w] = (uint16_t)n_comp;      // completion number; turned into the reversed numbering below
                            ++cn;
                            if (w == v) break;
                        }
                        // sort the component's nodes ascending (insertion sort: components are small)
                        for (uint32_t a = back + 1; a < back + cn; ++a) {
                            const uint8_t key = s_order[a];
                            uint32_t b = a;
                            while (b > back && s_order[b - 1] > key) { s_order[b] = s_order[b - 1]; --b; }
                            s_order[b] = key;
                        }
                        ++n_comp;
                    }
                }
            }
            // (every node is in exactly one component: back == 0 here).  Reversed component numbering + component starts
            for (uint32_t v = 0; v < M; ++v) s_comp[v] = (uint16_t)(n_comp - 1 - s_comp[v]);
            uint32_t cprev = EMPTY32, nc = 0;
            for (uint32_t x = 0; x < M; ++x) {
                s_pos[s_order[x]] = (uint16_t)x;
                if (s_comp[s_order[x]] != cprev) { s_comp_first[nc++] = (uint16_t)x; cprev = s_comp[s_order[x]]; }
            }
            s_comp_first[nc] = (uint16_t)M;
            s_n_comp = nc;
            // ---- roots and ends (FindSimplePathsTopSortStart; MergeContigs._terminals).  s_root / s_end: candidate flags first
            for (uint32_t v = 0; v < M; ++v) { s_root[v] = 1; s_end[v] = 1; }
            for (uint32_t v = 0; v < M; ++v)
                for (uint32_t e = s_adj_off[v]; e < s_adj_off[v + 1]; ++e) {
                    const uint32_t w = s_adj_dst[e];
                    if (s_comp[w] != s_comp[v]) { s_root[w] = 0; s_end[v] = 0; }
                }
            for (uint32_t c = 0; c < nc; ++c) {
                const uint32_t a = s_comp_first[c], b = s_comp_first[c + 1];
                if (b - a < 2) continue;
                bool whole_r = true, whole_e = true;
                for (uint32_t x = a; x < b; ++x) { whole_r = whole_r && s_root[s_order[x]]; whole_e = whole_e && s_end[s_order[x]]; }
                for (uint32_t x = a; x < b; ++x) {
                    const uint32_t v = s_order[x];
                    s_root[v] = (x == a) && whole_r;          // the component's first node, and only when all of them qualify
                    s_end[v] = (x == b - 1) && whole_e;       // ... its last node
                }
            }
            // sorted lists of the roots and of the ends (node order), in place of the flags
            uint32_t nr = 0, ne = 0;
            for (uint32_t v = 0; v < M; ++v) { if (s_root[v]) s_work_v[nr++] = (uint8_t)v; }
            for (uint32_t v = 0; v < M; ++v) { if (s_end[v]) s_stack[ne++] = (uint8_t)v; }
            for (uint32_t x = 0; x < nr; ++x) s_root[x] = s_work_v[x];
            for (uint32_t x = 0; x < ne; ++x) s_end[x] = s_stack[x];
            s_n_roots = nr; s_n_ends = ne;
        }
        __syncthreads();
        const uint32_t n_roots = s_n_roots, n_ends = s_n_ends;
        // ---- per root (one thread each): shortest-path DP along `order` with -overlap as edge length; an entry = (distance, predecessor,
        // "ends with its node twice": an edge from a node to itself improved its own entry while it was being processed)
        for (uint32_t ri = tid; ri < n_roots; ri += blockDim.x) {
            const uint32_t root = s_root[ri];
            int32_t* dist = dist_all + (size_t)ri * MG_MAX_NODES;
            uint8_t* pred = pred_all + (size_t)ri * MG_MAX_NODES;
            uint8_t* flag = flag_all + (size_t)ri * MG_MAX_NODES;
            for (uint32_t x = 0; x < M; ++x) flag[x] = 0;
            const uint32_t p0 = s_pos[root];
            dist[p0] = 0; pred[p0] = (uint8_t)p0; flag[p0] = 1;
            for (uint32_t i = p0; i < M; ++i) {
                if (!(flag[i] & 1)) continue;
                const int32_t d = dist[i];
                const uint32_t v = s_order[i];
                for (uint32_t e = s_adj_off[v]; e < s_adj_off[v + 1]; ++e) {
                    const uint32_t w = s_adj_dst[e], j = s_pos[w];
                    if (j < i) continue;
                    // GetEdgeTo: the FIRST edge to w counts (no two edges join the same nodes here; kept for the reference's rule)
                    int32_t len = s_adj_len[e];
                    for (uint32_t e2 = s_adj_off[v]; e2 < e; ++e2) if (s_adj_dst[e2] == w) { len = s_adj_len[e2]; break; }
                    if (!(flag[j] & 1) || d + len < dist[j]) {
                        dist[j] = d + len;
                        if (j == i) flag[j] = 3;                  // path + (v,): the entry now ends with v twice
                        else { pred[j] = (uint8_t)i; flag[j] = 1; }
                    }
                }
            }
            // the root's paths: one per reachable end (ends ascending), the MG_PER_ROOT longest of them (ties: the earlier end)
            uint32_t last_len = 0xFFFFFFFFu, last_q = 0;
            for (uint32_t round = 0; round < MG_PER_ROOT; ++round) {
                // next in (-len, q) order after (last_len, last_q)
                uint32_t best_len = 0, best_q = EMPTY32;
                for (uint32_t q = 0; q < n_ends; ++q) {
                    const uint32_t pe = s_pos[s_end[q]];
                    if (pe < p0 || !(flag[pe] & 1)) continue;
                    uint32_t len = 1, x = pe;
                    while (x != p0 && len <= MG_MAX_NODES) { x = pred[x]; ++len; }
                    if (flag[pe] & 2) ++len;
                    const bool after = round == 0 || len < last_len || (len == last_len && q > last_q);
                    if (!after) continue;
                    if (best_q == EMPTY32 || len > best_len) { best_len = len; best_q = q; }
                }
                if (best_q == EMPTY32) break;
                last_len = best_len; last_q = best_q;
                const uint32_t slot = atomicAdd(&s_n_paths, 1u);
                const uint32_t off = atomicAdd(&s_path_bytes, best_len);
                if (slot >= MG_MAX_PATHS || off + best_len > MG_PATH_BYTES) { s_bad = 1; break; }
                s_poff[slot] = off; s_plen[slot] = (uint16_t)best_len;
                uint32_t w = best_len, x = s_pos[s_end[best_q]];
                if (flag[x] & 2) paths[off + --w] = s_order[x];
                for (;;) { paths[off + --w] = s_order[x]; if (x == p0 || w == 0) break; x = pred[x]; }
            }
        }
        __syncthreads();
        if (s_bad) { if (tid == 0) atomicAdd(&P.stats[MG_SKIPPED_GRAPH], 1u); continue; }
        const uint32_t NP = s_n_paths;
        if (NP == 0) continue;
        __threadfence_block();
        auto less = [&](uint32_t a, uint32_t b) {      // tuple order of the node lists
            const uint8_t* pa = paths + s_poff[a];
            const uint8_t* pbp = paths + s_poff[b];
            const uint32_t la = s_plen[a], lb = s_plen[b], m = la < lb ? la : lb;
            for (uint32_t t = 0; t < m; ++t) if (pa[t] != pbp[t]) return pa[t] < pbp[t];
            return la < lb;
        };
        // rank of every path in sorted order (paths of different roots differ; a root's paths end at different nodes: no duplicates)
        for (uint32_t a = tid; a < NP; a += blockDim.x) {
            uint32_t r = 0;
            for (uint32_t b = 0; b < NP; ++b) if (b != a && less(b, a)) ++r;
            s_rank[a] = (uint16_t)r;
        }
        __syncthreads();
        // RemoveDupRevCompPaths: a path goes when its twin (nodes reversed, strands flipped) stands before it in sorted order
        for (uint32_t a = tid; a < NP; a += blockDim.x) {
            const uint8_t* pa = paths + s_poff[a];
            const uint32_t la = s_plen[a];
            bool gone = false;
            for (uint32_t b = 0; b < NP && !gone; ++b) {
                if (s_plen[b] != la || s_rank[b] >= s_rank[a]) continue;
                const uint8_t* pbp = paths + s_poff[b];
                bool twin = true;
                for (uint32_t t = 0; t < la && twin; ++t) twin = pbp[t] == (uint8_t)(pa[la - 1 - t] ^ 1u);
                gone = twin;
            }
            s_keep[a] = !gone && la > 1;
        }
        __syncthreads();
        // jobs of the set in sorted order: contiguous in the job list
        if (tid == 0) {
            uint32_t nj = 0, nb = 0;
            for (uint32_t a = 0; a < NP; ++a) if (s_keep[a]) { ++nj; nb += s_plen[a]; }
            uint32_t jb = 0, bb = 0;
            if (nj) { jb = atomicAdd(&P.stats[MG_N_JOBS], nj); bb = atomicAdd(&P.stats[MG_JOB_NODES], nb); }
            // (no room in the job list — a repeat-bearing draft with thousands of paths per gap: the set is left alone and counted; its
            // reserved slots stay EMPTY and the string kernel passes over them)
            if (nj && (jb + nj > P.job_cap || bb + nb > P.job_node_cap)) { atomicAdd(&P.stats[MG_SKIPPED_GRAPH], 1u); nj = 0; }
            s_n_edges = jb; s_n_comp = bb; s_n_roots = nj;      // (reused as broadcast slots)
            if (nj) { P.set_jobs[2 * st] = jb; P.set_jobs[2 * st + 1] = nj; atomicAdd(&P.stats[MG_SETS_WITH_JOBS], 1u); }
        }
        __syncthreads();
        if (s_n_roots == 0) continue;
        const uint32_t jb = s_n_edges, bb = s_n_comp;
        // position of a kept path among the kept ones in sorted order, and its byte offset
        for (uint32_t a = tid; a < NP; a += blockDim.x) {
            if (!s_keep[a]) continue;
            uint32_t r = 0, ob = 0;
            for (uint32_t b = 0; b < NP; ++b) if (s_keep[b] && s_rank[b] < s_rank[a]) { ++r; ob += s_plen[b]; }
            MgJob j; j.set = st; j.off = bb + ob; j.len = s_plen[a];
            P.jobs[jb + r] = j;
            for (uint32_t t = 0; t < s_plen[a]; ++t) P.job_nodes[bb + ob + t] = paths[s_poff[a] + t];
        }
    }
}

// contig records of the sets' jobs: set order, inside a set the sorted path order (NEW_CONTIG_MERGE_1, _2, ...)
__global__ __launch_bounds__(1024) void mg_scan_jobs_kernel(MgParams P) {
    __shared__ uint32_t s_w[20];
    const uint32_t n_sets = P.stats[MG_N_SETS], n0 = P.stats[MG_N0];
    uint32_t carry = 0;
    for (uint32_t s0 = 0; s0 < n_sets; s0 += blockDim.x) {
        const uint32_t st = s0 + threadIdx.x;
        const uint32_t nj = st < n_sets ? P.set_jobs[2 * st + 1] : 0u;
        uint32_t tot;
        const uint32_t ex = mg_block_scan_excl(nj, s_w, &tot);
        if (st < n_sets) P.set_rec[st] = n0 + carry + ex;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if ((unsigned long long)n0 + carry > P.contig_cap) atomicOr(&P.stats[MG_ERR], MG_E_CONTIGS);
        *P.n_contigs = n0 + carry;       // (beyond the capacity: the caller's overflow test sees it; the records beyond are not written)
    }
}

// FormMergedSeqFromPath (ContigsCompactor.cpp:1456-1520; MergeContigs.merged_strings): one workgroup per path
__global__ __launch_bounds__(OV_NT) void mg_strings_kernel(MgParams P) {
    extern __shared__ uint32_t sm[];
    __shared__ uint32_t s_job;
    __shared__ long long s_best_sc[OV_NT / 64];
    __shared__ unsigned long long s_best_rk[OV_NT / 64];
    __shared__ gf_ovl_result s_res;
    __shared__ unsigned long long s_out;
    const OvLds S(sm);
    const uint32_t tid = threadIdx.x;
    char* cur = P.cur_ws + (size_t)blockIdx.x * 32768;
    char* nxt = cur + 16384;
    gf_ovl_params pr = P.pr;
    pr.relax = 1.0;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_job = atomicAdd(&P.stats[MG_Q_JOBS], 1u);
        __syncthreads();
        const uint32_t ji = s_job;
        const uint32_t n_jobs = P.stats[MG_N_JOBS] < P.job_cap ? P.stats[MG_N_JOBS] : P.job_cap;
        if (ji >= n_jobs) break;
        const MgJob job = P.jobs[ji];
        const uint32_t st = job.set;
        if (st == EMPTY32 || P.set_jobs[2 * st + 1] == 0) continue;      // (a slot of a reservation that did not fit)
        const unsigned long long c0 = P.set_off[st];
        const uint8_t* path = P.job_nodes + job.off;
        auto node_src = [&](uint32_t v, int* n) { const unsigned long long o = P.contig_off[c0 + (v >> 1)]; *n = (int)(P.contig_off[c0 + (v >> 1) + 1] - o); return P.mseq + o; };
        int n1;
        {
            const char* src = node_src(path[0], &n1);
            const bool rc = path[0] & 1;
            for (int i = (int)tid; i < n1; i += OV_NT) { const char ch = src[rc ? n1 - 1 - i : i]; cur[i] = rc ? mg_comp(ch) : ch; }
        }
        __syncthreads();
        for (uint32_t step = 1; step < job.len; ++step) {
            int n2;
            const char* src2 = node_src(path[step], &n2);
            if (n1 > (int)MG_MAX_NODE || n2 > (int)MG_MAX_NODE) break;      // grown beyond the kernel's reach: the path ends here
            for (int i = (int)tid; i < n1; i += OV_NT) S.a[i] = cur[i];
            ov_load_node(S.b, src2, n2, path[step] & 1);
            ov_evaluate<OV_NT>(S, n1, n2, pr, s_best_sc, s_best_rk, &s_res);
            const gf_ovl_result r = s_res;
            const int re = r.row_end, ce = r.col_end, nc = r.nclip;
            int nn;
            if (r.contained && re + nc == n1 && n1 < n2) {                 // the running string lies inside the node: the node
                nn = n2;
                for (int i = (int)tid; i < nn; i += OV_NT) nxt[i] = S.b[i];
            } else if (r.contained && ce + nc == n2 && n2 < n1) {          // the node lies inside the running string: unchanged
                nn = n1;
                for (int i = (int)tid; i < nn; i += OV_NT) nxt[i] = S.a[i];
            } else if (re + nc == n1) {                                    // SetMergedStringConcat, MODE_1_2
                nn = (n1 - nc) + (n2 - ce);
                for (int i = (int)tid; i < nn; i += OV_NT) nxt[i] = i < n1 - nc ? S.a[i] : S.b[ce + (i - (n1 - nc))];
            } else {                                                       // MODE_2_1
                nn = (n2 - nc) + (n1 - re);
                for (int i = (int)tid; i < nn; i += OV_NT) nxt[i] = i < n2 - nc ? S.b[i] : S.a[re + (i - (n2 - nc))];
            }
            __syncthreads();
            char* t = cur; cur = nxt; nxt = t;
            n1 = nn;
        }
        // the merged string becomes a contig record of the set's gap (k = kv = 0: a merged contig)
        const uint32_t rec = P.set_rec[st] + (ji - P.set_jobs[2 * st]);
        if (tid == 0) s_out = atomicAdd(P.seq_len, (unsigned long long)n1);
        __syncthreads();
        const unsigned long long so = s_out;
        const bool fits = so + (unsigned long long)n1 <= P.seq_cap;
        if (!fits && tid == 0) atomicOr(&P.stats[MG_ERR], MG_E_OUTSEQ);
        if (fits) for (int i = (int)tid; i < n1; i += OV_NT) P.seq[so + i] = cur[i];
        if (tid == 0 && rec < P.contig_cap) {
            gf_contig c;
            memset(&c, 0, sizeof c);
            c.gap = P.pre_gap[P.set_pre[st]];
            c.k = 0; c.kv = 0;
            c.n_nodes = job.len;
            c.length = fits ? (uint32_t)n1 : 0u;
            c.cov_sum = 0;
            c.seq_off = fits ? so : 0ull;
            P.contigs[rec] = c;
        }
        __syncthreads();
    }
}

// ---- host side of the merge round
static inline size_t mg_align(size_t x) { return (x + 255) & ~(size_t)255; }

int launch_merge_round(gf_ctx* ctx, void* d_contigs, void* d_n_contigs, size_t contig_cap, void* d_seq, void* d_seq_len, size_t seq_cap,
                       const void* d_gap_best, size_t n_gaps, const gf_ovl_params* params, int kq, int max_set, const int* k_list, const int* kv_list,
                       int n_k, void* d_stats) {
    const unsigned grid = (unsigned)ctx->n_cu;
    const size_t ng = n_gaps;
    const size_t node_cap = contig_cap;
    const size_t mseq_cap = std::min<size_t>(seq_cap, (size_t)512 << 20);
    const size_t pair_cap = std::min<size_t>((size_t)4 << 20, std::max<size_t>(65536, 16 * contig_cap));
    const size_t job_cap = std::max<size_t>(65536, 16 * ng), job_node_cap = std::max<size_t>((size_t)4 << 20, 512 * ng);
    // one workspace, carved
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t o = at; at += mg_align(bytes); return o; };
    const size_t o_cnt = take(ng * 4), o_pre_of = take(ng * 4), o_pre_gap = take(ng * 4), o_pre_off = take((ng + 1) * 4), o_ids = take(contig_cap * 4),
                 o_kept = take(ng * 4), o_noden = take(ng * 4), o_nodeb = take(ng * 8), o_set_pre = take(ng * 4), o_set_base = take(ng * 8),
                 o_coff = take((node_cap + 1) * 8), o_soff = take((ng + 1) * 8), o_mseq = take(mseq_cap + 64), o_pairs = take(pair_cap * sizeof(gf_qcpair)),
                 o_res = take(pair_cap * sizeof(gf_ovl_result)), o_range = take(ng * 8), o_paths = take((size_t)grid * MG_PATH_BYTES),
                 o_dist = take((size_t)grid * MG_MAX_NODES * MG_MAX_NODES * 4), o_pred = take((size_t)grid * MG_MAX_NODES * MG_MAX_NODES),
                 o_flag = take((size_t)grid * MG_MAX_NODES * MG_MAX_NODES), o_jobs = take(job_cap * sizeof(MgJob)), o_jnodes = take(job_node_cap),
                 o_sjobs = take(ng * 8), o_srec = take(ng * 4), o_cur = take((size_t)grid * 32768), o_qcmat = take((size_t)grid * (((size_t)MG_MAX_NODES * MG_MAX_NODES + 31) / 32 + 1) * 4);
    int rc;
    if ((rc = ensure(ctx, ctx->merge_ws, at + 256))) return rc;
    if ((rc = ensure(ctx, ctx->counters, GF_COUNTER_BYTES))) return rc;
    uint8_t* W = (uint8_t*)ctx->merge_ws.p;
    MgParams P;
    memset(&P, 0, sizeof P);
    P.contigs = (gf_contig*)d_contigs; P.n_contigs = (uint32_t*)d_n_contigs; P.contig_cap = (uint32_t)contig_cap;
    P.seq = (char*)d_seq; P.seq_len = (unsigned long long*)d_seq_len; P.seq_cap = seq_cap;
    P.gap_best = (const unsigned long long*)d_gap_best; P.n_gaps = (uint32_t)ng; P.max_set = (uint32_t)max_set;
    P.stats = (uint32_t*)d_stats;
    P.cnt = (uint32_t*)(W + o_cnt); P.pre_of_gap = (uint32_t*)(W + o_pre_of); P.pre_gap = (uint32_t*)(W + o_pre_gap); P.pre_off = (uint32_t*)(W + o_pre_off);
    P.ids = (uint32_t*)(W + o_ids); P.kept_n = (uint32_t*)(W + o_kept); P.node_n = (uint32_t*)(W + o_noden); P.node_bytes = (unsigned long long*)(W + o_nodeb);
    P.set_pre = (uint32_t*)(W + o_set_pre); P.set_base = (unsigned long long*)(W + o_set_base); P.contig_off = (unsigned long long*)(W + o_coff);
    P.set_off = (unsigned long long*)(W + o_soff); P.node_cap = (uint32_t)node_cap; P.mseq = (char*)(W + o_mseq); P.mseq_cap = mseq_cap;
    P.pairs = (const gf_qcpair*)(W + o_pairs); P.res = (const gf_ovl_result*)(W + o_res); P.pair_cap = (uint32_t)pair_cap;
    P.set_range = (const uint32_t*)(W + o_range); P.path_ws = W + o_paths; P.dp_dist = (int32_t*)(W + o_dist); P.dp_pred = W + o_pred; P.dp_flag = W + o_flag;
    P.jobs = (MgJob*)(W + o_jobs); P.job_cap = (uint32_t)job_cap; P.job_nodes = W + o_jnodes; P.job_node_cap = (uint32_t)job_node_cap;
    P.set_jobs = (uint32_t*)(W + o_sjobs); P.set_rec = (uint32_t*)(W + o_srec); P.cur_ws = (char*)(W + o_cur);
    P.pr = *params;
    P.pr.relax = 0.0;
    P.n_k = (uint32_t)n_k;
    for (int q = 0; q < n_k; ++q) { P.k_list[q] = (uint16_t)k_list[q]; P.kv_list[q] = (uint16_t)kv_list[q]; }
    LaunchTimer tm(ctx, GF_KERNEL_MERGE);
    GF_HIP(ctx, hipMemsetAsync(d_stats, 0, MG_WORDS * 4, ctx->stream));
    GF_HIP(ctx, hipMemsetAsync(P.cnt, 0, ng * 4, ctx->stream));
    GF_HIP(ctx, hipMemsetAsync(P.jobs, 0xFF, job_cap * sizeof(MgJob), ctx->stream));
    uint32_t* d_next_qc = (uint32_t*)ctx->counters.p + 9;
    uint32_t* d_next_ov = (uint32_t*)ctx->counters.p + 11;
    zero_regions(ctx, ZeroList{{d_next_qc, d_next_ov, nullptr, nullptr}, {1, 1, 0, 0}});
    const unsigned cgrid = (unsigned)std::min<size_t>((contig_cap + 255) / 256, (size_t)ctx->n_cu * 8);
    hipLaunchKernelGGL(mg_count_kernel, dim3(cgrid), dim3(256), 0, ctx->stream, P);
    hipLaunchKernelGGL(mg_scan_gaps_kernel, dim3(1), dim3(1024), 0, ctx->stream, P);
    hipLaunchKernelGGL(mg_fill_kernel, dim3(cgrid), dim3(256), 0, ctx->stream, P);
    hipLaunchKernelGGL(mg_dedup_kernel, dim3(grid), dim3(1024), 0, ctx->stream, P);
    hipLaunchKernelGGL(mg_scan_sets_kernel, dim3(1), dim3(1024), 0, ctx->stream, P);
    hipLaunchKernelGGL(mg_copy_kernel, dim3(grid), dim3(256), 0, ctx->stream, P);
    {   // the merger's prefilter over the sets (their number is a device word)
        QcParams Q;
        Q.seq = P.mseq; Q.contig_off = P.contig_off; Q.set_off = P.set_off;
        Q.n_sets = (uint32_t)ng; Q.k = (uint32_t)kq; Q.max_nodes = MG_MAX_NODES;
        Q.words = (uint32_t)(((uint64_t)MG_MAX_NODES * MG_MAX_NODES + 31) / 32 + 1);
        Q.matrix = (uint32_t*)(W + o_qcmat);
        Q.out = (gf_qcpair*)(W + o_pairs); Q.cap = (uint32_t)pair_cap; Q.n_out = P.stats + MG_N_PAIRS;
        Q.next_set = d_next_qc; Q.error = P.stats + MG_QC_FLAGS; Q.d_n_sets = P.stats + MG_N_SETS; Q.set_range = (uint32_t*)(W + o_range);
        hipLaunchKernelGGL(quick_check_kernel, dim3(grid), dim3(1024), (size_t)QC_SLOTS * 8, ctx->stream, Q);
    }
    {   // ... and its overlap evaluation = the edges
        OvParams O;
        O.seq = P.mseq; O.contig_off = P.contig_off; O.set_off = P.set_off; O.pairs = P.pairs; O.n_pairs = (uint32_t)pair_cap; O.pr = P.pr;
        O.out = (gf_ovl_result*)(W + o_res); O.next = d_next_ov; O.d_n_pairs = P.stats + MG_N_PAIRS;
        hipLaunchKernelGGL(overlap_eval_kernel, dim3(grid), dim3(OV_NT), OV_LDS_BYTES, ctx->stream, O);
    }
    hipLaunchKernelGGL(mg_paths_kernel, dim3(grid), dim3(256), (size_t)MG_MAX_EDGES * 8, ctx->stream, P);
    hipLaunchKernelGGL(mg_scan_jobs_kernel, dim3(1), dim3(1024), 0, ctx->stream, P);
    hipLaunchKernelGGL(mg_strings_kernel, dim3(grid), dim3(OV_NT), OV_LDS_BYTES, ctx->stream, P);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
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
    P.d_n_sets = nullptr;
    P.set_range = nullptr;
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
    P.d_n_pairs = nullptr;
    const size_t lds = OV_LDS_BYTES;
    {
        LaunchTimer tm(ctx, GF_KERNEL_MERGE);
        hipLaunchKernelGGL(overlap_eval_kernel, dim3((unsigned)std::min<size_t>(n_pairs, (size_t)ctx->n_cu)), dim3(OV_NT), lds, ctx->stream, P);
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

int gf_merge_open_gaps_dev(gf_ctx* ctx, void* d_contigs, void* d_n_contigs, size_t contig_cap, void* d_seq, void* d_seq_len, size_t seq_cap,
                           const void* d_gap_best, size_t n_gaps, const gf_ovl_params* params, int kmer_len_quick, int max_set, const int* k_list,
                           const int* kv_list, int n_k, void* d_stats) {
    if (n_k < 0 || n_k > 16 || (n_k && (!k_list || !kv_list))) return GF_E_INVAL;
    if (!ctx || !d_contigs || !d_n_contigs || !d_seq || !d_seq_len || !d_gap_best || !params || !d_stats || contig_cap > 0x7FFFFFFFull ||
        n_gaps > 0xFFFFFFF0ull || kmer_len_quick < 4 || kmer_len_quick > 16 || max_set < 2 || max_set > (int)(MG_MAX_NODES / 2))
        return GF_E_INVAL;
    if (params->indel != (double)(int)params->indel || params->max_clip < 0 || params->max_clip > 1e6) {
        ctx->last_error = "gf_merge_open_gaps_dev: the indel score must be integral";
        return GF_E_UNSUPPORTED;
    }
    GF_HIP(ctx, hipSetDevice(ctx->device));
    if (!n_gaps) { GF_HIP(ctx, hipMemsetAsync(d_stats, 0, MG_WORDS * 4, ctx->stream)); return GF_OK; }
    return launch_merge_round(ctx, d_contigs, d_n_contigs, contig_cap, d_seq, d_seq_len, seq_cap, d_gap_best, n_gaps, params, kmer_len_quick, max_set,
                              k_list, kv_list, n_k, d_stats);
}

}  // extern "C"
