#define _POSIX_C_SOURCE 199309L
/* TEST INFRASTRUCTURE — see gp_oracle.h.  Each function cites the reference file:line it restates
 * (paths relative to /root/reference). */
#include "gp_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void or_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------ a-7: KmerUtils layout */
static inline unsigned code_of(char c) { /* KmerUtils.cpp:25-41: non-CGT -> A */
    return (c == 'C' || c == 'c') ? 1u : (c == 'G' || c == 'g') ? 2u : (c == 'T' || c == 't') ? 3u : 0u;
}

uint64_t or_pack_kmer64(const char* seq, int k) { /* KmerUtils.cpp:61-69 */
    uint64_t v = 0;
    for (int i = 0; i < k; ++i) v |= (uint64_t)code_of(seq[i]) << (62 - 2 * i);
    return v;
}

void or_unpack_reads(const uint8_t* packed, size_t n_reads, int read_len, char* ascii) {
    size_t rb = (size_t)(read_len + 3) / 4;
    for (size_t r = 0; r < n_reads; ++r)
        for (int i = 0; i < read_len; ++i)
            ascii[r * read_len + i] = "ACGT"[(packed[r * rb + (i >> 2)] >> (6 - 2 * (i & 3))) & 3];
}

/* ------------------------------------------------------------------ a-2: tagger */
static int put(or_taghit* out, size_t cap, size_t* n, uint32_t rec, uint32_t gap, int kind, int to_mate) {
    if (*n < cap) { out[*n].rec = rec; out[*n].gap = gap; out[*n].kind = (uint16_t)kind; out[*n].to_mate = (uint16_t)to_mate; }
    ++*n;
    return 0;
}

/* gaps of one scaffold: `by_scaf` lists gap indices grouped by scaffold (each group in the caller's gap order — the order
 * the reference reads gap_positions.txt in, :36-63), scaf_off[s] .. scaf_off[s+1] is scaffold s's group.  Same lines as a scan
 * over all gaps that skips the other scaffolds, without the O(records x gaps) cost at 20 000 gaps. */
static size_t tag_range(const or_alnrec* recs, size_t i0, size_t i1, const or_gap* gaps, const uint32_t* by_scaf,
                        const size_t* scaf_off, size_t n_scaf, int insert_size, int sd, int clip_dist, int anchor_mapq,
                        or_taghit* out, size_t cap) {
    const long dist1 = insert_size - 3L * sd, dist2 = insert_size + 3L * sd; /* collect_reads_for_gaps.py:5-6 */
    const int short_is = insert_size < 750;                                  /* :275 */
    size_t cnt = 0;
    for (size_t i = i0; i < i1; ++i) {
        const or_alnrec* r = &recs[i];
        if (r->ref >= n_scaf) continue;
        for (size_t q = scaf_off[r->ref]; q < scaf_off[r->ref + 1]; ++q) { /* every gap of the record's scaffold (:36-63) */
            const size_t g = by_scaf[q];
            const long start = gaps[g].start, end = gaps[g].end, pos = r->pos;
            int tag = -1; /* 0:0c 1:0d 2:1c 3:1d */
            /* focal_region keys start-i, i in range(dist2), start-i >= 0 (:47-55); end+i (:57-62) */
            if (start - pos >= 0 && start - pos < dist2 && pos >= 0) tag = (start - pos) <= clip_dist ? 0 : 1;
            else if (pos - end >= 0 && pos - end < dist2) tag = (pos - end) <= clip_dist ? 2 : 3;
            if (tag < 0) continue;
            if ((tag == 0 && r->clipflag >= 2) || (tag == 2 && (r->clipflag == 1 || r->clipflag == 3))) /* :119 */
                put(out, cap, &cnt, (uint32_t)i, (uint32_t)g, 0, 0);
            if ((r->flag & 0x4) == 0 && (r->flag & 0x8) == 0 && (int)r->mapq >= anchor_mapq) { /* :126 */
                if (r->mate_ref != r->ref) put(out, cap, &cnt, (uint32_t)i, (uint32_t)g, 1, 1); /* :127-133 */
                else {
                    long t = r->tlen < 0 ? -(long)r->tlen : (long)r->tlen;
                    if (t >= dist2 || (short_is && t <= dist1)) put(out, cap, &cnt, (uint32_t)i, (uint32_t)g, 1, 1); /* :143 / :243 */
                }
            } else if ((r->flag & 0x4) == 0 && (r->flag & 0x8) != 0) { /* :153 */
                put(out, cap, &cnt, (uint32_t)i, (uint32_t)g, 2, 1);
            }
        }
    }
    return cnt;
}

/* records are independent (the reference runs one pipeline per scaffold, run_multi_threads_collect_reads.py:35-38):
 * contiguous chunks per OpenMP thread, concatenated in record order */
size_t or_tag_alignments(const or_alnrec* recs, size_t n, const or_gap* gaps, size_t n_gaps, int insert_size, int sd,
                         int clip_dist, int anchor_mapq, or_taghit* out, size_t cap) {
    int nt = 1;
#ifdef _OPENMP
    nt = omp_get_max_threads();
#endif
    if (nt > 64) nt = 64;
    if (n < 4096) nt = 1;
    size_t cnts[64];
    or_taghit* bufs[64];
    size_t caps[64];
    /* gap indices grouped by scaffold (counting sort, stable: the caller's order inside a scaffold) */
    size_t n_scaf = 0;
    for (size_t g = 0; g < n_gaps; ++g) if ((size_t)gaps[g].scaffold + 1 > n_scaf) n_scaf = (size_t)gaps[g].scaffold + 1;
    size_t* scaf_off = calloc(n_scaf + 2, sizeof(size_t));
    uint32_t* by_scaf = malloc((n_gaps + 1) * sizeof(uint32_t));
    for (size_t g = 0; g < n_gaps; ++g) scaf_off[gaps[g].scaffold + 1]++;
    for (size_t sc = 0; sc < n_scaf; ++sc) scaf_off[sc + 1] += scaf_off[sc];
    {
        size_t* cur = malloc((n_scaf + 1) * sizeof(size_t));
        memcpy(cur, scaf_off, (n_scaf + 1) * sizeof(size_t));
        for (size_t g = 0; g < n_gaps; ++g) by_scaf[cur[gaps[g].scaffold]++] = (uint32_t)g;
        free(cur);
    }
#pragma omp parallel for schedule(static, 1) num_threads(nt)
    for (int t = 0; t < nt; ++t) {
        size_t i0 = n * t / nt, i1 = n * (t + 1) / nt;
        caps[t] = 4 * (i1 - i0) + 64;
        bufs[t] = malloc(caps[t] * sizeof(or_taghit));
        cnts[t] = tag_range(recs, i0, i1, gaps, by_scaf, scaf_off, n_scaf, insert_size, sd, clip_dist, anchor_mapq, bufs[t], caps[t]);
        if (cnts[t] > caps[t]) { /* dense overlap of windows: redo with exact room */
            free(bufs[t]);
            caps[t] = cnts[t];
            bufs[t] = malloc(caps[t] * sizeof(or_taghit));
            cnts[t] = tag_range(recs, i0, i1, gaps, by_scaf, scaf_off, n_scaf, insert_size, sd, clip_dist, anchor_mapq, bufs[t], caps[t]);
        }
    }
    size_t cnt = 0;
    for (int t = 0; t < nt; ++t) {
        for (size_t i = 0; i < cnts[t]; ++i) { if (cnt < cap) out[cnt] = bufs[t][i]; ++cnt; }
        free(bufs[t]);
    }
    free(scaf_off); free(by_scaf);
    return cnt;
}

/* ------------------------------------------------------------------ f-3: the contig merger's all-pairs k-mer prefilter
 * QuickCheckerContigsMatch (ContigsCompactor.cpp:1982-2095) as CompactVer3 applies it (:836-853, threadQuickCheck :1073-1098):
 * node list = [c0, revcomp(c0), c1, revcomp(c1), ...] (:782-800); pair (i, j), i <= j (the pair (i, i) included), is FEASIBLE iff
 * some k-mer of the first 30 or of the last 30 bases of node j (IsMatchFeasibleV2 :2020-2039, lenContigLen = 30) occurs
 * anywhere in node i (Init :2041-2056); k-mers are KmerUtils' 2-bit strings, so any symbol other than C/G/T counts as A
 * (KmerUtils.cpp:25-41) and nothing is canonical.  Contigs must have >= 30 bases (the reference reads out of bounds below).
 * seqs: contigs back to back, off[n+1]; out: pairs (i, j) in (i, j) order; returns the number of pairs (may exceed cap). */
static uint64_t qc_kmer(const char* s, int k) {
    uint64_t v = 0;
    for (int i = 0; i < k; ++i) v = (v << 2) | code_of(s[i]);
    return v;
}
static int qc_cmp(const void* a, const void* b) {
    const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : x > y;
}
size_t or_quick_check(const char* seqs, const uint64_t* off, size_t n, int k, uint32_t* out_i, uint32_t* out_j, size_t cap) {
    const size_t nn = 2 * n;
    char** node = malloc((nn + 1) * sizeof(char*));
    size_t* len = malloc((nn + 1) * sizeof(size_t));
    for (size_t c = 0; c < n; ++c) {
        const size_t l = (size_t)(off[c + 1] - off[c]);
        len[2 * c] = len[2 * c + 1] = l;
        node[2 * c] = malloc(l + 1);
        node[2 * c + 1] = malloc(l + 1);
        for (size_t i = 0; i < l; ++i) {
            char ch = seqs[off[c] + i];
            if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);          /* the FASTA reader upper-cases */
            node[2 * c][i] = ch;
        }
        for (size_t i = 0; i < l; ++i) {                               /* FastaSequence::RevsereComplement: A<->T, C<->G, rest kept */
            const char ch = node[2 * c][l - 1 - i];
            node[2 * c + 1][i] = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : ch;
        }
    }
    size_t cnt = 0;
    for (size_t i = 0; i < nn; ++i) {
        const size_t li = len[i];
        if (li < (size_t)k) continue;
        const size_t m = li - k + 1;
        uint64_t* set = malloc((m + 1) * 8);                           /* mapKmerFreqInRepeat of node i, as a sorted array */
        for (size_t p = 0; p < m; ++p) set[p] = qc_kmer(node[i] + p, k);
        qsort(set, m, 8, qc_cmp);
        for (size_t j = i; j < nn; ++j) {
            int hit = 0;
            for (int side = 0; side < 2 && !hit; ++side) {
                const char* e = side ? node[j] + len[j] - 30 : node[j];
                for (int p = 0; p + k <= 30 && !hit; ++p) {
                    const uint64_t x = qc_kmer(e + p, k);
                    size_t a = 0, b = m;
                    while (a < b) { size_t mid = (a + b) / 2; if (set[mid] < x) a = mid + 1; else b = mid; }
                    hit = a < m && set[a] == x;
                }
            }
            if (hit) { if (cnt < cap) { out_i[cnt] = (uint32_t)i; out_j[cnt] = (uint32_t)j; } ++cnt; }
        }
        free(set);
    }
    for (size_t i = 0; i < nn; ++i) free(node[i]);
    free(node); free(len);
    return cnt;
}

/* ------------------------------------------------------------------ f-3, second stage: the overlap evaluation of one node pair
   ContigsCompactor::Evaluate + IsScoreSignificant + ContigsCompactorAction::SetMergedStringConcat / IsContainment
   (ContigsCompactor-v0.2.0/ContigsMerger/ContigsCompactor.cpp:1572-1976, :108-159).  Overlap alignment: first row and column 0,
   match +1, mismatch (int)mismatch (the reference assigns the double to an int, :1640), indel as given; predecessor order
   diagonal, then (i-1, j), then (i, j-1), each replacing only on a strictly larger score (:1651-1666).  The end cell is the
   maximum over the last column / last row shifted in by c = 0 .. max_clip (:1679-1708; strict >, so the first cell in that scan
   order wins; the reference keeps the maximum in an int, :1674).  The trace back ends on row 0 or column 0 (:1763-1808); whether
   it is row 0 / column 0 is all the caller uses of it (the contained flag, :1830-1833), so that is carried forward with the scores
   instead of a trace-back table. */
void or_overlap_evaluate(const char* s1, int n1, const char* s2, int n2, const or_ovl_params* pr, or_ovl_result* out) {
    const int mis = (int)pr->mismatch;
    const int W = n2 + 1;
    const int clip = (int)pr->max_clip;                      /* `c <= maxOverlapClipLen` with a double bound */
    double* prev = malloc((size_t)W * sizeof(double));
    double* cur = malloc((size_t)W * sizeof(double));
    unsigned char* fprev = malloc((size_t)W);               /* bit 0: the path starts on row 0, bit 1: on column 0 */
    unsigned char* fcur = malloc((size_t)W);
    /* scores and start flags of the last clip + 1 columns (every row) and of the last clip + 1 rows (every column) */
    const int CW = clip + 1;
    double* colS = malloc((size_t)(n1 + 1) * CW * sizeof(double));
    unsigned char* colF = malloc((size_t)(n1 + 1) * CW);
    double* rowS = malloc((size_t)CW * W * sizeof(double));
    unsigned char* rowF = malloc((size_t)CW * W);
    for (int i = 0; i <= n1; ++i) {
        if (i == 0) {
            for (int j = 0; j <= n2; ++j) { cur[j] = 0.0; fcur[j] = (unsigned char)(1 | (j == 0 ? 2 : 0)); }
        } else {
            cur[0] = 0.0; fcur[0] = 2;
            for (int j = 1; j <= n2; ++j) {
                double sc = prev[j - 1] + (s1[i - 1] == s2[j - 1] ? 1 : mis);
                unsigned char f = fprev[j - 1];
                if (sc < prev[j] + pr->indel) { sc = prev[j] + pr->indel; f = fprev[j]; }
                if (sc < cur[j - 1] + pr->indel) { sc = cur[j - 1] + pr->indel; f = fcur[j - 1]; }
                cur[j] = sc; fcur[j] = f;
            }
        }
        for (int c = 0; c < CW; ++c) {
            const int col = n2 - c;
            if (col >= 0) { colS[(size_t)i * CW + c] = cur[col]; colF[(size_t)i * CW + c] = fcur[col]; }
        }
        if (n1 - i < CW) {
            const int c = n1 - i;
            for (int j = 0; j <= n2; ++j) { rowS[(size_t)c * W + j] = cur[j]; rowF[(size_t)c * W + j] = fcur[j]; }
        }
        { double* t = prev; prev = cur; cur = t; }
        { unsigned char* t = fprev; fprev = fcur; fcur = t; }
    }
    int score_max = -1000000000, row_end = -1, col_end = -1, nclip = -1;
    unsigned char fend = 0;
    for (int c = 0; c <= clip; ++c) {
        for (int i = 0; i <= n1; ++i) {
            const int icol = n2 - c;
            if (icol < 0) break;
            if (colS[(size_t)i * CW + c] > score_max) { score_max = (int)colS[(size_t)i * CW + c]; col_end = icol; row_end = i; nclip = c; fend = colF[(size_t)i * CW + c]; }
        }
        for (int j = 0; j <= n2; ++j) {
            const int irow = n1 - c;
            if (irow < 0) break;
            if (rowS[(size_t)c * W + j] > score_max) { score_max = (int)rowS[(size_t)c * W + j]; col_end = j; row_end = irow; nclip = c; fend = rowF[(size_t)c * W + j]; }
        }
    }
    free(prev); free(cur); free(fprev); free(fcur); free(colS); free(colF); free(rowS); free(rowF);
    /* IsScoreSignificant, :1875-1976 */
    int res;
    {
        int ov0 = n1 < n2 ? n1 : n2, ov1 = ov0, ov2 = ov0;
        if (row_end + nclip == n1) ov1 = col_end;
        if (col_end + nclip == n2) ov2 = row_end;
        int ov = ov1 < ov2 ? ov1 : ov2;
        if (ov0 < ov) ov = ov0;
        res = 2;
        if (ov < n1 * pr->frac_min_overlap && ov < n2 * pr->frac_min_overlap) res = 0;
        else if (row_end + nclip == n1 && col_end + 5 - 1 >= n2) res = 0;
        else if (col_end + nclip == n2 && row_end + 5 - 1 >= n1) res = 0;
        else if (score_max < ov * (1 - pr->frac_loss)) res = 0;
        else if (ov < pr->min_overlap_scaffold) res = 0;
        else if (ov < pr->min_overlap) res = 1;
        if (pr->relax != 0.0) res = 2;                       /* fRelax (:1712-1725): the significance test is skipped */
    }
    memset(out, 0, sizeof *out);
    out->res = res; out->row_end = row_end; out->col_end = col_end; out->nclip = nclip; out->score = score_max;
    if (res == 0) return;                                    /* the reference returns before it fills the action */
    const int contained = (row_end + nclip == n1 && (fend & 1)) || (col_end + nclip == n2 && (fend & 2));
    int merged;                                              /* SetMergedStringConcat, :108-153 */
    if (contained && row_end + nclip == n1 && n1 < n2) merged = n2;
    else if (contained && col_end + nclip == n2 && n2 < n1) merged = n1;
    else if (row_end + nclip == n1) merged = (n1 - nclip) + (n2 - col_end);
    else merged = (n2 - nclip) + (n1 - row_end);
    out->contained = contained;
    out->merged_len = merged;
    out->overlap = n1 + n2 - nclip - merged;                 /* GetOverlapSize */
    out->containment = contained && ((row_end + nclip == n1 && n1 < col_end) || (col_end + nclip == n2 && n2 < row_end));   /* IsContainment */
    out->first_goes_first = (row_end + nclip) == n1;         /* threadMergeContigV2, :656-670: MODE_1_2, else MODE_2_1 */
}

/* ------------------------------------------------------------------ a-3: second hop */
size_t or_tag_low_mapq(const or_alnrec* recs, size_t n, const or_dpos* table, size_t n_rows, or_taghit* out, size_t cap) {
    size_t cnt = 0;
    for (size_t i = 0; i < n; ++i) {
        const or_alnrec* r = &recs[i];
        if (r->mapq > 0) continue; /* collect_discordant_low_mapq_reads.py:52 */
        /* focal_region[p] = last q in file order with q-199 <= p <= q+299 (:21-25) */
        long best = -1;
        for (size_t t = 0; t < n_rows; ++t) {
            if (table[t].mate_scaffold != r->ref) continue;
            long q = table[t].mate_pos, p = r->pos;
            if (q - 199 <= p && p <= q + 299) best = q; /* later rows overwrite */
        }
        if (best < 0) continue;
        for (size_t t = 0; t < n_rows; ++t) /* m_pos_gaps[q]: every row of q, duplicates kept (:15-19) */
            if (table[t].mate_scaffold == r->ref && (long)table[t].mate_pos == best)
                put(out, cap, &cnt, (uint32_t)i, (uint32_t)t, 3, 0);
    }
    return cnt;
}

/* ------------------------------------------------------------------ north-star screen */
typedef struct { uint64_t hi, lo; uint32_t gap; } kent;

static int kent_cmp(const void* a, const void* b) {
    const kent *x = a, *y = b;
    if (x->hi != y->hi) return x->hi < y->hi ? -1 : 1;
    if (x->lo != y->lo) return x->lo < y->lo ? -1 : 1;
    if (x->gap != y->gap) return x->gap < y->gap ? -1 : 1;
    return 0;
}

/* canonical left-aligned 128-bit k-mer of seq[0..k); returns 0 if a non-ACGT (upper-case) byte occurs */
static int canon_kmer(const char* s, int k, uint64_t* hi, uint64_t* lo) {
    uint64_t fh = 0, fl = 0, rh = 0, rl = 0;
    for (int i = 0; i < k; ++i) {
        char c = s[i];
        if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T')) return 0;
        uint64_t f = code_of(c), r = 3 - code_of(s[k - 1 - i]);
        if (i < 32) { fh |= f << (62 - 2 * i); rh |= r << (62 - 2 * i); }
        else { fl |= f << (62 - 2 * (i - 32)); rl |= r << (62 - 2 * (i - 32)); }
    }
    if (rh < fh || (rh == fh && rl < fl)) { *hi = rh; *lo = rl; } else { *hi = fh; *lo = fl; }
    return 1;
}

static int hit_cmp(const void* a, const void* b) {
    const or_hit *x = a, *y = b;
    if (x->gap != y->gap) return x->gap < y->gap ? -1 : 1;
    if (x->read != y->read) return x->read < y->read ? -1 : 1;
    return 0;
}

/* seconds the LAST or_screen_reads call spent building its flank k-mer table (before the first read is looked at): lets a caller time
 * the per-read pass of that very call instead of subtracting a separately measured build (bench.py's cpu_baseline) */
static double g_screen_build_s = 0.0;
double or_screen_last_build_s(void) { return g_screen_build_s; }
static double or_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

size_t or_screen_reads(const char* reads, size_t n_reads, int L, const char* flank, const uint64_t* foff, size_t n_gaps,
                       int k, int min_hits, uint32_t max_occ, or_hit* out, size_t cap, int threads) {
    const double t_build0 = or_now();
    /* flank k-mer set per gap: canonical k-mers of left+right flank, k-mers with non-ACGT skipped */
    size_t total = 0;
    for (size_t g = 0; g < 2 * n_gaps; ++g) { uint64_t len = foff[g + 1] - foff[g]; if (len >= (uint64_t)k) total += len - k + 1; }
    kent* tab = malloc((total + 1) * sizeof(kent));
    size_t nt = 0;
    for (size_t g = 0; g < n_gaps; ++g)
        for (int side = 0; side < 2; ++side) {
            const char* s = flank + foff[2 * g + side];
            long len = (long)(foff[2 * g + side + 1] - foff[2 * g + side]);
            for (long p = 0; p + k <= len; ++p)
                if (canon_kmer(s + p, k, &tab[nt].hi, &tab[nt].lo)) tab[nt++].gap = (uint32_t)g;
        }
    qsort(tab, nt, sizeof(kent), kent_cmp);
    size_t w = 0; /* unique (k-mer, gap), then the max_occ repeat rule */
    for (size_t i = 0; i < nt; ++i) if (i == 0 || kent_cmp(&tab[i], &tab[i - 1]) != 0) tab[w++] = tab[i];
    nt = w;
    if (max_occ) {
        w = 0;
        for (size_t i = 0; i < nt;) {
            size_t j = i;
            while (j < nt && tab[j].hi == tab[i].hi && tab[j].lo == tab[i].lo) ++j;
            if (j - i <= max_occ) for (size_t t = i; t < j; ++t) tab[w++] = tab[t];
            i = j;
        }
        nt = w;
    }
    if (min_hits < 1) min_hits = 1;
    size_t cnt = 0;
    /* distinct k-mers -> run [first, last) in tab, through an open-addressing index (keys right-aligned in 128 bits) */
    typedef unsigned __int128 u128;
    size_t nd = 0;
    for (size_t i = 0; i < nt; ++i) if (i == 0 || tab[i].hi != tab[i - 1].hi || tab[i].lo != tab[i - 1].lo) ++nd;
    size_t hcap = 1024;
    while (hcap < 2 * nd + 2) hcap <<= 1;
    u128* hkey = malloc(hcap * sizeof(u128));
    uint32_t* hfirst = malloc(hcap * sizeof(uint32_t));
    uint32_t* hlast = malloc(hcap * sizeof(uint32_t));
    for (size_t i = 0; i < hcap; ++i) hfirst[i] = 0xFFFFFFFFu;
    const int sh = 128 - 2 * k;
    for (size_t i = 0; i < nt;) {
        size_t j = i;
        while (j < nt && tab[j].hi == tab[i].hi && tab[j].lo == tab[i].lo) ++j;
        u128 key = (((u128)tab[i].hi << 64) | tab[i].lo) >> sh;
        size_t hs = (size_t)((uint64_t)(key ^ (key >> 61)) * 0x9E3779B97F4A7C15ull >> 20) & (hcap - 1);
        while (hfirst[hs] != 0xFFFFFFFFu) hs = (hs + 1) & (hcap - 1);
        hkey[hs] = key; hfirst[hs] = (uint32_t)i; hlast[hs] = (uint32_t)j;
        i = j;
    }
    const u128 kmask = k == 64 ? ~(u128)0 : (((u128)1 << (2 * k)) - 1);
    g_screen_build_s = or_now() - t_build0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel
    {
        uint32_t* glist = malloc(sizeof(uint32_t) * 65536);
#pragma omp for schedule(dynamic, 4096)
        for (long r = 0; r < (long)n_reads; ++r) {
            size_t ng = 0;
            const char* s = reads + (size_t)r * L;
            u128 f = 0, rc = 0;   /* forward k-mer and its reverse complement, rolled base by base */
            int run = 0;          /* ACGT bases since the last other symbol */
            for (int i = 0; i < L; ++i) {
                const char c = s[i];
                if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T')) { run = 0; f = 0; rc = 0; continue; }
                const u128 b = code_of(c);
                f = ((f << 2) | b) & kmask;
                rc = (rc >> 2) | ((u128)(3 - b) << (2 * (k - 1)));
                if (++run < k) continue;
                const u128 key = f < rc ? f : rc;
                size_t hs = (size_t)((uint64_t)(key ^ (key >> 61)) * 0x9E3779B97F4A7C15ull >> 20) & (hcap - 1);
                while (hfirst[hs] != 0xFFFFFFFFu) {
                    if (hkey[hs] == key) {
                        for (uint32_t a = hfirst[hs]; a < hlast[hs] && ng < 65536; ++a) glist[ng++] = tab[a].gap;
                        break;
                    }
                    hs = (hs + 1) & (hcap - 1);
                }
            }
            if (!ng) continue;
            /* count positions per gap */
            for (size_t i = 1; i < ng; ++i) { uint32_t v = glist[i]; size_t j = i; while (j && glist[j - 1] > v) { glist[j] = glist[j - 1]; --j; } glist[j] = v; }
            for (size_t i = 0; i < ng;) {
                size_t j = i;
                while (j < ng && glist[j] == glist[i]) ++j;
                if ((int)(j - i) >= min_hits) {
                    size_t o;
#pragma omp atomic capture
                    o = cnt++;
                    if (o < cap) { out[o].gap = glist[i]; out[o].read = (uint32_t)r; }
                }
                i = j;
            }
        }
        free(glist);
    }
    free(hkey); free(hfirst); free(hlast);
    free(tab);
    qsort(out, cnt < cap ? cnt : cap, sizeof(or_hit), hit_cmp);
    return cnt;
}

/* ------------------------------------------------------------------ synthetic workload (definition: include/gf_synth.h) */
#include "../include/gf_synth.h"

void or_synth_pairs(const void* c_, uint64_t first_pair, size_t n_pairs, uint8_t* packed, or_alnrec* recs) {
    const gf_synth_cfg* c = (const gf_synth_cfg*)c_;
    const size_t rb = (c->read_len + 3) / 4;
#pragma omp parallel for schedule(static)
    for (long lp = 0; lp < (long)n_pairs; ++lp) {
        gfs_pair p;
        gfs_make_pair(c, first_pair + lp, &p);
        for (int end = 0; end < 2; ++end) {
            gfs_errs er;
            gfs_make_errs(c, p.err[end], &er);
            const uint32_t mate_no = ((end == 0) != (p.flip != 0)) ? 0u : 1u;
            uint8_t* o = packed + (2 * (size_t)lp + mate_no) * rb;
            memset(o, 0, rb);
            for (uint32_t i = 0; i < c->read_len; ++i)
                o[i >> 2] |= (uint8_t)(gfs_read_base(c, &p, &er, end, i) << (6 - 2 * (i & 3)));
        }
        if (recs) {
            uint32_t r[2][8];
            gfs_make_records(c, (uint64_t)lp, &p, r);   /* read id = index in this batch */
            memcpy(&recs[2 * lp], r[0], 32);
            memcpy(&recs[2 * lp + 1], r[1], 32);
        }
    }
}

void or_synth_layout(const void* c_, or_gap* gaps, char* flank_ascii, uint64_t* flank_off) {
    const gf_synth_cfg* c = (const gf_synth_cfg*)c_;
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
}

/* ------------------------------------------------------------------ a-6: count + de Bruijn unitigs (PARITY UNPINNED)
 * Data flow of run_assembly (assemble_gaps.py:82-136): `kmc -k{k}` (canonical k-mers, k-mers with non-ACGT skipped,
 * default min count 2) | `kmc_dump` (ascending) | every surviving k-mer becomes one Velvet read | `velveth {kv}`,
 * `velvetg -min_contig_lgth 40`.  KMC and Velvet are not in the reference tree, so this build DEFINES stage 2
 * (DESIGN.md "Assembly semantics"):
 *   nodes  = canonical kv-mers of the surviving k-mers (kv odd), multiplicity = number of (k-mer, offset) occurrences;
 *   edges  = consecutive kv-mers inside a surviving k-mer;
 *   oriented node (x,d): sequence x (d=0) or revcomp(x) (d=1); it is a unitig START iff its in-degree != 1 or its
 *            unique predecessor has out-degree != 1; a walk extends while out-degree == 1 and the successor's in-degree == 1;
 *   each unitig is found from both ends; the walk whose first kv-mer is <= the first kv-mer of the opposite walk is
 *   emitted (so the contig is min(seq, revcomp(seq)) whenever the two first kv-mers differ);
 *   contigs shorter than min_contig bases are dropped; isolated cycles (no start) are not reported.  Velvet's default error
 *   removal (tip clipping + bubble popping) runs on the unitig graph before the emission: simplify_round below.
 *   Output order: length descending, then sequence ascending.
 */
typedef struct { uint64_t hi, lo; } k128;
static inline int k128_lt(k128 a, k128 b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
static inline int k128_eq(k128 a, k128 b) { return a.hi == b.hi && a.lo == b.lo; }
static int k128_cmp(const void* a, const void* b) {
    const k128 *x = a, *y = b;
    return k128_lt(*x, *y) ? -1 : k128_lt(*y, *x) ? 1 : 0;
}
static inline unsigned k128_base(k128 v, int i) { return i < 32 ? (unsigned)(v.hi >> (62 - 2 * i)) & 3u : (unsigned)(v.lo >> (62 - 2 * (i - 32))) & 3u; }
static inline void k128_set(k128* v, int i, unsigned b) {
    if (i < 32) v->hi |= (uint64_t)b << (62 - 2 * i); else v->lo |= (uint64_t)b << (62 - 2 * (i - 32));
}
static k128 k128_rc(k128 v, int len) {
    k128 r = {0, 0};
    for (int i = 0; i < len; ++i) k128_set(&r, len - 1 - i, 3u - k128_base(v, i));
    return r;
}
static k128 k128_sub(k128 v, int off, int len) { /* bases off .. off+len-1 */
    k128 r = {0, 0};
    for (int i = 0; i < len; ++i) k128_set(&r, i, k128_base(v, off + i));
    return r;
}
static k128 k128_canon(k128 v, int len, int* flipped) {
    k128 r = k128_rc(v, len);
    if (k128_lt(r, v)) { if (flipped) *flipped = 1; return r; }
    if (flipped) *flipped = 0;
    return v;
}

/* canonical k-mers with count >= min_count, ascending; returns the number (may exceed cap) */
size_t or_count_kmers(const char* reads, size_t n_reads, int L, int k, int min_count, uint64_t* hi, uint64_t* lo,
                      uint32_t* cnt, size_t cap) {
    size_t tot = n_reads * (size_t)(L - k + 1), n = 0;
    k128* a = malloc((tot + 1) * sizeof(k128));
    for (size_t r = 0; r < n_reads; ++r)
        for (int p = 0; p + k <= L; ++p)
            if (canon_kmer(reads + r * L + p, k, &a[n].hi, &a[n].lo)) ++n;
    qsort(a, n, sizeof(k128), k128_cmp);
    size_t w = 0;
    for (size_t i = 0; i < n;) {
        size_t j = i;
        while (j < n && k128_eq(a[j], a[i])) ++j;
        if ((long)(j - i) >= min_count) {
            if (w < cap) { hi[w] = a[i].hi; lo[w] = a[i].lo; cnt[w] = (uint32_t)((j - i) > 10000000 ? 10000000 : (j - i)); }
            ++w;
        }
        i = j;
    }
    free(a);
    return w;
}

/* weak: one of the surviving k-mers the node came from was seen at most min_count + 1 times (the k-mer's true count) */
typedef struct { k128 key; uint32_t mult; uint8_t out, in, dead, weak; } or_node;
static long node_find(const or_node* nd, size_t n, k128 key) {
    size_t a = 0, b = n;
    while (a < b) { size_t m = (a + b) / 2; if (k128_lt(nd[m].key, key)) a = m + 1; else b = m; }
    return (a < n && k128_eq(nd[a].key, key)) ? (long)a : -1;
}
static inline unsigned rev4(unsigned b) { return ((b & 1) << 3) | ((b & 2) << 1) | ((b & 4) >> 1) | ((b & 8) >> 3); }
static inline unsigned bits_out(const or_node* nd, long i, int d) { return d ? rev4(nd[i].in) : nd[i].out; } /* bit c = next base c */
static inline unsigned bits_in(const or_node* nd, long i, int d) { return d ? rev4(nd[i].out) : nd[i].in; }  /* bit c = previous base c */
static inline int popc4(unsigned b) { return (b & 1) + ((b >> 1) & 1) + ((b >> 2) & 1) + ((b >> 3) & 1); }
static inline int ctz4(unsigned b) { return (b & 1) ? 0 : (b & 2) ? 1 : (b & 4) ? 2 : 3; }

typedef struct { uint32_t n_nodes, length, cov_sum; char* seq; } or_ctg;
static int ctg_cmp(const void* a, const void* b) {
    const or_ctg *x = a, *y = b;
    if (x->length != y->length) return x->length > y->length ? -1 : 1;
    return strcmp(x->seq, y->seq);
}

/* ---- the kv-mer graph of one pool.  Oriented node o = 2 * index + d (d = 1: the reverse complement of the stored key). */
typedef struct {
    or_node* nd; size_t nn; int kv;
    /* unitig decomposition of the live graph (graph_unitigs): per ORIENTED node */
    uint32_t* head;   /* head of the oriented unitig the oriented node lies on (NONE on an isolated cycle) */
    uint32_t* tail;   /* [head]: last oriented node */
    uint32_t* len;    /* [head]: nodes */
    uint32_t* cov;    /* [head]: sum of the nodes' multiplicities */
    uint32_t* wk;     /* [head]: number of weak nodes */
    uint32_t* next;   /* unitig-internal successor, or NONE */
    k128* ukey;       /* [head]: min(first kv-mer of the unitig, first kv-mer of its reverse) — orientation-free order */
    uint8_t* kill;    /* [node index]: marked for removal in this round */
} or_graph;
#define OR_NONE 0xFFFFFFFFu

static k128 g_oseq(const or_graph* G, uint32_t o) { return (o & 1) ? k128_rc(G->nd[o >> 1].key, G->kv) : G->nd[o >> 1].key; }
static unsigned g_out(const or_graph* G, uint32_t o) { return bits_out(G->nd, (long)(o >> 1), (int)(o & 1)); }
static unsigned g_in(const or_graph* G, uint32_t o) { return bits_in(G->nd, (long)(o >> 1), (int)(o & 1)); }
/* oriented successor of o along base c / oriented predecessor that has base c in front */
static uint32_t g_succ(const or_graph* G, uint32_t o, unsigned c) {
    k128 cur = g_oseq(G, o), y = {0, 0};
    for (int q = 0; q + 1 < G->kv; ++q) k128_set(&y, q, k128_base(cur, q + 1));
    k128_set(&y, G->kv - 1, c);
    int dy;
    k128 Y = k128_canon(y, G->kv, &dy);
    long iy = node_find(G->nd, G->nn, Y);
    return iy < 0 ? OR_NONE : (uint32_t)(2 * iy + dy);
}
static uint32_t g_pred(const or_graph* G, uint32_t o, unsigned c) {
    k128 cur = g_oseq(G, o), p = {0, 0};
    k128_set(&p, 0, c);
    for (int q = 1; q < G->kv; ++q) k128_set(&p, q, k128_base(cur, q - 1));
    int dp;
    k128 P = k128_canon(p, G->kv, &dp);
    long ip = node_find(G->nd, G->nn, P);
    return ip < 0 ? OR_NONE : (uint32_t)(2 * ip + dp);
}

/* unitigs of the live graph: an oriented node STARTS a unitig iff its in-degree != 1 or its unique predecessor's out-degree
 * != 1; a walk extends while out-degree == 1 and the successor's in-degree == 1 */
static void graph_unitigs(or_graph* G) {
    const size_t no = 2 * G->nn;
    for (size_t o = 0; o < no; ++o) { G->head[o] = OR_NONE; G->next[o] = OR_NONE; }
    for (size_t o = 0; o < no; ++o) {
        if (G->nd[o >> 1].dead) continue;
        unsigned ib = g_in(G, (uint32_t)o);
        int start = popc4(ib) != 1;
        if (!start) start = popc4(g_out(G, g_pred(G, (uint32_t)o, (unsigned)ctz4(ib)))) != 1;
        if (!start) continue;
        uint32_t cur = (uint32_t)o, n = 1, cov = G->nd[o >> 1].mult, wk = G->nd[o >> 1].weak;
        G->head[o] = (uint32_t)o;
        for (;;) {
            unsigned ob = g_out(G, cur);
            if (popc4(ob) != 1) break;
            uint32_t y = g_succ(G, cur, (unsigned)ctz4(ob));
            if (popc4(g_in(G, y)) != 1) break;
            G->next[cur] = y;
            G->head[y] = (uint32_t)o;
            cur = y; ++n; cov += G->nd[y >> 1].mult; wk += G->nd[y >> 1].weak;
        }
        G->tail[o] = cur; G->len[o] = n; G->cov[o] = cov; G->wk[o] = wk;
        k128 a = g_oseq(G, (uint32_t)o), b = k128_rc(g_oseq(G, cur), G->kv);
        G->ukey[o] = k128_lt(b, a) ? b : a;
    }
}

/* Velvet's default error removal (velvetg without -cov_cutoff: tip clipping + Tour Bus bubble popping), DEFINED
 * here on the unitig graph — Velvet itself is absent, and its coverage-based choices are coin flips on this input where every
 * surviving k-mer is one read (assemble_gaps.py:104-118), so ties are broken — north_star: "for fixed tie-breaking" — in one of two
 * modes.  tiebreak = 1 ("counts", the default): first by the evidence the k-mer counts still hold (a sequencing error that made it
 * past min_count was seen min_count times, rarely once more, the true sequence many more: the side with FEWER WEAK nodes stays), then
 * by an orientation-free sequence order.  tiebreak = 0 ("none", the reference-shaped mode): the sequence order alone — cvtFaToFq
 * drops the counts before Velvet reads a k-mer (assemble_gaps.py:56-79), so nothing here uses what Velvet could not have known.
 *   X beats Y  :=  (cov X, then fewer weak nodes, then the SMALLER ukey) wins, preceded by (nodes) where lengths can differ.
 * One round decides on ONE snapshot of the graph, for every oriented unitig X (head h, tail t, n nodes):
 *  TIP     out-degree(t) == 0, in-degree(h) == 1 with predecessor p of out-degree >= 2, n <= kv (i.e. n + kv - 1 < 2 kv bases:
 *          Velvet's tip length), and some OTHER branch Y leaving p beats X: Y is not tip-shaped itself (tip-shaped = dead end,
 *          n <= kv, in-degree(head) == 1), or (n, cov, smaller ukey) of Y > that of X.  So of several tips on one junction the
 *          best one stays when nothing longer leaves the junction.
 *  BUBBLE  in-degree(h) == 1 with predecessor p of out-degree >= 2, out-degree(t) == 1 with successor s of in-degree >= 2,
 *          n <= 2 kv, and there is an alternative path p -> A1 .. Am -> s of whole unitigs (m <= 4, none of them X or its reverse)
 *          with exactly n nodes in total (substitution bubbles), and (m >= 2, or A1 beats X by (cov, smaller ukey)).
 * Removed unitigs lose all their nodes and the arcs p -> h and t -> s.  Rounds repeat (at most `rounds`) while something was
 * removed; every round re-derives the unitigs. */
#define OR_ALT_DEPTH 4
static int tip_shaped(const or_graph* G, uint32_t h) {
    return popc4(g_out(G, G->tail[h])) == 0 && G->len[h] <= (uint32_t)G->kv && popc4(g_in(G, h)) == 1;
}
/* does Y (head y) beat X (head x)?  with_len: compare node counts first */
static int beats(const or_graph* G, uint32_t y, uint32_t x, int with_len) {
    if (with_len && G->len[y] != G->len[x]) return G->len[y] > G->len[x];
    if (G->cov[y] != G->cov[x]) return G->cov[y] > G->cov[x];
    if (G->wk[y] != G->wk[x]) return G->wk[y] < G->wk[x];      /* (tiebreak 0: no node is ever marked weak, wk is 0 everywhere) */
    return k128_lt(G->ukey[y], G->ukey[x]);
}
/* alternative paths from tail q (q = p at depth 0) to s with exactly `remain` nodes; returns 1 when X must go */
static int alt_search(const or_graph* G, uint32_t q, uint32_t s, uint32_t remain, int depth, uint32_t x) {
    const uint32_t xr = G->tail[x] ^ 1u;   /* head of X reversed */
    unsigned ob = g_out(G, q);
    for (unsigned c = 0; c < 4; ++c) {
        if (!(ob & (1u << c))) continue;
        uint32_t y = g_succ(G, q, c);
        if (y == OR_NONE || G->head[y] != y) continue;   /* (always a head; guards a corrupt graph) */
        if (y == x || y == xr) continue;
        if (G->len[y] > remain) continue;
        if (G->len[y] == remain) {
            unsigned tb = g_out(G, G->tail[y]);
            int reaches = 0;
            for (unsigned c2 = 0; c2 < 4; ++c2) if ((tb & (1u << c2)) && g_succ(G, G->tail[y], c2) == s) reaches = 1;
            if (!reaches) continue;
            if (depth >= 1 || beats(G, y, x, 0)) return 1;
        } else if (depth + 1 < OR_ALT_DEPTH) {
            if (alt_search(G, G->tail[y], s, remain - G->len[y], depth + 1, x)) return 1;
        }
    }
    return 0;
}
static size_t simplify_round(or_graph* G) {
    const size_t no = 2 * G->nn;
    memset(G->kill, 0, G->nn);
    size_t removed = 0;
    for (size_t o = 0; o < no; ++o) {
        if (G->nd[o >> 1].dead || G->head[o] != o) continue;
        const uint32_t h = (uint32_t)o, t = G->tail[h], n = G->len[h];
        unsigned ib = g_in(G, h);
        if (popc4(ib) != 1) continue;
        const uint32_t p = g_pred(G, h, (unsigned)ctz4(ib));
        const unsigned pb = g_out(G, p);
        if (popc4(pb) < 2) continue;
        int go = 0;
        const unsigned tb = g_out(G, t);
        if (popc4(tb) == 0 && n <= (uint32_t)G->kv) {                       /* TIP */
            for (unsigned c = 0; c < 4 && !go; ++c) {
                if (!(pb & (1u << c))) continue;
                uint32_t y = g_succ(G, p, c);
                if (y == OR_NONE || y == h || G->head[y] != y || y == (t ^ 1u)) continue;
                if (!tip_shaped(G, y) || beats(G, y, h, 1)) go = 1;
            }
        } else if (popc4(tb) == 1 && n <= 2u * (uint32_t)G->kv) {           /* BUBBLE */
            const uint32_t s = g_succ(G, t, (unsigned)ctz4(tb));
            if (s != OR_NONE && popc4(g_in(G, s)) >= 2) go = alt_search(G, p, s, n, 0, h);
        }
        if (!go) continue;
        for (uint32_t cur = h;; cur = G->next[cur]) { G->kill[cur >> 1] = 1; if (cur == t) break; }
        ++removed;
    }
    if (!removed) return 0;
    for (size_t i = 0; i < G->nn; ++i) if (G->kill[i]) G->nd[i].dead = 1;
    /* arcs into removed nodes disappear (by the rules these are exactly p -> h and t -> s of every removed unitig) */
    for (size_t o = 0; o < no; ++o) {
        if (G->nd[o >> 1].dead) continue;
        unsigned ob = g_out(G, (uint32_t)o);
        for (unsigned c = 0; c < 4; ++c) {
            if (!(ob & (1u << c))) continue;
            uint32_t y = g_succ(G, (uint32_t)o, c);
            if (y != OR_NONE && !G->nd[y >> 1].dead) continue;
            if (!(o & 1)) G->nd[o >> 1].out &= (uint8_t)~(1u << c); else G->nd[o >> 1].in &= (uint8_t)~(1u << (3 - c));
        }
    }
    return removed;
}

/* contigs of one pool.  seq_out receives the sequences back to back (no terminators); per contig n_nodes[], length[],
 * cov_sum[].  simplify = rounds of tip clipping + bubble popping (0: raw unitigs; 2: the default of the product).
 * Returns the number of contigs (may exceed cap; *seq_need = bytes needed). */
size_t or_assemble_pool3(const char* reads, size_t n_reads, int L, int k, int kv, int min_count, int min_contig, int simplify, int tiebreak,
                         uint32_t* n_nodes, uint32_t* length, uint32_t* cov_sum, size_t cap, char* seq_out, size_t seq_cap,
                         size_t* seq_need) {
    *seq_need = 0;
    if (kv >= k || !(kv & 1) || k > 64 || L < k) return 0;
    size_t tot = n_reads * (size_t)(L - k + 1);
    uint64_t* hi = malloc((tot + 1) * 8); uint64_t* lo = malloc((tot + 1) * 8); uint32_t* cn = malloc((tot + 1) * 4);
    size_t ns = or_count_kmers(reads, n_reads, L, k, min_count, hi, lo, cn, tot + 1);
    const int per = k - kv + 1;
    k128* keys = malloc((ns * per + 1) * sizeof(k128));
    size_t nk = 0;
    for (size_t s = 0; s < ns; ++s) {
        k128 t = {hi[s], lo[s]};
        for (int o = 0; o < per; ++o) keys[nk++] = k128_canon(k128_sub(t, o, kv), kv, NULL);
    }
    qsort(keys, nk, sizeof(k128), k128_cmp);
    or_node* nd = malloc((nk + 1) * sizeof(or_node));
    size_t nn = 0;
    for (size_t i = 0; i < nk;) {
        size_t j = i;
        while (j < nk && k128_eq(keys[j], keys[i])) ++j;
        nd[nn].key = keys[i]; nd[nn].mult = (uint32_t)(j - i); nd[nn].out = nd[nn].in = nd[nn].dead = nd[nn].weak = 0; ++nn;
        i = j;
    }
    for (size_t s = 0; tiebreak && s < ns; ++s) { /* weak k-mers mark their nodes */
        if (cn[s] > (uint32_t)(min_count < 1 ? 1 : min_count) + 1u) continue;
        k128 t = {hi[s], lo[s]};
        for (int o = 0; o < per; ++o) nd[node_find(nd, nn, k128_canon(k128_sub(t, o, kv), kv, NULL))].weak = 1;
    }
    for (size_t s = 0; s < ns; ++s) { /* edges */
        k128 t = {hi[s], lo[s]};
        for (int o = 0; o + 1 < per; ++o) {
            int da, db;
            k128 A = k128_canon(k128_sub(t, o, kv), kv, &da), B = k128_canon(k128_sub(t, o + 1, kv), kv, &db);
            unsigned c_out = k128_base(t, o + kv), c_in = k128_base(t, o);
            long ia = node_find(nd, nn, A), ib = node_find(nd, nn, B);
            if (!da) nd[ia].out |= 1u << c_out; else nd[ia].in |= 1u << (3 - c_out);
            if (!db) nd[ib].in |= 1u << c_in; else nd[ib].out |= 1u << (3 - c_in);
        }
    }
    or_graph G;
    G.nd = nd; G.nn = nn; G.kv = kv;
    G.head = malloc((2 * nn + 1) * 4); G.tail = malloc((2 * nn + 1) * 4); G.len = malloc((2 * nn + 1) * 4);
    G.cov = malloc((2 * nn + 1) * 4); G.wk = malloc((2 * nn + 1) * 4); G.next = malloc((2 * nn + 1) * 4); G.ukey = malloc((2 * nn + 1) * sizeof(k128));
    G.kill = malloc(nn + 1);
    graph_unitigs(&G);
    for (int round = 0; round < simplify; ++round) {
        if (!simplify_round(&G)) break;
        graph_unitigs(&G);
    }
    or_ctg* ctg = malloc((2 * nn + 1) * sizeof(or_ctg));
    size_t nc = 0;
    for (size_t o = 0; o < 2 * nn; ++o) {
        if (nd[o >> 1].dead || G.head[o] != o) continue;
        const uint32_t t = G.tail[o], nodes = G.len[o];
        const size_t len = (size_t)nodes + kv - 1;
        /* emit rule: first kv-mer of this walk <= first kv-mer of the opposite walk (= revcomp of the last kv-mer) */
        k128 first = g_oseq(&G, (uint32_t)o), opp = k128_rc(g_oseq(&G, t), kv);
        if (k128_lt(opp, first) || (int)len < min_contig) continue;
        char* seq = malloc(len + 1);
        size_t w = 0;
        for (int q = 0; q < kv; ++q) seq[w++] = "ACGT"[k128_base(first, q)];
        for (uint32_t cur = G.next[o]; cur != OR_NONE; cur = G.next[cur]) seq[w++] = "ACGT"[k128_base(g_oseq(&G, cur), kv - 1)];
        seq[w] = 0;
        ctg[nc].n_nodes = nodes; ctg[nc].length = (uint32_t)len; ctg[nc].cov_sum = G.cov[o]; ctg[nc].seq = seq; ++nc;
    }
    qsort(ctg, nc, sizeof(or_ctg), ctg_cmp);
    size_t off = 0;
    for (size_t c = 0; c < nc; ++c) {
        if (c < cap) { n_nodes[c] = ctg[c].n_nodes; length[c] = ctg[c].length; cov_sum[c] = ctg[c].cov_sum; }
        if (off + ctg[c].length <= seq_cap && c < cap) memcpy(seq_out + off, ctg[c].seq, ctg[c].length);
        off += ctg[c].length;
        free(ctg[c].seq);
    }
    *seq_need = off;
    free(ctg); free(G.head); free(G.tail); free(G.len); free(G.cov); free(G.wk); free(G.next); free(G.ukey); free(G.kill);
    free(nd); free(keys); free(hi); free(lo); free(cn);
    return nc;
}

size_t or_assemble_pool2(const char* reads, size_t n_reads, int L, int k, int kv, int min_count, int min_contig, int simplify,
                         uint32_t* n_nodes, uint32_t* length, uint32_t* cov_sum, size_t cap, char* seq_out, size_t seq_cap,
                         size_t* seq_need) {   /* the default tie-break ("counts") */
    return or_assemble_pool3(reads, n_reads, L, k, kv, min_count, min_contig, simplify, 1, n_nodes, length, cov_sum, cap, seq_out, seq_cap, seq_need);
}

size_t or_assemble_pool(const char* reads, size_t n_reads, int L, int k, int kv, int min_count, int min_contig,
                        uint32_t* n_nodes, uint32_t* length, uint32_t* cov_sum, size_t cap, char* seq_out, size_t seq_cap,
                        size_t* seq_need) {   /* raw unitigs (no error removal) */
    return or_assemble_pool2(reads, n_reads, L, k, kv, min_count, min_contig, 0, n_nodes, length, cov_sum, cap, seq_out, seq_cap, seq_need);
}
