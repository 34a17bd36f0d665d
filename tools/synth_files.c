/* synth_files.c — the seeded synthetic workload of include/gf_synth.h written out as the FILES the reference's CLI
 * takes: a draft FASTA with the planted gaps as N-runs, a coordinate-sorted BAM (BGZF, zlib level 1) and the FASTQ pair of one
 * library.  Bench / test utility (bench.py's `e2e_files` extra, tests/test_gpu_e2e_files.py): the same reads and alignment
 * records gf_synth_pairs_dev puts into HBM, so a CLI run on these files must recruit what the device-resident bench recruits.
 *
 *   synth_files OUTDIR seed scaffold_len n_scaffolds gaps_per_scaffold gap_len read_len insert_mean insert_sd library n_pairs
 *
 * writes OUTDIR/draft.fa (library 0 only), OUTDIR/lib{library}.bam, OUTDIR/lib{library}_1.fq, OUTDIR/lib{library}_2.fq.
 * Read names r{pair} (FASTQ headers @r{pair}/1, @r{pair}/2); scaffold names scf{index}; qualities constant 'I'.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "../include/gf_synth.h"

static void die(const char* m) { fprintf(stderr, "synth_files: %s\n", m); exit(1); }

typedef struct { uint32_t ref, pos; uint64_t idx; uint32_t meta, pad; } sortkey;   /* meta = flag | mapq << 16 | clip << 24 */
static int cmp_key(const void* a, const void* b) {
    const sortkey *x = a, *y = b;
    if (x->ref != y->ref) return x->ref < y->ref ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

/* gfs_align with the aligned length: soft clip = read_len - aligned */
static uint32_t aligned_len(const gf_synth_cfg* c, uint64_t lo) {
    const uint64_t L = c->read_len, hi = lo + L;
    uint64_t jj = lo * (c->gaps_per_scaffold + 1) / c->scaffold_len;
    for (int d = -1; d <= 1; ++d) {
        int64_t j = (int64_t)jj - 1 + d;
        if (j < 0 || j >= (int64_t)c->gaps_per_scaffold) continue;
        const uint64_t gs = gfs_gap_start(c, (uint32_t)j), ge = gs + c->gap_len;
        if (ge <= lo || gs >= hi) continue;
        const uint64_t left = gs > lo ? gs - lo : 0, right = hi > ge ? hi - ge : 0;
        return (uint32_t)(left >= right ? left : right);
    }
    return (uint32_t)L;
}

static int reg2bin(int64_t beg, int64_t end) {   /* SAMv1 §5.3 */
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

static void put32(uint8_t* p, uint32_t v) { p[0] = v; p[1] = v >> 8; p[2] = v >> 16; p[3] = v >> 24; }
static void put16(uint8_t* p, uint32_t v) { p[0] = v; p[1] = v >> 8; }

int main(int argc, char** argv) {
    if (argc != 12) die("usage: synth_files OUTDIR seed scaffold_len n_scaffolds gaps_per_scaffold gap_len read_len insert_mean insert_sd library n_pairs");
    const char* out = argv[1];
    gf_synth_cfg c;
    memset(&c, 0, sizeof c);
    c.seed = strtoull(argv[2], 0, 10);
    c.scaffold_len = strtoull(argv[3], 0, 10);
    c.n_scaffolds = (uint32_t)atol(argv[4]);
    c.gaps_per_scaffold = (uint32_t)atol(argv[5]);
    c.gap_len = (uint32_t)atol(argv[6]);
    c.read_len = (uint32_t)atol(argv[7]);
    c.insert_mean = (uint32_t)atol(argv[8]);
    c.insert_sd = (uint32_t)atol(argv[9]);
    c.library = (uint32_t)atol(argv[10]);
    const uint64_t n_pairs = strtoull(argv[11], 0, 10);
    c.err_q16 = 328; c.mapq0_q16 = 1311; c.chimeric_q16 = 655; c.flank_len = 300;   /* GapFill.synth_cfg's defaults (0.5 %, 2 %, 1 %) */
    const uint32_t L = c.read_len;
    if (L < 16 || L > 256) die("read_len");
    char path[4096];
    static const char ACGT[4] = {'A', 'C', 'G', 'T'};

    if (c.library == 0) {   /* ---- draft: true bases, planted gaps as N, 60 columns */
        snprintf(path, sizeof path, "%s/draft.fa", out);
        FILE* f = fopen(path, "wb");
        if (!f) die("cannot write draft.fa");
        char* line = malloc(c.scaffold_len + c.scaffold_len / 60 + 64);
        for (uint32_t s = 0; s < c.n_scaffolds; ++s) {
            fprintf(f, ">scf%u\n", s);
            uint64_t o = 0;
            uint32_t g = 0;
            for (uint64_t p = 0; p < c.scaffold_len; ++p) {
                while (g < c.gaps_per_scaffold && p >= gfs_gap_start(&c, g) + c.gap_len) ++g;
                const int in_gap = g < c.gaps_per_scaffold && p >= gfs_gap_start(&c, g);
                line[o++] = in_gap ? 'N' : ACGT[gfs_base(&c, s, p)];
                if (p % 60 == 59 || p + 1 == c.scaffold_len) line[o++] = '\n';
            }
            fwrite(line, 1, o, f);
        }
        free(line);
        fclose(f);
    }

    /* ---- FASTQ pair: record p of file m+1 = mate m+1 of pair p */
    const size_t rec_max = 32 + 2 * (size_t)L + 8;
    for (int m = 0; m < 2; ++m) {
        snprintf(path, sizeof path, "%s/lib%u_%d.fq", out, c.library, m + 1);
        FILE* f = fopen(path, "wb");
        if (!f) die("cannot write FASTQ");
        const size_t slab = 1 << 16;
        char* buf = malloc(slab * rec_max);
        for (uint64_t p0 = 0; p0 < n_pairs; p0 += slab) {
            const uint64_t p1 = p0 + slab < n_pairs ? p0 + slab : n_pairs;
            size_t* len = malloc((p1 - p0) * sizeof(size_t));
#pragma omp parallel for schedule(static)
            for (uint64_t p = p0; p < p1; ++p) {
                gfs_pair pr;
                gfs_errs er;
                gfs_make_pair(&c, p, &pr);
                const int end = (m == 0) ? (pr.flip ? 1 : 0) : (pr.flip ? 0 : 1);      /* mate 1 is the forward end unless flipped */
                gfs_make_errs(&c, pr.err[end], &er);
                char* o = buf + (p - p0) * rec_max;
                int n = sprintf(o, "@r%llu/%d\n", (unsigned long long)p, m + 1);
                for (uint32_t i = 0; i < L; ++i) o[n++] = ACGT[gfs_read_base(&c, &pr, &er, end, i)];
                o[n++] = '\n'; o[n++] = '+'; o[n++] = '\n';
                memset(o + n, 'I', L);
                n += L;
                o[n++] = '\n';
                len[p - p0] = (size_t)n;
            }
            for (uint64_t p = p0; p < p1; ++p) fwrite(buf + (p - p0) * rec_max, 1, len[p - p0], f);
            free(len);
        }
        free(buf);
        fclose(f);
    }

    /* ---- BAM: records of gfs_make_records, coordinate-sorted (unplaced pairs last) */
    const uint64_t n_rec = 2 * n_pairs;
    sortkey* keys = malloc(n_rec * sizeof(sortkey));
    if (!keys) die("out of memory");
#pragma omp parallel for schedule(static)
    for (uint64_t p = 0; p < n_pairs; ++p) {
        gfs_pair pr;
        uint32_t r[2][8];
        gfs_make_pair(&c, p, &pr);
        gfs_make_records(&c, p, &pr, r);
        for (int e = 0; e < 2; ++e) {
            keys[2 * p + e].ref = r[e][3];
            keys[2 * p + e].pos = r[e][0];
            keys[2 * p + e].idx = 2 * p + e;
            keys[2 * p + e].meta = r[e][5];
        }
    }
    qsort(keys, n_rec, sizeof(sortkey), cmp_key);
    /* header */
    size_t cap = 1 << 20, n = 0;
    uint8_t* bam = malloc(cap);
    {
        char* text = malloc(64 + 64 * (size_t)c.n_scaffolds);
        int tl = sprintf(text, "@HD\tVN:1.6\tSO:coordinate\n");
        for (uint32_t s = 0; s < c.n_scaffolds; ++s) tl += sprintf(text + tl, "@SQ\tSN:scf%u\tLN:%llu\n", s, (unsigned long long)c.scaffold_len);
        cap = (size_t)tl + 64 * (size_t)c.n_scaffolds + 64 + n_rec * (36 + 24 + 12 + (L + 1) / 2 + L);
        bam = realloc(bam, cap);
        if (!bam) die("out of memory");
        memcpy(bam, "BAM\1", 4);
        put32(bam + 4, (uint32_t)tl);
        memcpy(bam + 8, text, tl);
        n = 8 + tl;
        put32(bam + n, c.n_scaffolds);
        n += 4;
        for (uint32_t s = 0; s < c.n_scaffolds; ++s) {
            char nm[32];
            const int l = sprintf(nm, "scf%u", s) + 1;
            put32(bam + n, (uint32_t)l);
            memcpy(bam + n + 4, nm, l);
            put32(bam + n + 4 + l, (uint32_t)c.scaffold_len);
            n += 8 + l;
        }
        free(text);
    }
    /* record sizes -> offsets -> parallel fill */
    uint64_t* off = malloc((n_rec + 1) * sizeof(uint64_t));
    off[0] = n;
    for (uint64_t i = 0; i < n_rec; ++i) {
        uint64_t p = keys[i].idx >> 1;
        int l_name = 3;                                  /* 'r', one digit, NUL */
        while (p >= 10) { p /= 10; ++l_name; }
        const uint32_t flag = keys[i].meta & 0xFFFF, clip = keys[i].meta >> 24;
        const int n_cig = (flag & 4) ? 0 : (clip ? 2 : 1);
        off[i + 1] = off[i] + 36 + l_name + 4 * n_cig + (L + 1) / 2 + L;
    }
    if (off[n_rec] > cap) die("BAM size estimate too small");
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n_rec; ++i) {
        const uint64_t p = keys[i].idx >> 1;
        const int e = (int)(keys[i].idx & 1);
        gfs_pair pr;
        gfs_errs er;
        uint32_t r[2][8];
        gfs_make_pair(&c, p, &pr);
        gfs_make_records(&c, p, &pr, r);
        gfs_make_errs(&c, pr.err[e], &er);
        uint8_t* b = bam + off[i];
        char nm[32];
        const int l_name = sprintf(nm, "r%llu", (unsigned long long)p) + 1;
        const uint32_t flag = r[e][5] & 0xFFFF, mapq = (r[e][5] >> 16) & 0xFF, clip = r[e][5] >> 24;
        const int unmapped = (flag & 4) != 0;
        const int n_cig = unmapped ? 0 : (clip ? 2 : 1);
        const uint32_t al = unmapped ? 0 : aligned_len(&c, pr.p[e]);
        const int32_t ref = r[e][3] == 0xFFFFFFFFu ? -1 : (int32_t)r[e][3], mref = r[e][4] == 0xFFFFFFFFu ? -1 : (int32_t)r[e][4];
        const int32_t pos0 = ref < 0 ? -1 : (int32_t)r[e][0] - 1, mpos0 = mref < 0 ? -1 : (int32_t)r[e][1] - 1;
        put32(b, (uint32_t)(off[i + 1] - off[i] - 4));
        put32(b + 4, (uint32_t)ref);
        put32(b + 8, (uint32_t)pos0);
        b[12] = (uint8_t)l_name;
        b[13] = (uint8_t)mapq;
        put16(b + 14, (uint32_t)(pos0 < 0 ? 4680 : reg2bin(pos0, pos0 + (unmapped ? 1 : (int64_t)al))));
        put16(b + 16, (uint32_t)n_cig);
        put16(b + 18, flag);
        put32(b + 20, L);
        put32(b + 24, (uint32_t)mref);
        put32(b + 28, (uint32_t)mpos0);
        put32(b + 32, r[e][2]);
        memcpy(b + 36, nm, l_name);
        uint8_t* q = b + 36 + l_name;
        if (n_cig == 1) put32(q, (L << 4) | 0);                                       /* {L}M */
        else if (n_cig == 2) {
            if (clip == 1) { put32(q, ((L - al) << 4) | 4); put32(q + 4, (al << 4) | 0); }   /* {L-al}S{al}M */
            else { put32(q, (al << 4) | 0); put32(q + 4, ((L - al) << 4) | 4); }             /* {al}M{L-al}S */
        }
        q += 4 * n_cig;
        /* SEQ as SAM prints it: the read itself, reverse-complemented when FLAG has 0x10 (every reverse end here) */
        static const uint8_t NIB[4] = {1, 2, 4, 8};
        memset(q, 0, (L + 1) / 2);
        for (uint32_t i = 0; i < L; ++i) {
            const uint32_t rb = (flag & 0x10) ? 3u - gfs_read_base(&c, &pr, &er, e, L - 1 - i) : gfs_read_base(&c, &pr, &er, e, i);
            q[i >> 1] |= (uint8_t)(NIB[rb] << ((i & 1) ? 0 : 4));
        }
        memset(q + (L + 1) / 2, 40, L);
    }
    n = off[n_rec];
    free(off);
    free(keys);
    /* ---- BGZF: 0xff00-byte blocks, raw deflate level 1, compressed in parallel */
    const size_t BLK = 0xff00, n_blk = (n + BLK - 1) / BLK;
    uint8_t** cbuf = malloc(n_blk * sizeof(uint8_t*));
    size_t* clen = malloc(n_blk * sizeof(size_t));
#pragma omp parallel for schedule(dynamic, 16)
    for (size_t k = 0; k < n_blk; ++k) {
        const size_t a = k * BLK, len = a + BLK < n ? BLK : n - a;
        uint8_t* o = malloc(BLK + 1024);
        z_stream z;
        memset(&z, 0, sizeof z);
        if (deflateInit2(&z, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) die("deflateInit2");
        z.next_in = bam + a; z.avail_in = (uInt)len;
        z.next_out = o + 18; z.avail_out = (uInt)(BLK + 1024 - 26);
        if (deflate(&z, Z_FINISH) != Z_STREAM_END) die("deflate");
        const size_t cl = z.total_out;
        deflateEnd(&z);
        static const uint8_t H[12] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0};
        memcpy(o, H, 12);
        o[12] = 'B'; o[13] = 'C'; o[14] = 2; o[15] = 0;
        put16(o + 16, (uint32_t)(cl + 25));
        put32(o + 18 + cl, (uint32_t)crc32(crc32(0, 0, 0), bam + a, (uInt)len));
        put32(o + 22 + cl, (uint32_t)len);
        cbuf[k] = o;
        clen[k] = cl + 26;
    }
    snprintf(path, sizeof path, "%s/lib%u.bam", out, c.library);
    FILE* f = fopen(path, "wb");
    if (!f) die("cannot write BAM");
    for (size_t k = 0; k < n_blk; ++k) { fwrite(cbuf[k], 1, clen[k], f); free(cbuf[k]); }
    static const uint8_t EOFB[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    fwrite(EOFB, 1, 28, f);
    fclose(f);
    free(bam);
    fprintf(stderr, "synth_files: %llu pairs, %llu alignment records, %zu BAM bytes before compression\n", (unsigned long long)n_pairs,
            (unsigned long long)n_rec, n);
    return 0;
}
