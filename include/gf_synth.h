/* gf_synth.h — definition of the seeded synthetic workload: a hash-defined random genome with
 * evenly planted gaps, FR read pairs with substitution errors, and truth-derived alignment records.
 *
 * This is a WORKLOAD DEFINITION, not part of the reference's path: integer-only and stateless (every base,
 * pair and record is a pure function of (cfg, index)), so the HIP generator (gappadder_amd/csrc/synth.hip),
 * the host helpers and the CPU oracle (oracle/gp_oracle.c) produce bit-identical data on any machine without
 * ever materialising the genome.  PRNG = SplitMix64 finaliser, not Python `random`.
 */
#ifndef GF_SYNTH_H
#define GF_SYNTH_H
#include <stdint.h>

#if defined(__HIPCC__)
#define GFS_HD __host__ __device__ static inline
#else
#define GFS_HD static inline
#endif

typedef struct {
    uint64_t seed;
    uint64_t scaffold_len;      /* bases per scaffold (all scaffolds equal) */
    uint32_t n_scaffolds;
    uint32_t gaps_per_scaffold; /* evenly spaced */
    uint32_t gap_len;
    uint32_t read_len;          /* <= 256 */
    uint32_t insert_mean, insert_sd;
    uint32_t err_q16;           /* per-base substitution probability * 65536 (0.5 % = 328) */
    uint32_t mapq0_q16;         /* fraction of mapped reads reported with MAPQ 0, * 65536 (2 % = 1311) */
    uint32_t chimeric_q16;      /* fraction of pairs whose reverse read comes from a random other place (1 % = 655) */
    uint32_t flank_len;         /* 300 (configuration.json:38) */
    uint32_t library;           /* read library drawn from the SAME genome: 0 = first (reproduces the single-library data), 1, 2 ...
                                 * select independent pair streams (a second `alignments[]` entry of the reference's JSON, e.g. the
                                 * IS 5000 mate-pair library of BASELINE.json configs[4]) */
    uint32_t reserved;          /* 0 */
} gf_synth_cfg;

typedef struct {
    uint32_t s[2];     /* scaffold of the forward / reverse end */
    uint64_t p[2];     /* 0-based leftmost genome coordinate of each end */
    uint32_t flip;     /* 0: forward end is mate 1; 1: reverse end is mate 1 */
    uint32_t mapq0[2];
    uint64_t err[2];   /* error word per end */
} gfs_pair;

GFS_HD uint64_t gfs_mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* 32 true bases (2 bits each, base (pos&31) at bits 2*(pos&31)) of block pos>>5 of a scaffold */
GFS_HD uint64_t gfs_block(const gf_synth_cfg* c, uint32_t scaffold, uint64_t block) {
    return gfs_mix(c->seed ^ gfs_mix(((uint64_t)scaffold << 40) ^ block));
}
GFS_HD uint32_t gfs_base(const gf_synth_cfg* c, uint32_t scaffold, uint64_t pos) {
    return (uint32_t)(gfs_block(c, scaffold, pos >> 5) >> (2 * (pos & 31))) & 3u;
}

GFS_HD uint64_t gfs_gap_start(const gf_synth_cfg* c, uint32_t j) { /* j-th gap (0-based) of any scaffold */
    return (uint64_t)(j + 1) * c->scaffold_len / (c->gaps_per_scaffold + 1) - c->gap_len / 2;
}

GFS_HD void gfs_make_pair(const gf_synth_cfg* c, uint64_t pair, gfs_pair* o) {
    const uint64_t base = gfs_mix(c->seed * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull + (uint64_t)c->library * 0xA0761D6478BD642Full);
    const uint64_t r0 = gfs_mix(base + 4 * pair), r1 = gfs_mix(base + 4 * pair + 1), r2 = gfs_mix(base + 4 * pair + 2),
                   r3 = gfs_mix(base + 4 * pair + 3);
    const int64_t L = c->read_len;
    int64_t t = (int64_t)(r1 & 0xFFFF) + (int64_t)((r1 >> 16) & 0xFFFF) + (int64_t)((r1 >> 32) & 0xFFFF) +
                (int64_t)((r1 >> 48) & 0xFFFF) - 131070; /* ~N(0, 37837) */
    int64_t ins = (int64_t)c->insert_mean + (int64_t)c->insert_sd * t / 37837;
    if (ins < L + 1) ins = L + 1;
    if (ins > (int64_t)c->scaffold_len - 1) ins = (int64_t)c->scaffold_len - 1;
    o->s[0] = (uint32_t)(r0 % c->n_scaffolds);
    o->p[0] = (r0 >> 20) % (c->scaffold_len - (uint64_t)ins);
    o->s[1] = o->s[0];
    o->p[1] = o->p[0] + (uint64_t)(ins - L);
    if ((r2 & 0xFFFF) < c->chimeric_q16) {
        o->s[1] = (uint32_t)((r2 >> 16) % c->n_scaffolds);
        o->p[1] = gfs_mix(r2) % (c->scaffold_len - (uint64_t)L);
    }
    o->flip = (uint32_t)(r3 & 1);
    o->mapq0[0] = ((r3 >> 8) & 0xFFFF) < c->mapq0_q16;
    o->mapq0[1] = ((r3 >> 24) & 0xFFFF) < c->mapq0_q16;
    o->err[0] = gfs_mix(r3 + 1);
    o->err[1] = gfs_mix(r3 + 2);
}

/* substitution errors of one end: two slots, each active with probability read_len*err/2 */
typedef struct { uint32_t n; uint32_t pos[2]; uint32_t d[2]; } gfs_errs;
GFS_HD void gfs_make_errs(const gf_synth_cfg* c, uint64_t e, gfs_errs* o) {
    const uint32_t pe = (uint32_t)(((uint64_t)c->read_len * c->err_q16) / 2);
    const uint64_t e2 = gfs_mix(e);
    o->n = 0;
    if ((e & 0xFFFF) < pe) { o->pos[o->n] = (uint32_t)((e >> 16) & 0xFFFF) % c->read_len; o->d[o->n] = 1 + (uint32_t)((e >> 32) & 0xFF) % 3; o->n++; }
    if ((e2 & 0xFFFF) < pe) { o->pos[o->n] = (uint32_t)((e2 >> 16) & 0xFFFF) % c->read_len; o->d[o->n] = 1 + (uint32_t)((e2 >> 32) & 0xFF) % 3; o->n++; }
}

/* base i (read orientation) of end `end` (0 forward, 1 reverse-complemented) */
GFS_HD uint32_t gfs_read_base(const gf_synth_cfg* c, const gfs_pair* p, const gfs_errs* er, int end, uint32_t i) {
    uint32_t b;
    if (end == 0) b = gfs_base(c, p->s[0], p->p[0] + i);
    else b = 3u - gfs_base(c, p->s[1], p->p[1] + (c->read_len - 1 - i));
    for (uint32_t k = 0; k < er->n; ++k)
        if (er->pos[k] == i) b = (b + er->d[k]) & 3u;
    return b;
}

/* truth -> alignment of one end: aligned part = the longer stretch outside the (single) overlapping gap, >= 20 bp */
typedef struct { uint32_t mapped; uint32_t pos1; uint32_t clipflag; } gfs_aln;
GFS_HD void gfs_align(const gf_synth_cfg* c, uint64_t lo, gfs_aln* o) {
    const uint64_t L = c->read_len, hi = lo + L;
    o->mapped = 1; o->pos1 = (uint32_t)(lo + 1); o->clipflag = 0;
    /* nearest gaps by index */
    uint64_t jj = lo * (c->gaps_per_scaffold + 1) / c->scaffold_len; /* gap index + 1, approx */
    for (int d = -1; d <= 1; ++d) {
        int64_t j = (int64_t)jj - 1 + d;
        if (j < 0 || j >= (int64_t)c->gaps_per_scaffold) continue;
        const uint64_t gs = gfs_gap_start(c, (uint32_t)j), ge = gs + c->gap_len;
        if (ge <= lo || gs >= hi) continue;
        const uint64_t left = gs > lo ? gs - lo : 0, right = hi > ge ? hi - ge : 0;
        if (left >= right) {
            if (left >= 20) { o->pos1 = (uint32_t)(lo + 1); o->clipflag = 2; } else o->mapped = 0;
        } else {
            if (right >= 20) { o->pos1 = (uint32_t)(ge + 1); o->clipflag = 1; } else o->mapped = 0;
        }
        return;
    }
}

/* the two 32-byte alignment records of a pair, as 8 x uint32 each (layout of gf_alnrec); `pair` only numbers the read ids
 * (2 * pair + mate): the generators pass the pair's index inside the generated batch, so that record.read indexes the batch's reads */
GFS_HD void gfs_make_records(const gf_synth_cfg* c, uint64_t pair, const gfs_pair* p, uint32_t out[2][8]) {
    gfs_aln al[2];
    gfs_align(c, p->p[0], &al[0]);
    gfs_align(c, p->p[1], &al[1]);
    const uint64_t L = c->read_len;
    for (int i = 0; i < 2; ++i) {
        const int j = 1 - i;
        const uint32_t m = al[i].mapped, mm = al[j].mapped;
        const uint32_t mate_no = ((i == 0) != (p->flip != 0)) ? 0u : 1u; /* 0 = mate 1 */
        uint32_t flag = 1u | (mate_no == 0 ? 0x40u : 0x80u) | (i == 1 ? 0x10u : 0x20u);
        if (!m) flag |= 4u;
        if (!mm) flag |= 8u;
        const uint32_t ref = m ? p->s[i] : (mm ? p->s[j] : 0xFFFFFFFFu);
        const uint32_t pos = m ? al[i].pos1 : (mm ? al[j].pos1 : 0u);
        const uint32_t mref = mm ? p->s[j] : (m ? ref : 0xFFFFFFFFu);
        const uint32_t mpos = mm ? al[j].pos1 : pos;
        int32_t tlen = 0;
        if (m && mm && p->s[0] == p->s[1]) {
            const uint64_t lo = p->p[0] < p->p[1] ? p->p[0] : p->p[1];
            const uint64_t hi = (p->p[0] > p->p[1] ? p->p[0] : p->p[1]) + L;
            const int32_t span = (int32_t)(hi - lo);
            tlen = p->p[i] <= p->p[j] ? span : -span;
        }
        const uint32_t mapq = m ? (p->mapq0[i] ? 0u : 60u) : 0u;
        const uint32_t clip = m ? al[i].clipflag : 0u;
        const uint64_t rid = 2 * pair + mate_no;
        out[i][0] = pos; out[i][1] = mpos; out[i][2] = (uint32_t)tlen; out[i][3] = ref; out[i][4] = mref;
        out[i][5] = (flag & 0xFFFFu) | (mapq << 16) | (clip << 24);
        out[i][6] = (uint32_t)rid; out[i][7] = (uint32_t)(rid >> 32);
    }
}

#endif /* GF_SYNTH_H */
