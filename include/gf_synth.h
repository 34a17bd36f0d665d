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
    uint32_t repeats;           /* 0 = i.i.d. sequence everywhere.  Otherwise the stress workload: bits 0-7 = period P >= 4,
                                 * bits 8-15 = copies C: gap classes by (global gap index) mod P — class 0: the region at the gap's LEFT
                                 * edge is a copy of a repeat family shared by C such gaps (0.5-5 kb, ending up to 200 bases before the
                                 * gap or reaching up to 400 bases into it; every other copy reverse-complemented); classes 1 and 2: the
                                 * two of them share a 2-copy repeat at the gap's RIGHT edge; class 3: a low-complexity run (unit of 1-3
                                 * bases, 40-99 bases long) inside the left flank; the other classes stay unique.  Reads that lie wholly
                                 * inside a repeat copy are reported with MAPQ 0, as an aligner would. */
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
GFS_HD uint64_t gfs_gap_start(const gf_synth_cfg* c, uint32_t j) { /* j-th gap (0-based) of any scaffold */
    return (uint64_t)(j + 1) * c->scaffold_len / (c->gaps_per_scaffold + 1) - c->gap_len / 2;
}

/* ---- planted repeats (cfg.repeats != 0): what, if anything, is planted at gap j of a scaffold */
#define GFS_REP_MAXLEN 5000u
typedef struct {
    uint32_t kind;     /* 0 nothing, 1 copy of a repeat family, 2 low-complexity run */
    uint32_t rev;      /* kind 1: this copy is the reverse complement of the family sequence */
    uint64_t lo, hi;   /* scaffold interval [lo, hi) */
    uint64_t fam;      /* kind 1: family id; kind 2: the run's hash (unit length and bases) */
} gfs_rep;
GFS_HD void gfs_gap_repeat(const gf_synth_cfg* c, uint32_t scaffold, uint32_t j, gfs_rep* o) {
    const uint32_t P = c->repeats & 0xFFu, C = (c->repeats >> 8) & 0xFFu;
    o->kind = 0; o->rev = 0; o->lo = o->hi = 0; o->fam = 0;
    if (P < 4) return;
    const uint64_t gid = (uint64_t)scaffold * c->gaps_per_scaffold + j, grp = gid / P;
    const uint32_t cls = (uint32_t)(gid % P);
    const uint64_t gs = gfs_gap_start(c, j), ge = gs + c->gap_len;
    if (cls <= 2) {
        o->kind = 1;
        o->fam = cls == 0 ? 1 + grp / (C ? C : 1u) : 0x4000000000ull + grp;
        const uint64_t h = gfs_mix(c->seed ^ gfs_mix(0xF00D5EED00000000ull + o->fam));
        const uint64_t len = 500 + h % (GFS_REP_MAXLEN - 499);
        const int64_t e = (int64_t)((h >> 24) % 600) - 200;      /* > 0: reaches e bases into the gap; < 0: ends -e bases before it */
        o->rev = (uint32_t)(gfs_mix(gid ^ h) & 1u);
        if (cls == 0) { o->hi = (uint64_t)((int64_t)gs + e); o->lo = o->hi - len; }
        else { o->lo = (uint64_t)((int64_t)ge - e); o->hi = o->lo + len; }
    } else if (cls == 3) {
        o->kind = 2;
        o->fam = gfs_mix(c->seed ^ gfs_mix(0x10C0AAAA00000000ull + gid));
        const uint64_t run = 40 + (o->fam >> 8) % 60;
        o->hi = gs - 20 - (o->fam >> 16) % 100;
        o->lo = o->hi - run;
    }
}
/* base of a planted interval at scaffold position pos (lo <= pos < hi) */
GFS_HD uint32_t gfs_rep_base(const gf_synth_cfg* c, const gfs_rep* r, uint64_t pos) {
    if (r->kind == 2) {
        const uint32_t ul = 1 + (uint32_t)(r->fam % 3);
        return (uint32_t)(r->fam >> (32 + 2 * ((pos - r->lo) % ul))) & 3u;
    }
    const uint64_t off = r->rev ? (r->hi - 1 - pos) : (pos - r->lo);
    const uint32_t b = (uint32_t)(gfs_mix(c->seed ^ gfs_mix(0x5EC0000000000000ull ^ (r->fam << 24) ^ (off >> 5))) >> (2 * (off & 31))) & 3u;
    return r->rev ? 3u - b : b;
}
/* the planted interval that holds pos, if any (intervals stay within GFS_REP_MAXLEN + 400 bases of their gap and gaps are further
 * apart than twice that: only the two gaps next to pos can matter) */
GFS_HD int gfs_find_repeat(const gf_synth_cfg* c, uint32_t scaffold, uint64_t pos, gfs_rep* o) {
    const uint64_t jj = pos * (c->gaps_per_scaffold + 1) / c->scaffold_len;   /* gaps jj-1 (left of pos) and jj (right of it) */
    for (int d = -1; d <= 0; ++d) {
        const int64_t j = (int64_t)jj + d;
        if (j < 0 || j >= (int64_t)c->gaps_per_scaffold) continue;
        gfs_gap_repeat(c, scaffold, (uint32_t)j, o);
        if (o->kind && pos >= o->lo && pos < o->hi) return 1;
    }
    o->kind = 0;
    return 0;
}

GFS_HD uint32_t gfs_base(const gf_synth_cfg* c, uint32_t scaffold, uint64_t pos) {
    if (c->repeats) {
        gfs_rep r;
        if (gfs_find_repeat(c, scaffold, pos, &r)) return gfs_rep_base(c, &r, pos);
    }
    return (uint32_t)(gfs_block(c, scaffold, pos >> 5) >> (2 * (pos & 31))) & 3u;
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
    uint32_t in_rep[2] = {0, 0};   /* the end lies wholly inside a copy of a repeat family: an aligner reports MAPQ 0 */
    if (c->repeats)
        for (int i = 0; i < 2; ++i) {
            gfs_rep r;
            in_rep[i] = gfs_find_repeat(c, p->s[i], p->p[i], &r) && r.kind == 1 && p->p[i] + L <= r.hi;
        }
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
        const uint32_t mapq = m ? ((p->mapq0[i] || in_rep[i]) ? 0u : 60u) : 0u;
        const uint32_t clip = m ? al[i].clipflag : 0u;
        const uint64_t rid = 2 * pair + mate_no;
        out[i][0] = pos; out[i][1] = mpos; out[i][2] = (uint32_t)tlen; out[i][3] = ref; out[i][4] = mref;
        out[i][5] = (flag & 0xFFFFu) | (mapq << 16) | (clip << 24);
        out[i][6] = (uint32_t)rid; out[i][7] = (uint32_t)(rid >> 32);
    }
}

#endif /* GF_SYNTH_H */
