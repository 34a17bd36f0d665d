// index.hip — the flank k-mer index: cache per k, and the HOST construction of it (std::sort + upload).  The product builds the
// index on the device (index_dev.hip); this host builder is kept as its comparator (option "index_host" = 1,
// tests/test_gpu_parity.py::test_device_built_flank_index_equals_the_host_built_one) — 4.6 s for the human-scale layout.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "gf_internal.hpp"

namespace gf {

namespace {

struct Entry {
    K128 key;
    uint32_t gap;
};

struct Occ {        // one occurrence of a canonical 16-mer in a flank
    uint32_t key;
    uint32_t fid;   // 2 * gap + side
    uint32_t info;  // pos | strand << 16 | (last << 17, set later) | left room << 18 | right room << 24
};

inline bool is_acgt(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

int ceil_log2(size_t v) {
    int l = 0;
    while (((size_t)1 << l) < v) ++l;
    return l;
}

// canonical k-mers (and canonical 16-mers inside them) of one flank; k-mers touching a non-ACGT byte are skipped
void extract(const std::string& s, int k, uint32_t gap, std::vector<Entry>& kmers, std::vector<uint32_t>& s16,
             uint32_t fid, std::vector<Occ>& occ) {
    const int n = (int)s.size();
    int run = 0;  // length of the current ACGT run ending at i
    for (int i = 0; i < n; ++i) {
        run = is_acgt(s[i]) ? run + 1 : 0;
        if (run >= k) {
            K128 f{0, 0};
            const int p = i - k + 1;
            for (int j = 0; j < k; ++j) {
                uint64_t c = base_code(s[p + j]);
                if (j < 32) f.hi |= c << (62 - 2 * j);
                else f.lo |= c << (62 - 2 * (j - 32));
            }
            kmers.push_back({canonical(f, k), gap});
        }
    }
    // 16-mers: every 16-mer fully inside a maximal ACGT run of length >= k
    int i = 0;
    while (i < n) {
        if (!is_acgt(s[i])) { ++i; continue; }
        int j = i;
        while (j < n && is_acgt(s[j])) ++j;
        if (j - i >= k) {
            uint32_t w = 0;
            for (int q = i; q < j; ++q) {
                w = (w << 2) | base_code(s[q]);
                if (q - i + 1 >= 16) {
                    const uint32_t key = canon16(w);
                    s16.push_back(key);
                    const int pos = q - 15;                       // the 16-mer occupies [pos, pos + 16) of the run [i, j)
                    const uint32_t lroom = (uint32_t)std::min(63, pos - i), rroom = (uint32_t)std::min(63, j - (pos + 16));
                    occ.push_back({key, fid, (uint32_t)pos | (key != w ? 1u << 16 : 0u) | lroom << 18 | rroom << 24});
                }
            }
        }
        i = j;
    }
}

}  // namespace

void free_flank_index(gf_ctx*, FlankIndex& ix) {
    if (ix.d_bitmap) (void)hipFree(ix.d_bitmap);
    if (ix.d_bitmap_lds) (void)hipFree(ix.d_bitmap_lds);
    if (ix.d_bitmap_mid) (void)hipFree(ix.d_bitmap_mid);
    if (ix.d_sset) (void)hipFree(ix.d_sset);
    if (ix.d_table) (void)hipFree(ix.d_table);
    if (ix.d_sval) (void)hipFree(ix.d_sval);
    if (ix.d_occ) (void)hipFree(ix.d_occ);
    if (ix.d_fpk) (void)hipFree(ix.d_fpk);
    if (ix.d_foff) (void)hipFree(ix.d_foff);
    if (ix.d_sgrp) (void)hipFree(ix.d_sgrp);
    ix = FlankIndex();
}

int build_flank_index(gf_ctx* ctx, int k, FlankIndex** out) {
    auto it = ctx->index.find(k);
    if (it != ctx->index.end() && it->second.max_gaps_per_kmer == ctx->max_gaps_per_kmer) {
        *out = &it->second;
        return GF_OK;
    }
    if (it != ctx->index.end()) {
        free_flank_index(ctx, it->second);
        ctx->index.erase(it);
    }
    if (k < 16 || k > 64) return GF_E_UNSUPPORTED;
    if (!ctx->index_host) {   // the product path: built on the device (index_dev.hip); what follows is the host comparator
        FlankIndex dev;
        const int rc = build_flank_index_dev(ctx, k, dev);
        if (rc) {
            free_flank_index(ctx, dev);
            return rc;
        }
        auto ins = ctx->index.emplace(k, dev);
        *out = &ins.first->second;
        return GF_OK;
    }

    std::vector<Entry> ent;
    std::vector<uint32_t> s16;
    std::vector<Occ> occ;
    const size_t ng = ctx->gaps.size();
    ent.reserve(ng * 600);
    occ.reserve(ng * 600);
    for (size_t g = 0; g < ng; ++g) {
        extract(ctx->flank_left[g], k, (uint32_t)g, ent, s16, (uint32_t)(2 * g), occ);
        extract(ctx->flank_right[g], k, (uint32_t)g, ent, s16, (uint32_t)(2 * g + 1), occ);
    }
    std::sort(ent.begin(), ent.end(), [](const Entry& a, const Entry& b) {
        return a.key < b.key || (a.key == b.key && a.gap < b.gap);
    });
    ent.erase(std::unique(ent.begin(), ent.end(), [](const Entry& a, const Entry& b) { return a.key == b.key && a.gap == b.gap; }),
              ent.end());
    if (ctx->max_gaps_per_kmer) {  // drop k-mers shared by more than max_gaps_per_kmer gaps (repeat mask)
        std::vector<Entry> kept;
        kept.reserve(ent.size());
        size_t i = 0;
        while (i < ent.size()) {
            size_t j = i;
            while (j < ent.size() && ent[j].key == ent[i].key) ++j;
            if (j - i <= ctx->max_gaps_per_kmer) kept.insert(kept.end(), ent.begin() + i, ent.begin() + j);
            i = j;
        }
        ent.swap(kept);
    }
    std::sort(s16.begin(), s16.end());
    s16.erase(std::unique(s16.begin(), s16.end()), s16.end());

    FlankIndex ix;
    ix.k = k;
    ix.stride = k - 15;
    ix.n_kmers = ent.size();
    ix.n_s16 = s16.size();
    ix.n_occ = occ.size();
    ix.max_gaps_per_kmer = ctx->max_gaps_per_kmer;

    // level 3 table
    ix.t_log2 = std::max(8, ceil_log2(2 * ent.size() + 2));
    const size_t tcap = (size_t)1 << ix.t_log2;
    const bool wide = k > 32;
    const size_t slot_words = wide ? 8 : 4;
    std::vector<uint32_t> tab(tcap * slot_words, 0);
    for (size_t i = 0; i < tcap; ++i) tab[i * slot_words + (wide ? 4 : 2)] = EMPTY32;
    for (const Entry& e : ent) {
        uint32_t sl = hash_kmer(e.key, ix.t_log2);
        while (tab[sl * slot_words + (wide ? 4 : 2)] != EMPTY32) sl = (sl + 1) & (tcap - 1);
        uint32_t* t = &tab[sl * slot_words];
        t[0] = (uint32_t)e.key.hi; t[1] = (uint32_t)(e.key.hi >> 32);
        if (wide) { t[2] = (uint32_t)e.key.lo; t[3] = (uint32_t)(e.key.lo >> 32); t[4] = e.gap; }
        else t[2] = e.gap;
    }
    // level 2 set
    ix.s_log2 = std::max(8, ceil_log2(2 * s16.size() + 2));
    const size_t scap = (size_t)1 << ix.s_log2;
    std::vector<uint32_t> sset(scap + 4, EMPTY32);
    for (uint32_t key : s16) {
        uint32_t s = hash_s16_set(key, ix.s_log2);
        while (sset[s] != EMPTY32) s = (s + 1) & (scap - 1);
        sset[s] = key;
    }
    for (int i = 0; i < 4; ++i) sset[scap + i] = sset[i];   // wrap-around copy: a 4-slot read never needs the modulo
    // occurrence lists for the seed-and-extend verification: sorted by key, the last of a key flagged, sval[slot] = first
    std::sort(occ.begin(), occ.end(), [](const Occ& a, const Occ& b) {
        return a.key < b.key || (a.key == b.key && (a.fid < b.fid || (a.fid == b.fid && a.info < b.info)));
    });
    std::vector<uint32_t> sval(scap + 4, 0), occw(2 * occ.size() + 2, 0);
    for (size_t i = 0; i < occ.size(); ++i) {
        const bool first = i == 0 || occ[i - 1].key != occ[i].key, last = i + 1 == occ.size() || occ[i + 1].key != occ[i].key;
        if (first) {
            uint32_t s = hash_s16_set(occ[i].key, ix.s_log2);
            while (sset[s] != occ[i].key) s = (s + 1) & (scap - 1);
            sval[s] = (uint32_t)i;
        }
        occw[2 * i] = occ[i].fid;
        occw[2 * i + 1] = occ[i].info | (last ? 1u << 17 : 0u);
    }
    for (int i = 0; i < 4; ++i) sval[scap + i] = sval[i];
    // flanks, 2 bits per base (non-ACGT -> A; the room fields keep extensions inside ACGT runs), 4 zero words around each
    std::vector<uint32_t> foff(2 * ng + 1, 0), fpk;
    for (size_t f = 0; f < 2 * ng; ++f) {
        const std::string& fs = (f & 1) ? ctx->flank_right[f >> 1] : ctx->flank_left[f >> 1];
        fpk.insert(fpk.end(), 4, 0u);
        foff[f] = (uint32_t)fpk.size();
        const size_t nw = (fs.size() + 15) / 16;
        const size_t base = fpk.size();
        fpk.insert(fpk.end(), nw, 0u);
        for (size_t i = 0; i < fs.size(); ++i) fpk[base + i / 16] |= base_code(fs[i]) << (30 - 2 * (i % 16));
        fpk.insert(fpk.end(), 4, 0u);
    }
    foff[2 * ng] = (uint32_t)fpk.size();
    ix.ext_ok = true;
    for (size_t g = 0; g < ng; ++g) ix.ext_ok = ix.ext_ok && ctx->flank_left[g].size() < 65536 && ctx->flank_right[g].size() < 65536;
    if (fpk.empty()) fpk.push_back(0);
    // level 1 bitmap: at least 16 bits per key (<= 6 % false positives per probe, resolved by level 2) and never
    // fewer than 2^24 bits: the filter kernels test a 2^20-bit coarse copy in LDS first, and a level-1 bitmap only a few
    // times finer than that copy stops too little of what the copy lets through (measured on MI355X, 50 M reads:
    // 100-400 gaps 1.5-1.6 ms with 16 bits per key, 0.9-1.1 ms with 2^24 bits).  A 2 MiB bitmap stays in every XCD's
    // L2; a sparser 4 MiB one measured slower.
    int bl = std::max(24, ceil_log2(16 * s16.size() + 1));
    bl = std::min(28, bl);
    if (ctx->bitmap_log2_override) bl = std::min(31, std::max(10, ctx->bitmap_log2_override));
    ix.bm_log2 = bl;
    const size_t bwords = ((size_t)1 << bl) / 32;
    std::vector<uint32_t> bm(bwords, 0);
    for (uint32_t key : s16) {
        uint32_t h = hash_s16_bitmap(key, bl);
        bm[h >> 5] |= 1u << (h & 31);
        bm[h >> 5] |= 1u << hash_s16_bit2(key);   // second bit in the same word (blocked Bloom): kernels may test it too
        if (bl >= S16_BIT3_MIN_LOG2) bm[h >> 5] |= 1u << hash_s16_bit3(key, bl);   // ... and a third one in the large bitmaps
    }

    // coarse copy for the LDS pre-filter: bit c = OR of the level-1 bits h with (h >> (bl - lds_log2)) == c
    ix.lds_log2 = std::min(ctx->screen_lds_log2_max, bl);
    const size_t cwords = ((size_t)1 << ix.lds_log2) / 32;
    std::vector<uint32_t> cbm(cwords, 0);
    size_t cset = 0;
    for (uint32_t key : s16) {
        const uint32_t c = hash_s16_bitmap(key, bl) >> (bl - ix.lds_log2);
        if (!((cbm[c >> 5] >> (c & 31)) & 1u)) ++cset;
        cbm[c >> 5] |= 1u << (c & 31);
    }
    ix.lds_fill = (double)cset / (double)((size_t)1 << ix.lds_log2);

    GF_HIP(ctx, hipSetDevice(ctx->device));
    GF_HIP(ctx, hipMalloc((void**)&ix.d_bitmap_lds, cwords * 4));
    GF_HIP(ctx, hipMemcpy(ix.d_bitmap_lds, cbm.data(), cwords * 4, hipMemcpyHostToDevice));
    // key sets whose level-1 bitmap outgrows the L2 (> 2^24 bits) get a 2^24-bit OR-reduction of it that does stay there:
    // the plain filter kernel asks it first and sends only what it lets through to the big bitmap
    ix.mid_log2 = 0;
    if (bl > 24) {
        ix.mid_log2 = 24;
        const size_t mwords = ((size_t)1 << 24) / 32;
        std::vector<uint32_t> mbm(mwords, 0);
        size_t mset = 0;
        for (uint32_t key : s16) {
            const uint32_t c = hash_s16_bitmap(key, bl) >> (bl - 24);
            if (!((mbm[c >> 5] >> (c & 31)) & 1u)) ++mset;
            mbm[c >> 5] |= 1u << (c & 31);
        }
        ix.mid_fill = (double)mset / (double)((size_t)1 << 24);
        GF_HIP(ctx, hipMalloc((void**)&ix.d_bitmap_mid, mwords * 4));
        GF_HIP(ctx, hipMemcpy(ix.d_bitmap_mid, mbm.data(), mwords * 4, hipMemcpyHostToDevice));
    }
    GF_HIP(ctx, hipMalloc((void**)&ix.d_bitmap, bwords * 4));
    GF_HIP(ctx, hipMalloc((void**)&ix.d_sset, (scap + 4) * 4));
    GF_HIP(ctx, hipMalloc(&ix.d_table, tab.size() * 4));
    GF_HIP(ctx, hipMemcpy(ix.d_bitmap, bm.data(), bwords * 4, hipMemcpyHostToDevice));
    GF_HIP(ctx, hipMemcpy(ix.d_sset, sset.data(), (scap + 4) * 4, hipMemcpyHostToDevice));
    GF_HIP(ctx, hipMalloc((void**)&ix.d_sval, sval.size() * 4));
    GF_HIP(ctx, hipMemcpy(ix.d_sval, sval.data(), sval.size() * 4, hipMemcpyHostToDevice));
    GF_HIP(ctx, hipMalloc((void**)&ix.d_occ, occw.size() * 4));
    GF_HIP(ctx, hipMemcpy(ix.d_occ, occw.data(), occw.size() * 4, hipMemcpyHostToDevice));
    GF_HIP(ctx, hipMalloc((void**)&ix.d_fpk, fpk.size() * 4));
    GF_HIP(ctx, hipMemcpy(ix.d_fpk, fpk.data(), fpk.size() * 4, hipMemcpyHostToDevice));
    GF_HIP(ctx, hipMalloc((void**)&ix.d_foff, foff.size() * 4));
    GF_HIP(ctx, hipMemcpy(ix.d_foff, foff.data(), foff.size() * 4, hipMemcpyHostToDevice));
    GF_HIP(ctx, hipMemcpy(ix.d_table, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    {
        const int rc = build_sgrp_dev(ctx, ix);
        if (rc) { free_flank_index(ctx, ix); return rc; }
    }
    auto ins = ctx->index.emplace(k, ix);
    *out = &ins.first->second;
    return GF_OK;
}

}  // namespace gf
