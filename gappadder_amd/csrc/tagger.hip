// tagger.hip — alignment-record tagging.
//
// tag_kernel restates GapReadsCollector.parse_reads_fall_in_gaps_one_scaffold / _short_is
// (collect_reads_for_gaps.py:68-163 / :166-263) on decoded 32-byte records; the reference's per-position dict
// `focal_region` (get_focal_region_of_scaffold_v2, :31-65) becomes a closed-form window test per gap:
//   left  window: 0 <= start-POS < dist2  (keys start-i, i in range(dist2), start-i >= 0; 'c' if i <= clip_dist)
//   right window: 0 <= POS-end   < dist2  (keys end+i)
// low_mapq_kernel restates collect_discordant_low_mapq_reads.py:4-84: MAPQ==0 records, focal_region[p] = the
// LAST discordant mate position q (sorted file order) with q-199 <= p <= q+299, one hit per row of q.
#include <cstring>

#include "gf_internal.hpp"

namespace gf {

struct TagParams {
    const gf_alnrec* recs;
    uint32_t dbg;             // diagnostics (option tag_dbg; results change): 1 no MAPQ-0 by-product, 2 no bin test (nothing passes), 4 candidates dropped unsifted
    const uint2* keys;        // KEYS variants: {pos, ref | (mapq == 0) << 31} per record (gf_alnrec_keys_dev), n + 1 entries
    uint64_t n;
    const gf_gap* gaps;
    const uint32_t* scaf_off;
    uint32_t n_scaffolds;
    int32_t dist1, dist2, clip_dist, anchor_mapq;
    int32_t short_is;
    // coarse bin map: bit (bin_off[s] + (POS >> bin_shift)) is set iff some window of scaffold s touches that bin
    const uint32_t* bin_bits;
    const uint32_t* bin_off;  // n_scaffolds + 1
    const uint32_t* bin_first; // per bin: index of the first gap (of the scaffold) whose right window reaches the bin's first position
    uint32_t bin_shift, bin_words;
    uint32_t off_words;       // > 0: the n_scaffolds + 1 offsets follow the bits in the staged LDS copy as well
    // fine bin map (global, L2-resident, <= 4 MiB): same test on narrower bins for the records the LDS map lets through.
    // Only built when the LDS map's bins are wide (large genomes, many gaps); null otherwise.
    const uint32_t* fine_bits;
    const uint32_t* fine_off;
    uint32_t fine_shift;
    gf_taghit* out;
    uint32_t cap;
    uint32_t* n_out;
    // optional by-product for the second hop: every MAPQ==0 record as {pos, ref, record index}
    gf_lowrec* low;
    uint32_t low_cap;
    uint32_t* n_low;
    gf_lowrec* low_stage;     // LOWSTAGE entries per wave of the launch: where a wave collects its MAPQ==0 records (see flush_low)
};

// A single global counter serialises returning atomics at ~11 ns each (MI355X_MICROARCH.md "dequeue"/"fanin" rows), so hits are
// collected in an LDS buffer per WAVE (ballot + prefix count, no LDS atomic: the fill is wave-uniform) that leaves with one global
// atomic per 33-96 hits.  (Rounds 1-3 kept one buffer per workgroup, flushed once when the kernel ended, with direct global appends
// once it was full: at human scale a workgroup finds 10 000-25 000 hits, so nearly every hit took that fall-back — 600 000
// returning atomics per launch of the long-insert library, 6 of its 12 ms.)
constexpr uint32_t WHCAP = 96;   // < 33 waiting + <= 64 from one step
struct WaveHits {
    gf_taghit* h;     // this wave's WHCAP entries (LDS)
    uint32_t n;       // wave-uniform
};
__device__ __forceinline__ void tag_wave_sync();
__device__ __forceinline__ void flush_wave_hits(WaveHits& w, gf_taghit* out, uint32_t cap, uint32_t* n_out) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t gb = 0;
    if (lane == 0) gb = atomicAdd(n_out, w.n);
    gb = __shfl(gb, 0);
    for (uint32_t i = lane; i < w.n; i += 64)
        if (gb + i < cap) out[gb + i] = w.h[i];
    w.n = 0;
    tag_wave_sync();
}
__device__ __forceinline__ void emit_hit(bool want, const gf_taghit& h, WaveHits& w, gf_taghit* out, uint32_t cap, uint32_t* n_out) {
    const unsigned long long bal = __ballot(want);
    if (!bal) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t cnt = (uint32_t)__popcll(bal);
    if (w.n + cnt > WHCAP) flush_wave_hits(w, out, cap, n_out);
    if (want) w.h[w.n + __popcll(bal & ((1ull << lane) - 1))] = h;
    w.n += cnt;
    tag_wave_sync();
    if (w.n > WHCAP - 64) flush_wave_hits(w, out, cap, n_out);
}

// Records are streamed as whole 1-KiB wave loads (16 B per lane, consecutive lanes = consecutive 16-B halves):
// the even lane of a pair holds {pos, mate_pos, tlen, ref}, the odd lane {mate_ref, flag|mapq|clip, read id}.
// Two loads = 64 records; lane L gathers record L's fields from the lanes that hold its halves (tag_kernel).
struct LiveRec { uint32_t pos, ref, mate_ref, meta; int32_t tlen; uint32_t rec; };   // a record that passed the bin map
constexpr uint32_t LIVEQ = 96;   // per wave: < 64 waiting + what one step adds (a step that would overflow it drains first)
// With a second, finer map in global memory (human scale) the records that pass the LDS map (a twentieth) are not looked up in it on
// the spot — nearly every 1-KiB step has such a lane, and the whole wave then waits for two dependent global loads (PMC: waves wait
// 79 % of their cycles in a kernel that only streams) — but queued with 12 bytes each and sifted 64 at a time; the few that pass
// load their record again and join the queue of the window search.
struct CandRec { uint32_t pos, ref, rec; };
constexpr uint32_t CANDQ = 96;
__device__ __forceinline__ void tag_wave_sync() {   // LDS hand-off between lanes of ONE wave (in-order LDS: compiler fence only)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
constexpr uint32_t LOWSTAGE = 1024;   // MAPQ==0 records a wave collects before it appends them to the list
constexpr int TAG_UNROLL = 8;  // 8 KiB (256 records) per wave and stage, two stages in flight

// NW waves per workgroup.  BINS_LDS: the coarse bin map is staged in LDS (the stand-alone form: 33 KiB of LDS per workgroup at
// human scale); BINS_LDS = false reads it through L1/L2 instead and runs one-wave workgroups with ~6 KiB of LDS, so that the
// tagger's workgroups fit on the CUs NEXT TO the k-mer filter's (which own 137-151 KiB of every CU's LDS but leave most of its
// issue slots idle: PMC SQ_WAIT_ANY 53-77 %) — the "light" variant a pipeline launches on its second stream.
// KEYS: the stream is the 8-byte KEY COLUMN of the records — {pos, ref | (mapq == 0) << 31}, built once at ingest — instead of the 32-byte
// records themselves: 99 % of the records are decided by (scaffold, position) alone, and the by-product needs one more bit; the full record is
// fetched only for what passes the bin maps (the path the fine-map queue already had).  A quarter of the bytes per record for the stream.
template <int NW, bool BINS_LDS, bool NT = false, bool FINEQ = false, bool KEYS = false>
__global__ __launch_bounds__(64 * NW) void tag_kernel(TagParams P) {
    constexpr bool CQ = FINEQ || KEYS;   // records that pass the LDS map wait in the candidate queue (fine map, then the record's fetch)
    extern __shared__ uint32_t bins_lds[];  // the whole bin map (16 or 64 KiB), staged once per workgroup
    __shared__ gf_taghit whits[NW][WHCAP];
    __shared__ LiveRec liveq[NW][LIVEQ];
    __shared__ CandRec candq[CQ ? NW : 1][CQ ? CANDQ : 1];
    // MAPQ==0 records (2 % of a library) leave through ONE counter, and returning atomics on one address are served at ~11-15 ns
    // each: flushing a 64-entry LDS buffer per wave straight to the list, the 900 M records of C4 needed 360 000 of them — 4 of the
    // kernel's 5.2 ms; the stream ran at 4.9 instead of 6.0 TB/s.  The LDS buffer (whole 768-byte runs: single 12-byte stores left
    // partial lines in L2) now empties into the wave's private slice of a global staging buffer, without an atomic, and the slice
    // moves to the list 1 000 records at a time: a twentieth of the atomics.
    constexpr uint32_t LOWBUF = 64;
    __shared__ gf_lowrec lowbuf[NW][LOWBUF];
    uint32_t low_n = 0, stage_n = 0;    // wave-uniform: records in the LDS buffer, in the staging slice
    WaveHits hb{whits[threadIdx.x >> 6], 0};
    if (BINS_LDS) for (uint32_t i = threadIdx.x; i < P.bin_words + P.off_words; i += blockDim.x) bins_lds[i] = P.bin_bits[i];   // (bits, then offsets: one array)
    const uint32_t* bins = BINS_LDS ? bins_lds : P.bin_bits;
    const uint32_t* boff = BINS_LDS && P.off_words ? bins_lds + P.bin_words : P.bin_off;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    gf_lowrec* const wlow = lowbuf[threadIdx.x >> 6];
    gf_lowrec* const wstage = P.low ? P.low_stage + wave * LOWSTAGE : nullptr;
    auto flush_stage = [&]() {
        uint32_t gb = 0;
        if (lane == 0) gb = atomicAdd(P.n_low, stage_n);
        gb = __shfl(gb, 0);
        const uint32_t fit = gb < P.low_cap ? (stage_n < P.low_cap - gb ? stage_n : P.low_cap - gb) : 0;
        // the wave reads its own stores back from L2 (a store is acknowledged once it is there; the loads bypass the CU's L1,
        // which may still hold the slice as the previous flush read it)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t* src32 = reinterpret_cast<const uint32_t*>(wstage);
        uint32_t* dst32 = reinterpret_cast<uint32_t*>(P.low + gb);
        for (uint32_t i = lane; i < 3 * fit; i += 64) dst32[i] = __hip_atomic_load(src32 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stage_n = 0;
    };
    auto flush_low = [&]() {   // LDS buffer -> staging slice
        if (stage_n + low_n > LOWSTAGE) flush_stage();
        const uint32_t* src32 = reinterpret_cast<const uint32_t*>(wlow);
        uint32_t* dst32 = reinterpret_cast<uint32_t*>(wstage + stage_n);
        for (uint32_t i = lane; i < 3 * low_n; i += 64) dst32[i] = src32[i];
        stage_n += low_n;
        low_n = 0;
    };
    // ---- per-wave queue of records that passed the bin map
    LiveRec* wlive = liveq[threadIdx.x >> 6];
    uint32_t live_n = 0;   // wave-uniform
    auto drain = [&]() {   // window search + tagging of the last min(64, live_n) queued records, one per lane
        const uint32_t base = live_n > 64 ? live_n - 64 : 0;
        bool live = base + lane < live_n;
        gf_alnrec r = {};
        uint32_t rec = 0, g = 0, g_end = 0;
        if (live) {
            const LiveRec q = wlive[base + lane];
            r.pos = q.pos; r.ref = q.ref; r.tlen = q.tlen; r.mate_ref = q.mate_ref;
            r.flag = (uint16_t)(q.meta & 0xFFFF); r.mapq = (uint8_t)((q.meta >> 16) & 0xFF); r.clipflag = (uint8_t)(q.meta >> 24);
            rec = q.rec;
            g_end = P.scaf_off[r.ref + 1];
            // the first gap whose right window reaches the record's BIN (one load; the map's third part) — gaps between it and the first
            // one whose window reaches POS itself (end + dist2 > pos; two gaps within one bin) fall through the walk below untagged
            g = P.bin_first[boff[r.ref] + (r.pos >> P.bin_shift)];
        }
        const int64_t pos = r.pos;
        // walk the (few) gaps whose windows can contain POS; lanes without work idle through the ballots.  The first two gaps are
        // loaded together (the walk ends at the first gap whose left window starts behind POS: as a rule the second one), so the
        // common walk is two dependent round trips — bin -> first gap, gaps — instead of four
        gf_gap gq[2] = {};
        if (live && g < g_end) gq[0] = P.gaps[g];
        if (live && g + 1 < g_end) gq[1] = P.gaps[g + 1];
        for (uint32_t it = 0;; ++it) {
            gf_gap gp = it == 0 ? gq[0] : gq[1];
            if (it >= 2 && live && g < g_end) gp = P.gaps[g];
            const bool more = live && g < g_end && (int64_t)gp.start - P.dist2 < pos;
            live = more;          // (a lane whose walk has ended stays out: every lane still walking is at its first gap + it)
            if (!__any(more)) break;
            bool clip = false, pair = false, unmap = false;
            if (more) {
                const int64_t il = (int64_t)gp.start - pos, ir = pos - (int64_t)gp.end;
                int tag = -1;  // 0: 0c, 1: 0d, 2: 1c, 3: 1d
                if (il >= 0 && il < P.dist2) tag = il <= P.clip_dist ? 0 : 1;
                else if (ir >= 0 && ir < P.dist2) tag = ir <= P.clip_dist ? 2 : 3;
                if (tag >= 0) {
                    clip = (tag == 0 && r.clipflag >= 2) || (tag == 2 && (r.clipflag == 1 || r.clipflag == 3));
                    const bool mapped = (r.flag & 0x4) == 0, mate_mapped = (r.flag & 0x8) == 0;
                    if (mapped && mate_mapped && (int)r.mapq >= P.anchor_mapq) {
                        if (r.mate_ref != r.ref) pair = true;
                        else {
                            const int64_t t = r.tlen < 0 ? -(int64_t)r.tlen : (int64_t)r.tlen;
                            pair = t >= P.dist2 || (P.short_is && t <= P.dist1);
                        }
                    } else if (mapped && !mate_mapped) {
                        unmap = true;
                    }
                }
            }
            gf_taghit hit;
            hit.rec = rec; hit.gap = g;
            hit.kind = GF_KIND_CLIP; hit.to_mate = 0;
            emit_hit(clip, hit, hb, P.out, P.cap, P.n_out);
            hit.kind = pair ? GF_KIND_DISCORDANT : GF_KIND_UNMAP; hit.to_mate = 1;
            emit_hit(pair || unmap, hit, hb, P.out, P.cap, P.n_out);
            if (more) ++g;
        }
        tag_wave_sync();
        live_n = base;
    };
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t n_half = 2 * P.n;                       // 16-byte halves
    const uint64_t chunk = 64ull * TAG_UNROLL;             // halves per wave iteration
    const uint4* src = reinterpret_cast<const uint4*>(P.recs);
    // ---- per-wave queue of records that passed the LDS map and wait for the fine map (FINEQ)
    CandRec* wcand = candq[CQ ? threadIdx.x >> 6 : 0];
    uint32_t cand_n = 0;   // wave-uniform
    auto sift = [&]() {    // fine-map test of the last min(64, cand_n) queued records, one per lane
        const uint32_t base = cand_n > 64 ? cand_n - 64 : 0;
        bool live = base + lane < cand_n;
        CandRec c = {};
        if (live) {
            c = wcand[base + lane];
            if (FINEQ || P.fine_bits) {   // (KEYS without a fine map: the queue only gathers the records to fetch)
                const uint32_t f0 = P.fine_off[c.ref], fi = c.pos >> P.fine_shift, fb = f0 + fi;
                live = fi < P.fine_off[c.ref + 1] - f0 && ((P.fine_bits[fb >> 5] >> (fb & 31)) & 1u);
            }
        }
        tag_wave_sync();
        cand_n = base;
        const unsigned long long lb = __ballot(live);
        if (lb) {
            const uint32_t cnt = (uint32_t)__popcll(lb);
            if (live_n + cnt > LIVEQ) drain();             // (live_n < 64 on entry: one drain makes room)
            if (live) {
                const uint4 a = src[2 * (uint64_t)c.rec], b = src[2 * (uint64_t)c.rec + 1];
                LiveRec q;
                q.pos = a.x; q.ref = a.w; q.tlen = (int32_t)a.z; q.mate_ref = b.x; q.meta = b.y; q.rec = c.rec;
                wlive[live_n + __popcll(lb & ((1ull << lane) - 1))] = q;
            }
            live_n += cnt;
            tag_wave_sync();
            if (live_n >= 64) drain();
        }
    };
    if constexpr (KEYS) {
        // the key column: 16 bytes = two records per lane and load, TAG_UNROLL loads (1 024 records per wave) in flight behind the chunk at work
        const uint4* ksrc = reinterpret_cast<const uint4*>(P.keys);
        const uint64_t n_k16 = (P.n + 1) / 2;
        const uint64_t kchunk = 64ull * TAG_UNROLL;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* ksrc4 = reinterpret_cast<const u32x4*>(ksrc);
        // (plain loops, no lambda and no named pad value: anything whose address a closure takes goes to scratch memory — the first form of
        //  this loop kept its sixteen prefetched keys there, 144 bytes per lane, and ran no faster than the 32-byte stream)
        u32x4 kn[TAG_UNROLL];
#pragma unroll
        for (int u = 0; u < TAG_UNROLL; ++u) {
            const uint64_t h = wave * kchunk + 64ull * u + lane;
            kn[u] = h < n_k16 ? __builtin_nontemporal_load(ksrc4 + h) : u32x4{0u, 0x7FFFFFFFu, 0u, 0x7FFFFFFFu};
        }
        for (uint64_t h0 = wave * kchunk; h0 < n_k16; h0 += n_waves * kchunk) {
            u32x4 v[TAG_UNROLL];
#pragma unroll
            for (int u = 0; u < TAG_UNROLL; ++u) v[u] = kn[u];
#pragma unroll
            for (int u = 0; u < TAG_UNROLL; ++u) {
                const uint64_t h = h0 + n_waves * kchunk + 64ull * u + lane;
                kn[u] = h < n_k16 ? __builtin_nontemporal_load(ksrc4 + h) : u32x4{0u, 0x7FFFFFFFu, 0u, 0x7FFFFFFFu};
            }
#pragma unroll
            for (int u = 0; u < TAG_UNROLL; ++u) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const uint64_t rec = 2 * (h0 + 64ull * u + lane) + half;
                    const uint32_t pos = half ? v[u].z : v[u].x, w = half ? v[u].w : v[u].y;
                    const uint32_t ref = w & 0x7FFFFFFFu;
                    bool live = rec < P.n && ref < P.n_scaffolds;
                    if (P.low) {   // wave-uniform: compact the MAPQ==0 records for the second hop
                        const bool z = live && (w >> 31) && !(P.dbg & 1u);
                        const unsigned long long zb = __ballot(z);
                        if (zb) {
                            const uint32_t cnt = (uint32_t)__popcll(zb);
                            if (low_n + cnt > LOWBUF) flush_low();
                            if (z) wlow[low_n + __popcll(zb & ((1ull << lane) - 1))] = gf_lowrec{pos, ref, (uint32_t)rec};
                            low_n += cnt;
                        }
                    }
                    if (P.dbg & 2u) live = false;
                    if (live) {
                        const uint32_t b0 = boff[ref], nbin = boff[ref + 1] - b0, bi = pos >> P.bin_shift;
                        live = bi < nbin && ((bins[(b0 + bi) >> 5] >> ((b0 + bi) & 31)) & 1u);
                    }
                    if (P.dbg & 4u) live = false;
                    const unsigned long long cb = __ballot(live);
                    if (!cb) continue;
                    const uint32_t cnt = (uint32_t)__popcll(cb);
                    if (cand_n + cnt > CANDQ) sift();                  // (cand_n < 64 on entry: one sift empties the queue)
                    if (live) wcand[cand_n + __popcll(cb & ((1ull << lane) - 1))] = CandRec{pos, ref, (uint32_t)rec};
                    cand_n += cnt;
                    tag_wave_sync();
                    if (cand_n >= 64) sift();
                }
            }
        }
    } else {
    // software pipeline: the next chunk's loads are issued before the current chunk is processed
    uint4 vn[TAG_UNROLL];
    auto fetch = [&](uint64_t h0) {
#pragma unroll
        for (int u = 0; u < TAG_UNROLL; ++u) {
            const uint64_t h = h0 + 64ull * u + lane;
            if (NT) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 t = h < n_half ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src) + h) : u32x4{0, 0, 0, 0xFFFFFFFFu};
                vn[u] = make_uint4(t.x, t.y, t.z, t.w);
            } else
            vn[u] = h < n_half ? src[h] : make_uint4(0, 0, 0, 0xFFFFFFFFu);
        }
    };
    fetch(wave * chunk);
    for (uint64_t h0 = wave * chunk; h0 < n_half; h0 += n_waves * chunk) {
        uint4 v[TAG_UNROLL];
#pragma unroll
        for (int u = 0; u < TAG_UNROLL; ++u) v[u] = vn[u];
        fetch(h0 + n_waves * chunk);
        // Two 1-KiB loads hold 64 records as 128 halves, even lanes {pos, mate_pos, tlen, ref}, odd lanes {mate_ref, flag|mapq|clip, ..}:
        // lane L takes record L of the 64 — its halves sit in lanes 2L and 2L + 1 of the first load (L < 32) or of the second — so
        // that every step below works on 64 records, not on the 32 even lanes of one load (the kernel issues ~60 instructions per step
        // whatever the number of busy lanes: at 13 % of the records passing the map, the long-insert library, it was issue-bound).
        const uint32_t s0 = (lane << 1) & 63u, s1 = s0 + 1;
        const bool upper = lane >= 32;
#pragma unroll
        for (int u = 0; u < TAG_UNROLL; u += 2) {
            auto gather = [&](uint32_t a, uint32_t b, uint32_t sl) -> uint32_t {
                const uint32_t x = __shfl(a, sl), y = __shfl(b, sl);
                return upper ? y : x;
            };
            const uint64_t h = h0 + 64ull * u + 2ull * lane;      // first half of this lane's record
            gf_alnrec r = {};
            r.pos = gather(v[u].x, v[u + 1].x, s0);
            r.ref = gather(v[u].w, v[u + 1].w, s0);               // (halves beyond the input carry ref = 0xFFFFFFFF)
            const uint32_t meta = gather(v[u].y, v[u + 1].y, s1);
            bool live = h < n_half && r.ref < P.n_scaffolds;
            if (P.low) {   // wave-uniform: compact the MAPQ==0 records for the second hop
                const bool z = live && ((meta >> 16) & 0xFF) == 0;
                const unsigned long long zb = __ballot(z);
                if (zb) {
                    const uint32_t cnt = (uint32_t)__popcll(zb);
                    if (low_n + cnt > LOWBUF) flush_low();
                    if (z) wlow[low_n + __popcll(zb & ((1ull << lane) - 1))] = gf_lowrec{r.pos, r.ref, (uint32_t)(h >> 1)};
                    low_n += cnt;
                }
            }
            if (live) {  // coarse test first: almost every record lies far from every gap
                const uint32_t b0 = boff[r.ref], nbin = boff[r.ref + 1] - b0, bi = r.pos >> P.bin_shift;
                live = bi < nbin && ((bins[(b0 + bi) >> 5] >> ((b0 + bi) & 31)) & 1u);
            }
            if (FINEQ) {   // (launched with FINEQ only when the fine map exists)
                const unsigned long long cb = __ballot(live);
                if (!cb) continue;
                const uint32_t cnt = (uint32_t)__popcll(cb);
                if (cand_n + cnt > CANDQ) sift();                  // (cand_n < 64 on entry: one sift empties the queue)
                if (live) wcand[cand_n + __popcll(cb & ((1ull << lane) - 1))] = CandRec{r.pos, r.ref, (uint32_t)(h >> 1)};
                cand_n += cnt;
                tag_wave_sync();
                if (cand_n >= 64) sift();
                continue;
            }
            if (live && P.fine_bits) {
                const uint32_t f0 = P.fine_off[r.ref], fi = r.pos >> P.fine_shift, fb = f0 + fi;
                live = fi < P.fine_off[r.ref + 1] - f0 && ((P.fine_bits[fb >> 5] >> (fb & 31)) & 1u);
            }
            // survivors (a few per mille to a tenth) are queued; the window search runs on 64 of them at a time instead of once per
            // load with a few busy lanes
            const unsigned long long lb = __ballot(live);
            if (!lb) continue;
            const uint32_t tlen = gather(v[u].z, v[u + 1].z, s0), mate_ref = gather(v[u].x, v[u + 1].x, s1);
            const uint32_t cnt = (uint32_t)__popcll(lb);
            if (live_n + cnt > LIVEQ) drain();                     // (live_n < 64 on entry: one drain empties the queue)
            if (live) {
                LiveRec q;
                q.pos = r.pos; q.ref = r.ref; q.tlen = (int32_t)tlen; q.mate_ref = mate_ref; q.meta = meta; q.rec = (uint32_t)(h >> 1);
                wlive[live_n + __popcll(lb & ((1ull << lane) - 1))] = q;
            }
            live_n += cnt;
            tag_wave_sync();
            if (live_n >= 64) drain();
        }
    }
    }
    if (CQ) while (cand_n) sift();
    while (live_n) drain();
    if (P.low && low_n) flush_low();
    if (P.low && stage_n) flush_stage();
    if (hb.n) flush_wave_hits(hb, P.out, P.cap, P.n_out);
}

// the key column of a record array: {pos, ref | (mapq == 0) << 31}; an unplaced record ('*': ref 0xFFFFFFFF) keeps an invalid scaffold; entry
// n (the pad of an odd count: the tagger loads two keys at a time) is invalid too
__global__ __launch_bounds__(256) void alnrec_keys_kernel(const gf_alnrec* recs, uint64_t n, uint2* keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    uint2 k = make_uint2(0u, 0x7FFFFFFFu);
    if (i < n) {
        const uint4 a = reinterpret_cast<const uint4*>(recs)[2 * i], b = reinterpret_cast<const uint4*>(recs)[2 * i + 1];
        const uint32_t ref = a.w < 0x7FFFFFFFu ? a.w : 0x7FFFFFFFu, mapq = (b.y >> 16) & 0xFFu;
        k = make_uint2(a.x, ref | (mapq == 0 ? 0x80000000u : 0u));
    }
    keys[i] = k;
}

// (gf_internal.hpp: HOP_NEAR_LOG2, HOP_NEAR_SHIFT, hop_near_bit)
struct LowParams {
    const gf_alnrec* recs;     // full records ...
    uint64_t n;
    const gf_lowrec* low;      // ... or the compacted MAPQ==0 list with its device-side count
    const uint32_t* n_low;
    uint32_t low_cap;
    const uint32_t* upos;      // unique (scaffold-grouped) mate positions, ascending inside a scaffold
    const uint32_t* urow;      // n_unique+1 offsets into the row table
    const uint32_t* scaf_off;  // n_scaffolds+1 offsets into upos
    uint32_t n_scaffolds;
    gf_taghit* out;
    uint32_t cap;
    uint32_t* n_out;
    const uint32_t* near_bits; // optional (hop.hip): bit hop_near_bit(scaffold, pos >> 9) is set for every pos a table row lies near — see low_mapq_compact_kernel
};

// largest index uq in [first, hi) with upos[uq] <= lim, or `first - 1` when there is none: upos ascends inside a scaffold and the positions
// (mates of chimeric pairs) are spread evenly, so the search starts from the interpolated place and gallops — three or four loads instead
// of the seventeen of a bisection over 1.7e5 positions (the second hop was 0.55 ms of look-ups into an L2-resident array)
__device__ __forceinline__ uint32_t hop_upper(const uint32_t* upos, uint32_t first, uint32_t hi, uint64_t lim) {
    if (first >= hi) return first - 1;
    uint32_t lo = first;                       // invariant: everything below lo is <= lim, everything from hi on is > lim
    const uint32_t p0 = upos[first], p1 = upos[hi - 1];
    if ((uint64_t)p0 > lim) return first - 1;
    if ((uint64_t)p1 <= lim) return hi - 1;
    uint32_t g = first + (uint32_t)(((lim - p0) * (uint64_t)(hi - 1 - first)) / (uint64_t)(p1 - p0));   // p0 <= lim < p1
    if ((uint64_t)upos[g] <= lim) {
        lo = g + 1;
        for (uint32_t st = 1; lo < hi; st <<= 1) {         // gallop up
            const uint32_t t = lo + st - 1 < hi - 1 ? lo + st - 1 : hi - 1;
            if ((uint64_t)upos[t] <= lim) lo = t + 1; else { hi = t; break; }
        }
    } else {
        hi = g;
        for (uint32_t st = 1; lo < hi; st <<= 1) {         // gallop down
            const uint32_t t = hi > lo + st ? hi - st : lo;
            if ((uint64_t)upos[t] > lim) hi = t; else { lo = t + 1; break; }
        }
    }
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint64_t)upos[mid] <= lim) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}

__global__ __launch_bounds__(256) void low_mapq_kernel(LowParams P) {
    __shared__ gf_taghit whits[4][WHCAP];
    WaveHits hb{whits[threadIdx.x >> 6], 0};
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t n_half = 2 * P.n;
    const uint64_t chunk = 64ull * TAG_UNROLL;
    const uint4* src = reinterpret_cast<const uint4*>(P.recs);
    for (uint64_t h0 = wave * chunk; h0 < n_half; h0 += n_waves * chunk) {
        uint4 v[TAG_UNROLL];
#pragma unroll
        for (int u = 0; u < TAG_UNROLL; ++u) {
            const uint64_t h = h0 + 64ull * u + lane;
            v[u] = h < n_half ? src[h] : make_uint4(0, 0xFFFFFFFFu, 0, 0xFFFFFFFFu);
        }
#pragma unroll
        for (int u = 0; u < TAG_UNROLL; ++u) {
            const uint64_t h = h0 + 64ull * u + lane;
            const uint32_t nb_y = __shfl_down(v[u].y, 1);  // flag | mapq<<16 | clip<<24 of the even lane's record
            uint32_t row = 0, row_end = 0;
            if (!(lane & 1) && h < n_half) {
                const uint32_t pos = v[u].x, ref = v[u].w, mapq = (nb_y >> 16) & 0xFF;
                if (mapq == 0 && ref < P.n_scaffolds) {
                    const uint32_t first = P.scaf_off[ref];
                    const uint32_t uq = hop_upper(P.upos, first, P.scaf_off[ref + 1], (uint64_t)pos + 199);  // largest q <= pos+199
                    if (uq + 1 > first) {
                        if ((uint64_t)P.upos[uq] + 299 >= pos) {
                            row = P.urow[uq];
                            row_end = P.urow[uq + 1];
                        }
                    }
                }
            }
            while (true) {
                const bool more = row < row_end;
                if (!__any(more)) break;
                gf_taghit hit;
                hit.rec = (uint32_t)(h >> 1); hit.gap = row; hit.kind = GF_KIND_LOWMAPQ; hit.to_mate = 0;
                emit_hit(more, hit, hb, P.out, P.cap, P.n_out);
                if (more) ++row;
            }
        }
    }
    if (hb.n) flush_wave_hits(hb, P.out, P.cap, P.n_out);
}

// second hop over the compacted list (2 % of the records, 12 B each) instead of a second pass over every record
__global__ __launch_bounds__(256) void low_mapq_compact_kernel(LowParams P) {
    __shared__ gf_taghit whits[4][WHCAP];
    WaveHits hb{whits[threadIdx.x >> 6], 0};
    const uint32_t n = *P.n_low < P.low_cap ? *P.n_low : P.low_cap;
    const uint32_t n_round = (n + 63) & ~63u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += gridDim.x * blockDim.x) {
        uint32_t row = 0, row_end = 0, rec = 0;
        if (i < n) {
            const gf_lowrec e = P.low[i];
            rec = e.rec;
            // 99 % of the MAPQ-0 records lie near no row of the table: one look-up in a hashed bit map of the rows' neighbourhoods (built
            // with the table: 2^22 bits, a twelfth set at C4) spares them the search — a chain of eight dependent loads that set the
            // kernel's pace (0.57 ms for 18 M records).  A bit shared by chance only sends a record through the search it would have
            // made anyway.
            bool near = e.ref < P.n_scaffolds;
            if (near && P.near_bits) {
                const uint32_t b = hop_near_bit(e.ref, e.pos >> HOP_NEAR_SHIFT);
                near = (P.near_bits[b >> 5] >> (b & 31)) & 1u;
            }
            if (near) {
                const uint32_t first = P.scaf_off[e.ref];
                const uint32_t uq = hop_upper(P.upos, first, P.scaf_off[e.ref + 1], (uint64_t)e.pos + 199);  // largest q <= pos+199
                if (uq + 1 > first) {
                    if ((uint64_t)P.upos[uq] + 299 >= e.pos) {
                        row = P.urow[uq];
                        row_end = P.urow[uq + 1];
                    }
                }
            }
        }
        while (true) {
            const bool more = row < row_end;
            if (!__any(more)) break;
            gf_taghit hit;
            hit.rec = rec; hit.gap = row; hit.kind = GF_KIND_LOWMAPQ; hit.to_mate = 0;
            emit_hit(more, hit, hb, P.out, P.cap, P.n_out);
            if (more) ++row;
        }
    }
    if (hb.n) flush_wave_hits(hb, P.out, P.cap, P.n_out);
}

// static LDS of tag_kernel<16, ..., FINEQ = true>: hit buffers, MAPQ-0 buffers, the two queues (+ alignment slack)
constexpr size_t TAG16_STATIC_LDS = 16 * (WHCAP * sizeof(gf_taghit) + 64 * sizeof(gf_lowrec) + LIVEQ * sizeof(LiveRec) + CANDQ * sizeof(CandRec)) + 64;
static_assert(TAG16_STATIC_LDS + 64 * 1024 + 2049 * 4 <= 160 * 1024, "tag_kernel<16>: queues + a 64-KiB bin map + 2 049 scaffold offsets must fit a CU's LDS");

static unsigned stream_grid(gf_ctx* ctx, size_t n) {
    size_t blocks = (n + 255) / 256;
    return (unsigned)std::max<size_t>(1, std::min<size_t>(blocks, (size_t)ctx->n_cu * 8));
}

void drop_tag_maps(gf_ctx* ctx) {
    if (ctx->tag_maps.empty()) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (gf_ctx::TagMap& m : ctx->tag_maps) {
        if (m.map.p) (void)hipFree(m.map.p);
        if (m.fine.p) (void)hipFree(m.fine.p);
    }
    ctx->tag_maps.clear();
}

// Coarse bin map of the gap windows for one dist2 (host build: n_gaps work; kept until the gaps change, the four last used dist2).
// Bits cover, per scaffold, positions 0 .. last window end; everything beyond has no window.
static int ensure_bin_map(gf_ctx* ctx, int dist2, gf_ctx::TagMap** out) {
    for (gf_ctx::TagMap& m : ctx->tag_maps)
        if (m.dist2 == dist2 && m.map.p) { m.used = ++ctx->tag_map_clock; *out = &m; return GF_OK; }
    if (ctx->tag_maps.size() >= 4) {   // drop the one that was not used for the longest time
        size_t old = 0;
        for (size_t i = 1; i < ctx->tag_maps.size(); ++i) if (ctx->tag_maps[i].used < ctx->tag_maps[old].used) old = i;
        GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->tag_maps[old].map.p) (void)hipFree(ctx->tag_maps[old].map.p);
        if (ctx->tag_maps[old].fine.p) (void)hipFree(ctx->tag_maps[old].fine.p);
        ctx->tag_maps.erase(ctx->tag_maps.begin() + (long)old);
    }
    gf_ctx::TagMap M;
    const int64_t d2 = dist2 > 0 ? dist2 : 0;
    std::vector<uint64_t> span(ctx->n_scaffolds, 0);  // exclusive end of the covered positions per scaffold
    for (const gf_gap& g : ctx->gaps) span[g.scaffold] = std::max<uint64_t>(span[g.scaffold], (uint64_t)g.end + d2);
    auto nbits = [&](int sh) { uint64_t t = 0; for (uint64_t v : span) t += (v >> sh) + (v ? 1 : 0); return t; };
    // one map = bit words followed by the n_scaffolds+1 bit offsets
    auto build = [&](int shift, std::vector<uint32_t>& h, uint32_t& words) {
        std::vector<uint32_t> off(ctx->n_scaffolds + 1, 0);
        for (uint32_t s = 0; s < ctx->n_scaffolds; ++s) off[s + 1] = off[s] + (uint32_t)((span[s] >> shift) + (span[s] ? 1 : 0));
        words = (off[ctx->n_scaffolds] + 31) / 32 + 1;
        h.assign(words + off.size(), 0);
        for (const gf_gap& g : ctx->gaps) {
            if (d2 == 0) break;
            const int64_t lo = std::max<int64_t>(0, (int64_t)g.start - d2 + 1), hi = (int64_t)g.end + d2 - 1;
            for (int64_t b = lo >> shift; b <= (hi >> shift); ++b) {
                const uint32_t bit = off[g.scaffold] + (uint32_t)b;
                h[bit >> 5] |= 1u << (bit & 31);
            }
        }
        for (size_t i = 0; i < off.size(); ++i) h[words + i] = off[i];
    };
    int shift = 6;
    while (nbits(shift) > (1u << ctx->tag_bins_log2)) ++shift;  // <= 16 KiB of LDS by default (measured: an 8 KiB map sends more records down the slow path and loses)
    uint32_t words = 0;
    std::vector<uint32_t> h;
    build(shift, h, words);
    // Many gaps on a large genome (human scale: 19 840 windows over 3.1 Gb) leave a 16-KiB map with 24-kb bins, a fifth of them
    // set: then a 64-KiB map (6-kb bins, a twentieth set) shared by ONE 16-wave workgroup per CU takes its place — the same 16 waves
    // and 132 KiB of LDS per CU as four 4-wave workgroups with 16 KiB each (which is why a 64-KiB map per 4-wave workgroup lost:
    // half the waves).  Only when the option was left at its default.
    if (ctx->tag_bins_log2 == 17 && shift > 8) {
        uint64_t set0 = 0;
        for (uint32_t i = 0; i < words; ++i) set0 += (uint64_t)__builtin_popcount(h[i]);
        if (set0 * 10 > nbits(shift)) {
            while (shift > 6 && nbits(shift - 1) <= (1u << 19)) --shift;
            build(shift, h, words);
        }
    }
    uint64_t set_bits = 0;
    for (uint32_t i = 0; i < words; ++i) set_bits += (uint64_t)__builtin_popcount(h[i]);
    // Behind the offsets: per bin, the first gap of the scaffold whose RIGHT window still reaches the bin's first position
    // (end + dist2 > bin << shift; ends ascend) — where the window search of a record in that bin starts.  It replaces a binary
    // search over the scaffold's gaps (five dependent loads at 32 gaps per scaffold) by one load.
    {
        const size_t n_off = (size_t)ctx->n_scaffolds + 1;
        const uint32_t total_bits = h[words + ctx->n_scaffolds];
        std::vector<uint32_t> first(total_bits, 0);
        size_t g = 0;
        for (uint32_t s = 0; s < ctx->n_scaffolds; ++s) {
            while (g < ctx->gaps.size() && ctx->gaps[g].scaffold < s) ++g;
            const uint32_t b0 = h[words + s], nb = h[words + s + 1] - b0;
            for (uint32_t b = 0; b < nb; ++b) {
                while (g < ctx->gaps.size() && ctx->gaps[g].scaffold == s && (int64_t)ctx->gaps[g].end + d2 <= ((int64_t)b << shift)) ++g;
                first[b0 + b] = (uint32_t)g;   // (== the scaffold's last gap + 1 when no window reaches the bin)
            }
        }
        h.resize(words + n_off);
        h.insert(h.end(), first.begin(), first.end());
    }
    int rc = ensure(ctx, M.map, h.size() * 4);
    if (rc) return rc;
    GF_HIP(ctx, hipMemcpyAsync(M.map.p, h.data(), h.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // Human-scale genomes leave the LDS map with ~24 kb bins, a third of them set by 20 000 gaps: a second, finer map in
    // global memory (<= 2^25 bits, stays in L2) restores the per-mille pass rate before the window search.
    if (shift > 9 && set_bits * 20 > nbits(shift)) {   // worth a second look-up only when the LDS map passes > 5 %
        int fs = 7;
        while (nbits(fs) > (1u << ctx->tag_fine_log2)) ++fs;
        if (fs + 2 <= shift) {
            std::vector<uint32_t> hf;
            uint32_t fw = 0;
            build(fs, hf, fw);
            // ... and only when the finer bins stop a good part of what the LDS map lets through.  Long-insert libraries do not:
            // their windows (2 x dist2 + the gap, 15 kb at IS 5 000) are wider than the LDS map's bins, 13 % of the records pass the
            // LDS map and 10 % lie in a window — the second look-up and the re-load of the record it implies (13 % of the records a
            // second time, at random) cost more than the window search of the 3 % it would stop.
            uint64_t set_fine = 0;
            for (uint32_t i = 0; i < fw; ++i) set_fine += (uint64_t)__builtin_popcount(hf[i]);
            const double cover_lds = (double)set_bits * (double)(1ull << shift), cover_fine = (double)set_fine * (double)(1ull << fs);
            if (cover_fine <= 0.6 * cover_lds) {
                rc = ensure(ctx, M.fine, hf.size() * 4);
                if (rc) { (void)hipFree(M.map.p); return rc; }
                GF_HIP(ctx, hipMemcpyAsync(M.fine.p, hf.data(), hf.size() * 4, hipMemcpyHostToDevice, ctx->stream));
                GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
                M.fine_words = fw;
                M.fine_shift = fs;
            }
        }
    }
    M.dist2 = dist2;
    M.shift = shift;
    M.words = words;
    M.used = ++ctx->tag_map_clock;
    ctx->tag_maps.push_back(M);
    *out = &ctx->tag_maps.back();
    return GF_OK;
}

int launch_alnrec_keys(gf_ctx* ctx, const void* d_recs, size_t n, void* d_keys) {
    LaunchTimer tm(ctx, GF_KERNEL_INGEST);
    hipLaunchKernelGGL(alnrec_keys_kernel, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, ctx->stream, (const gf_alnrec*)d_recs, (uint64_t)n, (uint2*)d_keys);
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int launch_tag(gf_ctx* ctx, const void* d_recs, size_t n, int insert_size, int sd, int clip_dist, int anchor_mapq,
               void* d_out, size_t cap, void* d_n_out, void* d_low, size_t low_cap, void* d_n_low, const void* d_keys) {
    if (!ctx->d_gaps) return GF_E_STATE;
    if (n >= 0xFFFFFFFFull || cap > 0xFFFFFFFFull) return GF_E_INVAL;
    zero_regions(ctx, ZeroList{{(uint32_t*)d_n_out, (uint32_t*)d_n_low, nullptr, nullptr}, {1, d_n_low ? 1u : 0u, 0, 0}});
    if (n == 0) return GF_OK;
    TagParams P;
    P.low = (gf_lowrec*)d_low;
    P.low_cap = (uint32_t)std::min<size_t>(low_cap, 0xFFFFFFFFu);
    P.n_low = (uint32_t*)d_n_low;
    if (!d_low || !d_n_low) { P.low = nullptr; P.n_low = nullptr; }
    P.recs = (const gf_alnrec*)d_recs;
    P.keys = (const uint2*)d_keys;
    P.dbg = (uint32_t)ctx->tag_dbg;
    P.n = n;
    P.gaps = ctx->d_gaps;
    P.scaf_off = ctx->d_scaf_off;
    P.n_scaffolds = ctx->n_scaffolds;
    P.dist1 = insert_size - 3 * sd;
    P.dist2 = insert_size + 3 * sd;
    P.clip_dist = clip_dist;
    P.anchor_mapq = anchor_mapq;
    P.short_is = insert_size < 750;  // collect_reads_for_gaps.py:275
    P.out = (gf_taghit*)d_out;
    P.cap = (uint32_t)cap;
    P.n_out = (uint32_t*)d_n_out;
    gf_ctx::TagMap* M = nullptr;
    int rc = ensure_bin_map(ctx, P.dist2, &M);
    if (rc) return rc;
    P.bin_bits = (const uint32_t*)M->map.p;
    P.bin_off = P.bin_bits + M->words;
    P.bin_first = P.bin_off + ctx->n_scaffolds + 1;
    P.bin_shift = M->shift;
    P.bin_words = M->words;
    P.off_words = ctx->n_scaffolds + 1 <= 2048 ? ctx->n_scaffolds + 1 : 0;   // the per-scaffold offsets ride along in LDS when they are few
    size_t lds_map = ((size_t)M->words + P.off_words) * 4;
    // the 16-wave form keeps TAG16_STATIC_LDS bytes of queues and buffers beside the staged map: when the two do not fit a CU's LDS
    // the per-scaffold offsets stay in global memory
    if ((size_t)M->words * 4 > 32 * 1024 && TAG16_STATIC_LDS + lds_map > 160 * 1024) {
        P.off_words = 0;
        lds_map = (size_t)M->words * 4;
        if (TAG16_STATIC_LDS + lds_map > 160 * 1024) return GF_E_UNSUPPORTED;   // (a 64-KiB map is the largest ensure_bin_map builds)
    }
    P.fine_bits = M->fine_words ? (const uint32_t*)M->fine.p : nullptr;
    P.fine_off = P.fine_bits ? P.fine_bits + M->fine_words : nullptr;
    P.fine_shift = M->fine_shift;
    const size_t map_bytes = (size_t)M->words * 4;
    // workgroups x waves of the variant that runs (the staging buffer of the MAPQ-0 by-product has a slice per wave)
    const bool big_map = map_bytes > 32 * 1024;
    const unsigned nw = ctx->tag_light ? 1u : big_map ? 16u : 4u;
    const unsigned grid = ctx->tag_light ? 4 * stream_grid(ctx, n)
                          : big_map ? (unsigned)std::max<size_t>(1, std::min<size_t>((n + 1023) / 1024, (size_t)ctx->n_cu)) : stream_grid(ctx, n);
    P.low_stage = nullptr;
    if (P.low) {
        if ((rc = ensure(ctx, ctx->tag_stage, (size_t)grid * nw * LOWSTAGE * sizeof(gf_lowrec)))) return rc;
        P.low_stage = (gf_lowrec*)ctx->tag_stage.p;
    }
    {
        LaunchTimer tm(ctx, GF_KERNEL_TAG);
        if (d_keys && !ctx->tag_light) {   // the key column as the stream (gf_tag_alignments_keys_dev)
            if (big_map) hipLaunchKernelGGL((tag_kernel<16, true, true, false, true>), dim3(grid), dim3(1024), lds_map, ctx->stream, P);
            else hipLaunchKernelGGL((tag_kernel<4, true, true, false, true>), dim3(grid), dim3(256), lds_map, ctx->stream, P);
        } else
        if (ctx->tag_light)   // one-wave workgroups without the LDS bin map: co-resident with the k-mer filter's workgroups
            hipLaunchKernelGGL((tag_kernel<1, false>), dim3(grid), dim3(64), 0, ctx->stream, P);
        else if (big_map && P.fine_bits)   // the 64-KiB map: one 16-wave workgroup per CU
            hipLaunchKernelGGL((tag_kernel<16, true, true, true>), dim3(grid), dim3(1024), lds_map, ctx->stream, P);
        else if (big_map)
            hipLaunchKernelGGL((tag_kernel<16, true, true, false>), dim3(grid), dim3(1024), lds_map, ctx->stream, P);
        else if (ctx->tag_nt && P.fine_bits)
            hipLaunchKernelGGL((tag_kernel<4, true, true, true>), dim3(grid), dim3(256), lds_map, ctx->stream, P);
        else if (ctx->tag_nt)
            hipLaunchKernelGGL((tag_kernel<4, true, true>), dim3(grid), dim3(256), lds_map, ctx->stream, P);
        else
            hipLaunchKernelGGL((tag_kernel<4, true>), dim3(grid), dim3(256), lds_map, ctx->stream, P);
    }
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int launch_low_mapq(gf_ctx* ctx, const void* d_recs, size_t n, const gf_dpos* table, size_t n_rows, void* d_out,
                    size_t cap, void* d_n_out, const void* d_low, const void* d_n_low, size_t low_cap) {
    if (n >= 0xFFFFFFFFull || cap > 0xFFFFFFFFull || n_rows >= 0xFFFFFFFFull || low_cap > 0xFFFFFFFFull) return GF_E_INVAL;
    if (ctx->n_scaffolds == 0) return GF_E_STATE;
    GF_HIP(ctx, hipMemsetAsync(d_n_out, 0, 4, ctx->stream));
    if ((n == 0 && !d_low) || n_rows == 0) return GF_OK;
    // host: unique positions per scaffold + row offsets (the table is tiny next to the record stream); the device copy
    // is cached and re-used while the caller passes the same rows (a pipeline calls this once per batch of records)
    const bool cached = ctx->low_rows.size() == n_rows * 4 && ctx->table.p &&
                        memcmp(ctx->low_rows.data(), table, n_rows * sizeof(gf_dpos)) == 0;
    if (!cached) {
        std::vector<uint32_t> upos, urow, soff(ctx->n_scaffolds + 1, 0);
        uint32_t cur_s = 0;
        for (size_t r = 0; r < n_rows; ++r) {
            if (r) {
                const gf_dpos &a = table[r - 1], &b = table[r];
                if (a.mate_scaffold > b.mate_scaffold || (a.mate_scaffold == b.mate_scaffold && a.mate_pos > b.mate_pos))
                    return GF_E_INVAL;  // must be sorted (run_multi_threads_discordant.py:103)
            }
            if (table[r].mate_scaffold >= ctx->n_scaffolds) return GF_E_INVAL;
            if (r == 0 || table[r].mate_scaffold != table[r - 1].mate_scaffold || table[r].mate_pos != table[r - 1].mate_pos) {
                while (cur_s < table[r].mate_scaffold) soff[++cur_s] = (uint32_t)upos.size();
                upos.push_back(table[r].mate_pos);
                urow.push_back((uint32_t)r);
            }
        }
        urow.push_back((uint32_t)n_rows);
        while (cur_s < ctx->n_scaffolds) soff[++cur_s] = (uint32_t)upos.size();
        const size_t b1 = upos.size() * 4, b2 = urow.size() * 4, b3 = soff.size() * 4;
        int rc;
        if ((rc = ensure(ctx, ctx->table, b1 + b2 + b3 + 64))) return rc;
        uint8_t* base = (uint8_t*)ctx->table.p;
        GF_HIP(ctx, hipMemcpyAsync(base, upos.data(), b1, hipMemcpyHostToDevice, ctx->stream));
        GF_HIP(ctx, hipMemcpyAsync(base + b1, urow.data(), b2, hipMemcpyHostToDevice, ctx->stream));
        GF_HIP(ctx, hipMemcpyAsync(base + b1 + b2, soff.data(), b3, hipMemcpyHostToDevice, ctx->stream));
        GF_HIP(ctx, hipStreamSynchronize(ctx->stream));  // host vectors go out of scope
        ctx->low_rows.assign((const uint32_t*)table, (const uint32_t*)table + n_rows * 4);
        ctx->low_b1 = b1;
        ctx->low_b2 = b2;
    }
    const size_t b1 = ctx->low_b1, b2 = ctx->low_b2;
    uint8_t* base = (uint8_t*)ctx->table.p;
    LowParams P;
    P.recs = (const gf_alnrec*)d_recs;
    P.n = n;
    P.low = (const gf_lowrec*)d_low;
    P.n_low = (const uint32_t*)d_n_low;
    P.low_cap = (uint32_t)low_cap;
    P.upos = (const uint32_t*)base;
    P.urow = (const uint32_t*)(base + b1);
    P.scaf_off = (const uint32_t*)(base + b1 + b2);
    P.n_scaffolds = ctx->n_scaffolds;
    P.out = (gf_taghit*)d_out;
    P.cap = (uint32_t)cap;
    P.near_bits = nullptr;
    P.n_out = (uint32_t*)d_n_out;
    {
        LaunchTimer tm(ctx, GF_KERNEL_LOWMAPQ);
        if (d_low)
            hipLaunchKernelGGL(low_mapq_compact_kernel, dim3(stream_grid(ctx, std::max<size_t>(low_cap, 1))), dim3(256), 0, ctx->stream, P);
        else
            hipLaunchKernelGGL(low_mapq_kernel, dim3(stream_grid(ctx, n)), dim3(256), 0, ctx->stream, P);
    }
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

// second hop over the compacted MAPQ-0 list with look-up arrays that already live on the device (hop.hip builds them)
int launch_low_mapq_devtable(gf_ctx* ctx, const void* d_low, const void* d_n_low, size_t low_cap, const uint32_t* upos, const uint32_t* urow,
                             const uint32_t* soff, const uint32_t* near_bits, void* d_out, size_t cap, void* d_n_out) {
    LowParams P;   // *d_n_out was zeroed by the kernel that built the look-up arrays
    P.recs = nullptr;
    P.n = 0;
    P.low = (const gf_lowrec*)d_low;
    P.n_low = (const uint32_t*)d_n_low;
    P.low_cap = (uint32_t)low_cap;
    P.upos = upos;
    P.urow = urow;
    P.scaf_off = soff;
    P.n_scaffolds = ctx->n_scaffolds;
    P.out = (gf_taghit*)d_out;
    P.cap = (uint32_t)cap;
    P.n_out = (uint32_t*)d_n_out;
    P.near_bits = near_bits;
    {
        LaunchTimer tm(ctx, GF_KERNEL_LOWMAPQ);
        hipLaunchKernelGGL(low_mapq_compact_kernel, dim3(stream_grid(ctx, std::max<size_t>(low_cap, 1))), dim3(256), 0, ctx->stream, P);
    }
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

}  // namespace gf
