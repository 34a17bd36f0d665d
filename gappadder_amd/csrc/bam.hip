// BAM ingest on the device ("BGZF-BAM -> 32 B records").  The reference never opens a BAM itself: it pipes
// `samtools view` text into collect_reads_for_gaps.py:76-91 / collect_discordant_low_mapq_reads.py:40-52, ~350 B of text per
// record and a second full pass for the MAPQ-0 hop.  Here the BGZF file bytes go to HBM as they are:
//   1. bgzf_inflate_kernel — one wavefront per BGZF block (<= 64 KiB of output, RFC 1951 DEFLATE): the wave shares one bit
//      buffer (lane i of a register holds dword i of a 256-B window of the compressed stream, `v_readlane` feeds the buffer),
//      Huffman codes are resolved by ONE wave-wide compare (lane l holds the left-aligned upper limit of the length-l codes:
//      ballot + ctz gives the code length, one LDS read the symbol), matches are copied by all lanes at once, and the block's
//      CRC-32 is checked in 64 slices recombined with GF(2) shifts — a corrupt block is an error, never silent.
//   2. the BAM record chain (each record starts where the previous one's block_size says) is followed speculatively: one lane
//      per 64-KiB segment guesses the first record of its segment by validating candidate offsets, walks to the segment end,
//      and the exits are then compared with the next segment's guess; segment 0 starts at the known first record, so once every
//      exit equals the next guess the whole chain is the true one (exact, not heuristic); mismatching segments are re-walked
//      from the exit of their predecessor until none is left.
//   3. bam_emit_kernel decodes the fixed-offset fields to gf_alnrec exactly as the SAM text path does (ingest.hip):
//      POS/PNEXT = pos+1, clipflag from the first/last CIGAR operation (GapReadsCollector.is_clipped :13-26), RNAME/RNEXT
//      through the caller's refID -> .fai index map.
#include <algorithm>
#include <cstring>
#include <vector>

#include "gf_internal.hpp"

namespace gf {

struct BgzfBlock {
    uint64_t in_off;   // first byte of the DEFLATE data in the staged file bytes
    uint64_t out_off;  // where the block's bytes go in the inflated stream
    uint32_t clen, isize, crc, pad;
};

enum : uint32_t {
    INF_OK = 0,
    INF_BAD_BTYPE = 1, INF_BAD_STORED = 2, INF_BAD_LENGTHS = 3, INF_BAD_CODE = 4, INF_BAD_DIST = 5, INF_OVERRUN = 6,
    INF_SHORT = 7, INF_TRAILING = 8, INF_CRC = 9,
};

__device__ const uint32_t INF_LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint32_t INF_LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint32_t INF_DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint32_t INF_DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint32_t INF_CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ void wave_lds_sync() {   // LDS hand-off between lanes of one wave
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// A byte another lane of this wave stored earlier (after an s_waitcnt vmcnt(0)): read through L2, never from a stale L1 line.
__device__ __forceinline__ uint32_t ld_word_l2(const uint8_t* p4) {
    return __hip_atomic_load(reinterpret_cast<const uint32_t*>(p4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_byte_l2(const uint8_t* p) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    return (ld_word_l2(reinterpret_cast<const uint8_t*>(a & ~(uintptr_t)3)) >> ((a & 3) * 8)) & 0xFFu;
}

// residues modulo the CRC-32 polynomial in the reflected bit order of the CRC register (bit 31 = x^0)
__device__ __forceinline__ uint32_t gf2_mul(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 31; i >= 0; --i) {
        if ((a >> i) & 1u) p ^= b;
        b = (b >> 1) ^ ((b & 1u) ? 0xEDB88320u : 0u);
    }
    return p;
}
__device__ __forceinline__ uint32_t gf2_xpow_bytes(uint32_t n_bytes) {   // x^(8 n)
    uint32_t r = 0x80000000u, sq = 0x00800000u;   // 1, x^8
    while (n_bytes) {
        if (n_bytes & 1u) r = gf2_mul(r, sq);
        sq = gf2_mul(sq, sq);
        n_bytes >>= 1;
    }
    return r;
}

constexpr int INF_WAVES = 4;

__global__ __launch_bounds__(64 * INF_WAVES) void bgzf_inflate_kernel(const uint8_t* in, const BgzfBlock* blocks, uint32_t n_blocks,
                                                                      uint8_t* out, uint32_t* status) {
    __shared__ uint32_t crc_tab[256];
    __shared__ uint16_t s_ll[INF_WAVES][288], s_d[INF_WAVES][32], s_cl[INF_WAVES][20];
    __shared__ uint8_t s_len[INF_WAVES][320 + 64];
    // everything below is per WAVE: the wave index goes through readfirstlane so that the compiler keeps the decoder state in
    // scalar registers and branches on it with scalar branches (derived from threadIdx it would count as divergent)
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        uint32_t c = tid;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? 0xEDB88320u : 0u);
        crc_tab[tid] = c;
    }
    __syncthreads();
    const uint32_t b = blockIdx.x * INF_WAVES + w;
    if (b >= n_blocks) return;
    const BgzfBlock blk = blocks[b];
    uint16_t* SLL = s_ll[w];
    uint16_t* SD = s_d[w];
    uint16_t* SCL = s_cl[w];
    uint8_t* LEN = s_len[w];
    uint8_t* dst = out + blk.out_off;
    const uint32_t isize = blk.isize;

    // ---- shared bit reader (everything here is wave-uniform except cur)
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(in + blk.in_off);
    const uint32_t* wp;
    {   // scalar base for the window loads (readfirstlane returns int: rebuild the pointer from unsigned halves)
        const uint64_t v = a0 & ~(uint64_t)3;
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)), lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        wp = reinterpret_cast<const uint32_t*>(((uint64_t)hi << 32) | lo);
    }
    uint32_t widx = 0, cnt = 0, used = 0;   // next dword to take; bits in buf; bits consumed since wp
    uint64_t buf = 0;
    uint32_t cur;                            // lane i: dword (widx & ~63) + i of the stream
    // the window load carries its own wait: left to the compiler, the pending load is tracked across the decode loop and every
    // refill gets an s_waitcnt vmcnt(0) that also waits for all literal stores in flight (measured: 2x slower)
    auto load_window = [&](uint32_t first_dword) {
        const uint32_t off = (first_dword + lane) * 4;
        // s_nop: the scalar base may have just been written by a v_readlane (SGPR spill reload), and the compiler does not
        // insert the VALU-writes-SGPR -> VMEM wait states in front of inline asm
        asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(cur) : "v"(off), "s"(wp) : "memory");
    };
    load_window(0);
    auto refill = [&]() {
        while (cnt <= 32) {
            const uint32_t v = __builtin_amdgcn_readlane(cur, widx & 63);
            buf |= (uint64_t)v << cnt;
            cnt += 32;
            ++widx;
            if ((widx & 63) == 0) load_window(widx);
        }
    };
    auto drop = [&](uint32_t n) { buf >>= n; cnt -= n; used += n; };
    auto take = [&](uint32_t n) -> uint32_t {   // n <= 16, bits already in buf
        const uint32_t v = (uint32_t)buf & ((1u << n) - 1u);
        drop(n);
        return v;
    };
    auto seek_bit = [&](uint32_t bit) {   // reposition at absolute bit offset from wp
        widx = bit >> 5;
        load_window(widx & ~63u);
        buf = 0; cnt = 0; used = bit & ~31u;
        refill();
        drop(bit & 31u);
    };
    // ---- canonical Huffman decoder state: lane l (1..15) keeps the limit and the symbol-table base of the length-l codes
    auto build = [&](const uint8_t* L, uint32_t n, uint16_t* S, uint32_t& lim, uint32_t& bas) -> bool {
        uint32_t cntv = 0;
        for (uint32_t j = 0; j < n; j += 64) {
            const uint32_t my = j + lane < n ? L[j + lane] : 0u;
            for (uint32_t l = 1; l <= 15; ++l) {
                const uint32_t c = (uint32_t)__popcll(__ballot(my == l));
                if (lane == l) cntv += c;
            }
        }
        uint32_t code = 0, off = 0, nextv = 0;
        int left = 1;
        bool ok = true;
        lim = 0; bas = 0;
        for (uint32_t l = 1; l <= 15; ++l) {
            const uint32_t c = __builtin_amdgcn_readlane(cntv, l);
            left = (left << 1) - (int)c;
            if (left < 0) ok = false;   // over-subscribed
            if (lane == l) { lim = (code + c) << (15 - l); bas = off - code; nextv = off; }
            code = (code + c) << 1;
            off += c;
        }
        for (uint32_t j = 0; j < n; j += 64) {
            const uint32_t my = j + lane < n ? L[j + lane] : 0u;
            for (uint32_t l = 1; l <= 15; ++l) {
                const unsigned long long m = __ballot(my == l);
                if (!m) continue;
                const uint32_t at = __builtin_amdgcn_readlane(nextv, l);
                if (my == l) S[at + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)(j + lane);
                if (lane == l) nextv += (uint32_t)__popcll(m);
            }
        }
        wave_lds_sync();
        return ok;
    };
    uint32_t err = INF_OK;
    auto decode = [&](const uint16_t* S, uint32_t lim, uint32_t bas) -> uint32_t {   // <= 15 bits in buf
        const uint32_t code15 = __brev((uint32_t)buf) >> 17;
        const unsigned long long m = __ballot(code15 < lim);
        if (!m) { err = INF_BAD_CODE; return 0; }
        const uint32_t len = (uint32_t)__builtin_ctzll(m);
        const uint32_t idx = (uint32_t)__builtin_amdgcn_readlane(bas, len) + (code15 >> (15 - len));
        drop(len);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)S[idx]);
    };

    refill();
    drop((uint32_t)(a0 & 3) * 8);
    uint32_t pos = 0, synced = 0;   // bytes produced; bytes known to have reached L2
    bool last = false;
    while (!last && !err) {
        refill();
        last = take(1);
        const uint32_t btype = take(2);
        if (btype == 3) { err = INF_BAD_BTYPE; break; }
        if (btype == 0) {   // stored: LEN, ~LEN, bytes
            drop((0u - used) & 7u);
            refill();
            const uint32_t len = take(16);
            refill();
            const uint32_t nlen = take(16);
            if ((len ^ nlen) != 0xFFFFu) { err = INF_BAD_STORED; break; }
            if (pos + len > isize) { err = INF_OVERRUN; break; }
            // the stored bytes must lie inside this block's compressed data (a corrupt LEN would read past the staged chunk)
            if ((used >> 3) + len > blk.clen + (uint32_t)(a0 & 3)) { err = INF_OVERRUN; break; }
            const uint8_t* src = reinterpret_cast<const uint8_t*>(wp) + (used >> 3);
            for (uint32_t i = lane; i < len; i += 64) dst[pos + i] = src[i];
            pos += len;
            seek_bit(used + len * 8);
            continue;
        }
        uint32_t ll_lim, ll_bas, d_lim, d_bas;
        if (btype == 1) {   // fixed code
            for (uint32_t s = lane; s < 288; s += 64) LEN[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < 32) LEN[288 + lane] = 5;
            wave_lds_sync();
            build(LEN, 288, SLL, ll_lim, ll_bas);
            build(LEN + 288, 30, SD, d_lim, d_bas);
        } else {            // dynamic code
            refill();
            const uint32_t hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
            if (hlit > 286 || hdist > 30) { err = INF_BAD_LENGTHS; break; }
            if (lane < 19) LEN[lane] = 0;
            wave_lds_sync();
            for (uint32_t i = 0; i < hclen; ++i) {
                refill();
                const uint32_t v = take(3);
                if (lane == 0) LEN[INF_CLORDER[i]] = (uint8_t)v;
            }
            wave_lds_sync();
            uint32_t cl_lim, cl_bas;
            if (!build(LEN, 19, SCL, cl_lim, cl_bas)) { err = INF_BAD_LENGTHS; break; }
            uint8_t* LL = LEN + 32;   // hlit + hdist code lengths, decoded one after the other
            uint32_t i = 0, prev = 0;
            while (i < hlit + hdist && !err) {
                refill();
                const uint32_t sym = decode(SCL, cl_lim, cl_bas);
                if (err) break;
                if (sym < 16) {
                    if (lane == 0) LL[i] = (uint8_t)sym;
                    prev = sym;
                    ++i;
                } else {
                    uint32_t rep, val = 0;
                    if (sym == 16) {
                        if (i == 0) { err = INF_BAD_LENGTHS; break; }
                        val = prev;
                        rep = 3 + take(2);
                    } else if (sym == 17) rep = 3 + take(3);
                    else rep = 11 + take(7);
                    if (i + rep > hlit + hdist) { err = INF_BAD_LENGTHS; break; }
                    if (lane < rep) LL[i + lane] = (uint8_t)val;
                    if (lane + 64 < rep) LL[i + 64 + lane] = (uint8_t)val;
                    if (lane + 128 < rep) LL[i + 128 + lane] = (uint8_t)val;
                    prev = val;
                    i += rep;
                }
            }
            if (err) break;
            wave_lds_sync();
            if (__builtin_amdgcn_readfirstlane((int)LL[256]) == 0) { err = INF_BAD_LENGTHS; break; }   // no end-of-block code
            if (!build(LL, hlit, SLL, ll_lim, ll_bas) || !build(LL + hlit, hdist, SD, d_lim, d_bas)) { err = INF_BAD_LENGTHS; break; }
        }
        // ---- symbols of this block
        while (!err) {
            refill();
            // a corrupt block must not decode on into the next blocks: stop once the bit position is past this block's data
            // (+4 bytes of slack for the bits held in the buffer); the exact end is checked below (INF_TRAILING)
            if (used > 8 * (blk.clen + 4 + (uint32_t)(a0 & 3))) { err = INF_OVERRUN; break; }
            uint32_t sym = decode(SLL, ll_lim, ll_bas);
            if (err) break;
            if (sym < 256) {
                if (pos >= isize) { err = INF_OVERRUN; break; }
                if (lane == 0) dst[pos] = (uint8_t)sym;
                ++pos;
                continue;
            }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) { err = INF_BAD_CODE; break; }
            const uint32_t len = INF_LBASE[sym] + take(INF_LEXT[sym]);
            refill();
            const uint32_t ds = decode(SD, d_lim, d_bas);
            if (err) break;
            if (ds >= 30) { err = INF_BAD_DIST; break; }
            const uint32_t dist = INF_DBASE[ds] + take(INF_DEXT[ds]);
            if (dist > pos) { err = INF_BAD_DIST; break; }
            if (pos + len > isize) { err = INF_OVERRUN; break; }
            const uint32_t run = dist < len ? dist : len;   // distinct source bytes
            if (pos - dist + run > synced) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                synced = pos;
            }
            const uint8_t* src = dst + pos - dist;
            if (dist >= len) {
                for (uint32_t i = lane; i < len; i += 64) dst[pos + i] = (uint8_t)ld_byte_l2(src + i);
            } else {   // overlapping match: the last `dist` bytes repeat
                for (uint32_t i = lane; i < len; i += 64) dst[pos + i] = (uint8_t)ld_byte_l2(src + i % dist);
            }
            pos += len;
        }
    }
    if (!err && pos != isize) err = INF_SHORT;
    if (!err && ((used + 7) >> 3) - (uint32_t)(a0 & 3) != blk.clen) err = INF_TRAILING;
    if (!err) {   // CRC-32 of the block in 64 slices: crc(A|B) = crc(A) * x^(8|B|) + crc(B)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t per = (isize + 63) / 64;
        const uint32_t s0 = lane * per < isize ? lane * per : isize, s1 = s0 + per < isize ? s0 + per : isize;
        uint32_t c = 0;
        if (s0 < s1) {
            c = 0xFFFFFFFFu;
            const uintptr_t pa = reinterpret_cast<uintptr_t>(dst + s0), pb = reinterpret_cast<uintptr_t>(dst + s1);
            for (uintptr_t q = pa & ~(uintptr_t)3; q < pb; q += 4) {
                const uint32_t wv = ld_word_l2(reinterpret_cast<const uint8_t*>(q));
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (q + k >= pa && q + k < pb) c = crc_tab[(c ^ (wv >> (8 * k))) & 0xFFu] ^ (c >> 8);
            }
            c ^= 0xFFFFFFFFu;
            c = gf2_mul(c, gf2_xpow_bytes(isize - s1));
        }
        for (int d = 32; d >= 1; d >>= 1) c ^= __shfl_xor(c, d);
        if (c != blk.crc) err = INF_CRC;
    }
    if (lane == 0) status[b] = err;
}

// ------------------------------------------------------------------------------------------------ BAM record chain
constexpr uint64_t BAM_SEG = 65536;
constexpr uint64_t BAM_TAIL = 1ull << 63;   // exit flag: the chain ended in an incomplete record at that offset

struct BamStream {
    const uint8_t* p;
    uint64_t n, first;
    int32_t n_ref;
};

__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint32_t ld16u(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// does a plausible alignment record start at o?  (fields per SAMv1 §4.2; only used to GUESS — the chain check decides)
__device__ bool bam_plausible(const BamStream& B, uint64_t o) {
    if (o + 36 > B.n) return false;
    const uint8_t* r = B.p + o;
    const int64_t bs = (int32_t)ld32u(r);
    const int32_t ref = (int32_t)ld32u(r + 4), pos = (int32_t)ld32u(r + 8), mref = (int32_t)ld32u(r + 24), mpos = (int32_t)ld32u(r + 28);
    const uint32_t l_name = r[12], n_cig = ld16u(r + 16);
    const int64_t l_seq = (int32_t)ld32u(r + 20);
    if (bs < 32 || bs > (1 << 28) || l_seq < 0 || l_name < 1) return false;
    if (ref < -1 || ref >= B.n_ref || mref < -1 || mref >= B.n_ref || pos < -1 || mpos < -1) return false;
    if (bs < 32 + (int64_t)l_name + 4 * (int64_t)n_cig + l_seq + (l_seq + 1) / 2) return false;
    if (o + 36 + l_name <= B.n) {
        if (r[36 + l_name - 1] != 0) return false;
        for (uint32_t i = 0; i + 1 < l_name; ++i)
            if (r[36 + i] < 0x21 || r[36 + i] > 0x7E) return false;
    }
    return true;
}

// walk from `entry` to the first record start >= seg_end; counts the complete records that start inside [.., seg_end)
template <bool EMIT>
__device__ uint64_t bam_walk(const BamStream& B, uint64_t entry, uint64_t seg_end, uint32_t& count,
                             const uint32_t* ref_map, gf_alnrec* recs, unsigned long long* rec_begin, uint64_t idx, uint64_t cap) {
    count = 0;
    if (entry & BAM_TAIL) return entry;
    uint64_t o = entry;
    while (o < seg_end) {
        if (o + 4 > B.n) return o | BAM_TAIL;
        const int64_t bs = (int32_t)ld32u(B.p + o);
        if (bs < 32) return o | BAM_TAIL;   // cannot be a record: a wrong guess, or (on the true chain) a corrupt file — the host looks
        if (o + 4 + (uint64_t)bs > B.n) return o | BAM_TAIL;
        if (EMIT && idx < cap) {
            const uint8_t* r = B.p + o;
            const int32_t ref = (int32_t)ld32u(r + 4), mref = (int32_t)ld32u(r + 24);
            const uint32_t l_name = r[12], n_cig = ld16u(r + 16);
            gf_alnrec a;
            a.pos = ld32u(r + 8) + 1u;
            a.mate_pos = ld32u(r + 28) + 1u;
            a.tlen = (int32_t)ld32u(r + 32);
            a.ref = ref >= 0 && ref < B.n_ref ? ref_map[ref] : 0xFFFFFFFFu;
            a.mate_ref = mref >= 0 && mref < B.n_ref ? ref_map[mref] : 0xFFFFFFFFu;
            a.flag = (uint16_t)ld16u(r + 18);
            a.mapq = r[13];
            uint32_t cf = 0;
            if (n_cig && 36 + (uint64_t)l_name + 4ull * n_cig <= (uint64_t)bs + 4) {
                const uint8_t* cg = r + 36 + l_name;
                const uint32_t op0 = ld32u(cg) & 0xFu, op1 = ld32u(cg + 4 * (n_cig - 1)) & 0xFu;   // 4 = S, 5 = H
                cf = ((op1 == 4 || op1 == 5) ? 2u : 0u) + ((op0 == 4 || op0 == 5) ? 1u : 0u);
            }
            a.clipflag = (uint8_t)cf;
            a.read = idx;
            recs[idx] = a;
            if (rec_begin) rec_begin[idx] = o;
        }
        ++idx;
        ++count;
        o += 4 + (uint64_t)bs;
    }
    return o;
}

__device__ __forceinline__ uint64_t bam_seg_start(const BamStream& B, uint64_t s) { return B.first + s * BAM_SEG; }
__device__ __forceinline__ uint64_t bam_seg_end(const BamStream& B, uint64_t s) {
    const uint64_t e = B.first + (s + 1) * BAM_SEG;
    return e < B.n ? e : B.n;
}

// guess[s] for s >= 1: the first offset in the segment from which three records in a row look plausible (or the segment
// end when there is none); guess[0] = first
__global__ __launch_bounds__(256) void bam_guess_kernel(BamStream B, uint64_t n_seg, unsigned long long* guess) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    if (s == 0) { guess[0] = B.first; return; }
    const uint64_t a = bam_seg_start(B, s), e = bam_seg_end(B, s);
    uint64_t g = e;
    for (uint64_t o = a; o < e; ++o) {
        if (!bam_plausible(B, o)) continue;
        uint64_t q = o;
        bool ok = true;
        for (int d = 0; d < 2 && ok; ++d) {
            q += 4 + (uint64_t)ld32u(B.p + q);
            if (q + 36 > B.n) break;   // runs off the data: nothing left to contradict it
            ok = bam_plausible(B, q);
        }
        if (ok) { g = o; break; }
    }
    guess[s] = g;
}

// exit[s] and count[s] from guess[s]; only segments flagged in `todo` (null: all).  A segment whose guess lies beyond its
// end (a record spans it entirely) passes the guess on.
__global__ __launch_bounds__(256) void bam_walk_kernel(BamStream B, uint64_t n_seg, const unsigned long long* guess, unsigned long long* exit_,
                                                       uint32_t* count, const uint8_t* todo) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg || (todo && !todo[s])) return;
    uint32_t c = 0;
    exit_[s] = bam_walk<false>(B, guess[s], bam_seg_end(B, s), c, nullptr, nullptr, nullptr, 0, 0);
    count[s] = c;
}

// compare every exit with the next segment's guess; a mismatching segment takes its predecessor's exit as its new guess
__global__ __launch_bounds__(256) void bam_check_kernel(uint64_t n_seg, unsigned long long* guess, const unsigned long long* exit_, uint8_t* todo,
                                                        uint32_t* n_mismatch) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    bool redo = false;
    if (s > 0 && guess[s] != exit_[s - 1]) {
        guess[s] = exit_[s - 1];
        redo = true;
    }
    todo[s] = redo;
    if (redo) atomicAdd(n_mismatch, 1u);
}

__global__ __launch_bounds__(1024) void bam_scan_kernel(const uint32_t* cnt, uint32_t n, unsigned long long* off) {
    __shared__ unsigned long long part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = (n + 1023) / 1024;
    const uint32_t a = (uint64_t)tid * chunk < n ? tid * chunk : n, b = a + chunk < n ? a + chunk : n;
    unsigned long long s = 0;
    for (uint32_t i = a; i < b; ++i) s += cnt[i];
    part[tid] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const unsigned long long v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned long long run = tid ? part[tid - 1] : 0;
    for (uint32_t i = a; i < b; ++i) {
        off[i] = run;
        run += cnt[i];
    }
    if (tid == 1023) off[n] = part[1023];
}

__global__ __launch_bounds__(256) void bam_emit_kernel(BamStream B, uint64_t n_seg, const unsigned long long* guess, const unsigned long long* off,
                                                       const uint32_t* ref_map, gf_alnrec* recs, unsigned long long* rec_begin, uint64_t cap) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    uint32_t c = 0;
    bam_walk<true>(B, guess[s], bam_seg_end(B, s), c, ref_map, recs, rec_begin, off[s], cap);
}

// slice i of the device-resident stream -> dst[dst_off[i] ...), one wavefront per slice
__global__ __launch_bounds__(256) void bam_fetch_kernel(const uint8_t* stream, const unsigned long long* begin, const unsigned long long* end,
                                                        const unsigned long long* dst_off, uint64_t n, uint8_t* dst) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint8_t* s = stream + begin[i];
    uint8_t* d = dst + dst_off[i];
    const uint64_t len = end[i] - begin[i];
    for (uint64_t k = lane; k < len; k += 64) d[k] = s[k];
}

// The record chain of the inflated stream in ctx->bam_stream: per 64-KiB segment the offset of the first record that starts in it
// (guess -> walk -> check rounds until every segment's entry equals its predecessor's exit) and the exclusive scan of the segments'
// record counts.  Workspace in ctx->pool_ws; `extra_ws` more bytes are reserved behind it for the caller (*d_extra).
void bam_scan_launch(gf_ctx* ctx, const uint32_t* cnt, uint32_t n, unsigned long long* off) {
    hipLaunchKernelGGL(bam_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, cnt, n, off);
}

int bam_chain(gf_ctx* ctx, size_t n_bytes, size_t first, size_t n_ref, unsigned long long** d_guess_out, unsigned long long** d_rec_off_out, uint64_t* n_seg_out,
              size_t* n_recs, size_t* n_consumed, size_t extra_ws, uint8_t** d_extra) {
    int rc;
    const uint64_t n_seg = (n_bytes - first + BAM_SEG - 1) / BAM_SEG;
    if (n_seg >= 0xFFFFFFFFull) return GF_E_UNSUPPORTED;
    const size_t b_g = (n_seg * 8 + 63) & ~(size_t)63, b_c = (n_seg * 4 + 63) & ~(size_t)63, b_o = ((n_seg + 1) * 8 + 63) & ~(size_t)63,
                 b_t = (n_seg + 63) & ~(size_t)63;
    if ((rc = ensure(ctx, ctx->pool_ws, 2 * b_g + b_c + b_o + b_t + 128 + extra_ws))) return rc;
    uint8_t* wsp = (uint8_t*)ctx->pool_ws.p;
    unsigned long long* d_guess = (unsigned long long*)wsp;
    unsigned long long* d_exit = (unsigned long long*)(wsp + b_g);
    uint32_t* d_cnt = (uint32_t*)(wsp + 2 * b_g);
    unsigned long long* d_off = (unsigned long long*)(wsp + 2 * b_g + b_c);
    uint8_t* d_todo = wsp + 2 * b_g + b_c + b_o;
    uint32_t* d_flags = (uint32_t*)(wsp + 2 * b_g + b_c + b_o + b_t);   // [0] mismatches
    *d_extra = wsp + 2 * b_g + b_c + b_o + b_t + 128;
    const BamStream B{(const uint8_t*)ctx->bam_stream.p, (uint64_t)n_bytes, (uint64_t)first, (int32_t)n_ref};
    const unsigned grid = (unsigned)((n_seg + 255) / 256);
    LaunchTimer tm(ctx, GF_KERNEL_INGEST);
    hipLaunchKernelGGL(bam_guess_kernel, dim3(grid), dim3(256), 0, ctx->stream, B, n_seg, d_guess);
    hipLaunchKernelGGL(bam_walk_kernel, dim3(grid), dim3(256), 0, ctx->stream, B, n_seg, d_guess, d_exit, d_cnt, (const uint8_t*)nullptr);
    for (uint64_t round = 0;; ++round) {
        uint32_t fl[1] = {0};
        GF_HIP(ctx, hipMemsetAsync(d_flags, 0, 4, ctx->stream));
        hipLaunchKernelGGL(bam_check_kernel, dim3(grid), dim3(256), 0, ctx->stream, n_seg, d_guess, d_exit, d_todo, d_flags);
        GF_HIP(ctx, hipMemcpyAsync(fl, d_flags, 4, hipMemcpyDeviceToHost, ctx->stream));
        GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (fl[0] == 0) break;
        if (round > n_seg) return GF_E_FORMAT;
        hipLaunchKernelGGL(bam_walk_kernel, dim3(grid), dim3(256), 0, ctx->stream, B, n_seg, d_guess, d_exit, d_cnt, (const uint8_t*)d_todo);
    }
    hipLaunchKernelGGL(bam_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_cnt, (uint32_t)n_seg, d_off);
    unsigned long long total = 0, last_exit = 0;
    GF_HIP(ctx, hipMemcpyAsync(&total, d_off + n_seg, 8, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipMemcpyAsync(&last_exit, d_exit + (n_seg - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((last_exit & BAM_TAIL) && (last_exit & ~BAM_TAIL) + 4 <= n_bytes) {   // the chain stopped early: incomplete record, or corrupt?
        int32_t bs = 0;
        GF_HIP(ctx, hipMemcpyAsync(&bs, B.p + (last_exit & ~BAM_TAIL), 4, hipMemcpyDeviceToHost, ctx->stream));
        GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (bs < 32) {
            ctx->last_error = "BAM record at inflated offset " + std::to_string(last_exit & ~BAM_TAIL) + " has block_size " + std::to_string(bs);
            return GF_E_FORMAT;
        }
    }
    *n_recs = (size_t)total;
    *n_consumed = (size_t)(last_exit & ~BAM_TAIL);
    *d_guess_out = d_guess;
    *d_rec_off_out = d_off;
    *n_seg_out = n_seg;
    return GF_OK;
}

}  // namespace gf

using namespace gf;

extern "C" {

int gf_bgzf_inflate(gf_ctx* ctx, const uint8_t* bgzf, size_t n_bytes, const uint8_t* carry, size_t n_carry, uint8_t* out_or_null,
                    size_t cap, size_t* n_out, size_t* n_consumed) {
    if (!ctx || !n_out || !n_consumed || (n_bytes && !bgzf) || (n_carry && !carry)) return GF_E_INVAL;
    *n_out = 0;
    *n_consumed = 0;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    // block table: 18-byte fixed header with the BC subfield (BSIZE = block size - 1), CRC32 + ISIZE at the end
    std::vector<BgzfBlock> blocks;
    size_t p = 0, total = n_carry;
    while (p + 18 <= n_bytes) {
        const uint8_t* h = bgzf + p;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return GF_E_FORMAT;
        const size_t xlen = h[10] | (h[11] << 8);
        if (p + 12 + xlen > n_bytes) break;
        size_t bsize = 0;
        for (size_t q = 12; q + 4 <= 12 + xlen;) {
            const size_t slen = h[q + 2] | (h[q + 3] << 8);
            if (h[q] == 'B' && h[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + xlen) bsize = (size_t)(h[q + 4] | (h[q + 5] << 8)) + 1;
            q += 4 + slen;
        }
        if (bsize < 12 + xlen + 8) return GF_E_FORMAT;
        if (p + bsize > n_bytes) break;   // partial block: the caller brings it back with the next chunk
        BgzfBlock b;
        b.in_off = p + 12 + xlen;
        b.clen = (uint32_t)(bsize - 12 - xlen - 8);
        std::memcpy(&b.crc, h + bsize - 8, 4);
        std::memcpy(&b.isize, h + bsize - 4, 4);
        b.out_off = total;
        b.pad = 0;
        if (b.isize > 65536) return GF_E_FORMAT;
        if (b.isize) blocks.push_back(b);
        total += b.isize;
        p += bsize;
    }
    *n_consumed = p;
    *n_out = total;
    if (out_or_null && cap < total) return GF_E_NOSPACE;
    int rc;
    if ((rc = ensure(ctx, ctx->bam_stream, total + 256))) return rc;
    ctx->bam_stream_len = total;
    uint8_t* d_out = (uint8_t*)ctx->bam_stream.p;
    if (n_carry) GF_HIP(ctx, hipMemcpyAsync(d_out, carry, n_carry, hipMemcpyHostToDevice, ctx->stream));
    if (!blocks.empty()) {
        const size_t b_in = (p + 1024 + 63) & ~(size_t)63, b_blk = (blocks.size() * sizeof(BgzfBlock) + 63) & ~(size_t)63;
        if ((rc = ensure(ctx, ctx->stage_in, b_in + b_blk + blocks.size() * 4 + 64))) return rc;
        uint8_t* d_in = (uint8_t*)ctx->stage_in.p;
        BgzfBlock* d_blk = (BgzfBlock*)(d_in + b_in);
        uint32_t* d_status = (uint32_t*)(d_in + b_in + b_blk);
        GF_HIP(ctx, hipMemcpyAsync(d_in, bgzf, p, hipMemcpyHostToDevice, ctx->stream));
        GF_HIP(ctx, hipMemsetAsync(d_in + p, 0, b_in - p, ctx->stream));
        GF_HIP(ctx, hipMemcpyAsync(d_blk, blocks.data(), blocks.size() * sizeof(BgzfBlock), hipMemcpyHostToDevice, ctx->stream));
        if (blocks.size() >= 0xFFFFFFFFull) return GF_E_UNSUPPORTED;
        {
            LaunchTimer tm(ctx, GF_KERNEL_INGEST);
            hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((unsigned)((blocks.size() + INF_WAVES - 1) / INF_WAVES)), dim3(64 * INF_WAVES), 0, ctx->stream,
                               d_in, d_blk, (uint32_t)blocks.size(), d_out, d_status);
        }
        std::vector<uint32_t> st(blocks.size());
        GF_HIP(ctx, hipMemcpyAsync(st.data(), d_status, st.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        GF_HIP(ctx, hipGetLastError());
        for (size_t i = 0; i < st.size(); ++i)
            if (st[i]) {
                ctx->last_error = "BGZF block " + std::to_string(i) + " (file offset " + std::to_string(blocks[i].in_off) + "): inflate error " +
                                  std::to_string(st[i]) + (st[i] == INF_CRC ? " (CRC mismatch)" : "");
                return GF_E_FORMAT;
            }
    }
    if (out_or_null && total) {
        GF_HIP(ctx, hipMemcpyAsync(out_or_null, d_out, total, hipMemcpyDeviceToHost, ctx->stream));
        GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return GF_OK;
}

int gf_bam_pack(gf_ctx* ctx, const uint8_t* stream_or_null, size_t n_bytes, size_t first, const uint32_t* ref_map, size_t n_ref,
                gf_alnrec* recs, size_t cap_recs, uint64_t* rec_begin_or_null, size_t* n_recs, size_t* n_consumed) {
    if (!ctx || !n_recs || !n_consumed || (cap_recs && !recs) || (n_ref && !ref_map) || first > n_bytes || n_ref > 0x7FFFFFFF) return GF_E_INVAL;
    *n_recs = 0;
    *n_consumed = first;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if (stream_or_null) {
        if ((rc = ensure(ctx, ctx->bam_stream, n_bytes + 256))) return rc;
        GF_HIP(ctx, hipMemcpyAsync(ctx->bam_stream.p, stream_or_null, n_bytes, hipMemcpyHostToDevice, ctx->stream));
        ctx->bam_stream_len = n_bytes;
    } else if (n_bytes != ctx->bam_stream_len || !ctx->bam_stream.p) {
        return GF_E_STATE;   // no stream left on the device by gf_bgzf_inflate, or a different length
    }
    if (first == n_bytes) return GF_OK;
    unsigned long long *d_guess = nullptr, *d_off = nullptr;
    uint64_t n_seg = 0;
    uint8_t* d_extra = nullptr;
    const size_t b_map = (n_ref * 4 + 63) & ~(size_t)63;
    if ((rc = bam_chain(ctx, n_bytes, first, n_ref, &d_guess, &d_off, &n_seg, n_recs, n_consumed, b_map + 64, &d_extra))) return rc;
    uint32_t* d_map = (uint32_t*)d_extra;
    if (n_ref) GF_HIP(ctx, hipMemcpyAsync(d_map, ref_map, n_ref * 4, hipMemcpyHostToDevice, ctx->stream));
    const BamStream B{(const uint8_t*)ctx->bam_stream.p, (uint64_t)n_bytes, (uint64_t)first, (int32_t)n_ref};
    const unsigned grid = (unsigned)((n_seg + 255) / 256);
    const unsigned long long total = *n_recs;
    LaunchTimer tm(ctx, GF_KERNEL_INGEST);
    ctx->bam_n_recs = 0;
    if (total > cap_recs) return GF_E_NOSPACE;
    if (total) {
        // the records also stay on the device (bam_recs) for gf_tag_alignments_bam / gf_tag_low_mapq_bam
        const size_t b_rec = (total * sizeof(gf_alnrec) + 63) & ~(size_t)63, b_rb = rec_begin_or_null ? (total * 8 + 63) & ~(size_t)63 : 0;
        if ((rc = ensure(ctx, ctx->bam_recs, b_rec + b_rb))) return rc;
        gf_alnrec* d_recs = (gf_alnrec*)ctx->bam_recs.p;
        unsigned long long* d_rb = rec_begin_or_null ? (unsigned long long*)((uint8_t*)ctx->bam_recs.p + b_rec) : nullptr;
        hipLaunchKernelGGL(bam_emit_kernel, dim3(grid), dim3(256), 0, ctx->stream, B, n_seg, d_guess, d_off, d_map, d_recs, d_rb, (uint64_t)total);
        GF_HIP(ctx, hipMemcpyAsync(recs, d_recs, total * sizeof(gf_alnrec), hipMemcpyDeviceToHost, ctx->stream));
        if (rec_begin_or_null) GF_HIP(ctx, hipMemcpyAsync(rec_begin_or_null, d_rb, total * 8, hipMemcpyDeviceToHost, ctx->stream));
        GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->bam_n_recs = (size_t)total;
    }
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

int gf_bam_fetch(gf_ctx* ctx, const uint64_t* begin, const uint64_t* end, size_t n, uint8_t* dst, size_t cap, size_t* n_bytes) {
    if (!ctx || !n_bytes || (n && (!begin || !end))) return GF_E_INVAL;
    *n_bytes = 0;
    if (n == 0) return GF_OK;
    if (!ctx->bam_stream.p) return GF_E_STATE;
    std::vector<unsigned long long> h(3 * n);   // begin | end | dst offsets
    size_t total = 0;
    for (size_t i = 0; i < n; ++i) {
        if (begin[i] > end[i] || end[i] > ctx->bam_stream_len) return GF_E_INVAL;
        h[i] = begin[i];
        h[n + i] = end[i];
        h[2 * n + i] = total;
        total += (size_t)(end[i] - begin[i]);
    }
    *n_bytes = total;
    if (total > cap || (total && !dst)) return GF_E_NOSPACE;
    if (total == 0) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    const size_t b_idx = (3 * n * 8 + 63) & ~(size_t)63;
    if ((rc = ensure(ctx, ctx->stage_out, b_idx + total + 64))) return rc;
    unsigned long long* d_idx = (unsigned long long*)ctx->stage_out.p;
    uint8_t* d_dst = (uint8_t*)ctx->stage_out.p + b_idx;
    GF_HIP(ctx, hipMemcpyAsync(d_idx, h.data(), 3 * n * 8, hipMemcpyHostToDevice, ctx->stream));
    if ((n + 3) / 4 >= 0xFFFFFFFFull) return GF_E_UNSUPPORTED;
    hipLaunchKernelGGL(bam_fetch_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, (const uint8_t*)ctx->bam_stream.p, d_idx, d_idx + n,
                       d_idx + 2 * n, (uint64_t)n, d_dst);
    GF_HIP(ctx, hipMemcpyAsync(dst, d_dst, total, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

}  // extern "C"
