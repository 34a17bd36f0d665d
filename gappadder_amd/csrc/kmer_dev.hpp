// kmer_dev.hpp — 2-bit k-mer arithmetic for gfx950 (host + device).
//
// Bit layout follows the reference's KmerUtils (ContigsCompactor-v0.2.0/ContigsMerger/KmerUtils.cpp:22-58):
// A=00 C=01 G=10 T=11, base i of a k-mer at bits (W-1-2i, W-2-2i) of a W-bit word — MSB-first,
// left-aligned, unused low bits zero.  W = 64 for k <= 32 (the reference's KmerTypeShort) and the same
// layout extended to 128 bits (hi:lo) for k <= 64.  The reference has no reverse complement; canonical =
// min(fwd, revcomp) on that value, i.e. lexicographic with A<C<G<T.
//
// Packed reads use the same convention per byte (base i in byte i/4, MSB-first), so a big-endian load of the
// byte stream is the left-aligned k-mer stream.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GF_HD __host__ __device__ __forceinline__
#else
#define GF_HD inline
#endif

namespace gf {

struct K128 {
    uint64_t hi, lo;
};

GF_HD bool operator<(const K128& a, const K128& b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
GF_HD bool operator==(const K128& a, const K128& b) { return a.hi == b.hi && a.lo == b.lo; }

GF_HD uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

// reverse the order of the 2-bit groups of x
GF_HD uint32_t revpairs32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    x = __brev(x);
#else
    x = ((x >> 16) | (x << 16));
    x = ((x & 0xFF00FF00u) >> 8) | ((x & 0x00FF00FFu) << 8);
    x = ((x & 0xF0F0F0F0u) >> 4) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x & 0xCCCCCCCCu) >> 2) | ((x & 0x33333333u) << 2);
    x = ((x & 0xAAAAAAAAu) >> 1) | ((x & 0x55555555u) << 1);
#endif
    return ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);
}

GF_HD uint64_t revpairs64(uint64_t x) {
    return ((uint64_t)revpairs32((uint32_t)x) << 32) | revpairs32((uint32_t)(x >> 32));
}

// canonical form of a 16-mer held in a whole 32-bit word
GF_HD uint32_t canon16(uint32_t w) {
    uint32_t rc = revpairs32(~w);
    return w < rc ? w : rc;
}

// reverse complement of a left-aligned k-mer (k <= 64)
GF_HD K128 revcomp(K128 v, int k) {
    uint64_t rh = revpairs64(~v.lo), rl = revpairs64(~v.hi);  // (rh:rl) = reversed complement, right-aligned
    int sh = 128 - 2 * k;                                      // shift left to re-align
    K128 r;
    if (sh == 0) {
        r.hi = rh; r.lo = rl;
    } else if (sh < 64) {
        r.hi = (rh << sh) | (rl >> (64 - sh));
        r.lo = rl << sh;
    } else if (sh == 64) {
        r.hi = rl; r.lo = 0;
    } else {
        r.hi = rl << (sh - 64);
        r.lo = 0;
    }
    return r;
}

GF_HD K128 mask_k(K128 v, int k) {
    int bits = 2 * k;
    if (bits >= 128) return v;
    if (bits > 64) {
        v.lo &= ~0ull << (128 - bits);
    } else {
        v.lo = 0;
        v.hi = bits == 64 ? v.hi : (bits == 0 ? 0 : (v.hi & (~0ull << (64 - bits))));
    }
    return v;
}

GF_HD K128 canonical(K128 f, int k) {
    K128 r = revcomp(f, k);
    return (r < f) ? r : f;
}

// hash of a k-mer key -> table slot (capacity = 2^log2cap)
GF_HD uint32_t hash_kmer(K128 v, int log2cap) {
    uint64_t x = v.hi ^ (v.lo * 0x9E3779B97F4A7C15ull) ^ (v.lo >> 29);
    x ^= x >> 31;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    return (uint32_t)(x >> (64 - log2cap));
}

GF_HD uint32_t hash_s16_bitmap(uint32_t key, int log2bits) { return (key * 0x9E3779B1u) >> (32 - log2bits); }
// second bit of a key inside its level-1 bitmap word: the low product bits, which no word or first-bit index uses while
// the bitmap has <= 2^27 bits (beyond that the two overlap: still exact, only less selective)
GF_HD uint32_t hash_s16_bit2(uint32_t key) { return (key * 0x9E3779B1u) & 31u; }
// third bit, set only in bitmaps of >= 2^27 bits (S16_BIT3_MIN_LOG2: the key sets the partitioned filter takes): every product bit is
// taken by the word index and the first two bits by then, so the third position is a function of those two — a probe that finds its
// first two bits set by OTHER keys still has to find this one set (false positives of the word test 0.67 % -> 0.17 % at C4's 1.1e7 keys
// in 2^28 bits).  Kernels that test two bits stay exact (a superset passes); the partitioned filter's pass B tests all three.
constexpr int S16_BIT3_MIN_LOG2 = 27;
GF_HD uint32_t s16_bit3_of(uint32_t b1, uint32_t b2) { return (b1 * 7u + b2 * 13u + 5u) & 31u; }
GF_HD uint32_t hash_s16_bit3(uint32_t key, int log2bits) { return s16_bit3_of(hash_s16_bitmap(key, log2bits) & 31u, hash_s16_bit2(key)); }
GF_HD uint32_t hash_s16_set(uint32_t key, int log2cap) {
    uint32_t x = key * 0x85EBCA6Bu;
    x ^= x >> 15;
    x *= 0xC2B2AE35u;
    return x >> (32 - log2cap);
}

// base code per KmerUtils.cpp:25-41: anything that is not C/G/T (either case) is A
GF_HD uint32_t base_code(char c) {
    return (c == 'C' || c == 'c') ? 1u : (c == 'G' || c == 'g') ? 2u : (c == 'T' || c == 't') ? 3u : 0u;
}

// ---- bit-stream access to packed bases held as bytes (big-endian base order) ------------------------
// `words` is the byte stream viewed as little-endian uint32 (as a GPU/x86 load sees it); returns the 32 bits
// starting at bit offset `bitoff` of the stream, first base in the top bits.
template <typename P>
GF_HD uint32_t stream32(P words, uint32_t bitoff) {
    uint32_t d = bitoff >> 5, sh = bitoff & 31;
    uint64_t v = ((uint64_t)bswap32(words[d]) << 32) | bswap32(words[d + 1]);
    return (uint32_t)((v << sh) >> 32);
}

// same for a BYTE-aligned offset: on the device the unaligned fetch and the byte swap collapse into one v_perm_b32
template <typename P>
GF_HD uint32_t stream32_bytes(P words, uint32_t byteoff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t d = byteoff >> 2, o = byteoff & 3;
    const uint32_t sel = 0x00010203u + o * 0x01010101u;   // output bytes (MSB..LSB) = stream bytes o, o+1, o+2, o+3
    return __builtin_amdgcn_perm(words[d + 1], words[d], sel);
#else
    return stream32(words, byteoff * 8);
#endif
}

// the k-mer starting at bit offset `bitoff` (k <= 64), left-aligned and masked
template <typename P>
GF_HD K128 stream_kmer(P words, uint32_t bitoff, int k) {
    K128 v;
    v.hi = ((uint64_t)stream32(words, bitoff) << 32) | stream32(words, bitoff + 32);
    v.lo = 0;
    if (k > 32) v.lo = ((uint64_t)stream32(words, bitoff + 64) << 32) | stream32(words, bitoff + 96);
    return mask_k(v, k);
}

// k <= 32: the same arithmetic on the one 64-bit word that carries the k-mer (the low word is zero)
template <typename P>
GF_HD uint64_t stream_kmer64(P words, uint32_t bitoff, int k) {
    const uint64_t v = ((uint64_t)stream32(words, bitoff) << 32) | stream32(words, bitoff + 32);
    return v & (~0ull << (64 - 2 * k));
}
GF_HD uint64_t canonical64(uint64_t f, int k) {
    const uint64_t r = revpairs64(~f) << (64 - 2 * k);
    return r < f ? r : f;
}

}  // namespace gf
