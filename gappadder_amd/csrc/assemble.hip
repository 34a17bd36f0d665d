// assemble.hip — per-gap local assembly: what `run_assembly` (assemble_gaps.py:82-136) obtains from
// `kmc -k{k}` | `kmc_dump` | `velveth {kv}` | `velvetg -min_contig_lgth 40`, for one (k, kv) pair per launch.
// Semantics (PARITY UNPINNED, defined by this build): oracle/gp_oracle.c `or_assemble_pool2`, DESIGN.md §2.
//
// One workgroup per gap.  The gap's packed read pool is staged in LDS when it fits; the open-addressing tables (canonical k-mer ->
// count, canonical kv-mer -> adjacency/multiplicity) live in LDS when the gap fits an LDS plan, else in a per-workgroup slice of a
// global workspace, and store 32-bit INSTANCE ids (read-in-pool << 10 | offset) instead of keys, so one 64-bit CAS claims a slot
// for any k <= 64 and a key is re-derived from the staged reads when slots are compared.
//   P1 count      bit-array pre-count (k-mers seen fewer than min_count times never enter the table), then every remaining k-mer
//                 window -> canonical -> table (key / fingerprint / instance-id slots, ranked behind the pre-count)
//   P2 survivors  slots with count >= min_count -> compacted instance list (bit 31: the k-mer is WEAK, count <= min_count + 1)
//   P3 graph      each survivor contributes k-kv+1 kv-mer nodes and k-kv edges (4+4 adjacency bits per node); dense node indices
//   P4 links      unitig-internal edges; oriented nodes that no internal edge enters start a unitig
//   ER            error removal rounds: tip clipping + bubble popping on snapshots of the graph (heads walk their unitigs)
//   P5 ranking    every oriented node learns its unitig's head and its rank by pointer jumping
//   P6 emission   the walk whose first kv-mer <= the opposite walk's is emitted; every node writes its own base
#include <type_traits>

#include "gf_internal.hpp"

namespace gf {

// Threads per workgroup = per gap: 1024 (one workgroup per CU with all 152 KiB of LDS: deep pools), 512 (two per CU, 76 KiB each)
// or 256 (four per CU, 38 KiB each).  The CU runs 16 waves either way; with several gaps per CU one gap's barriers and its
// poorly parallel phases (graph construction over a few thousand survivors, error-removal walks, ranking rounds) overlap another
// gap's work instead of leaving waves idle.
constexpr uint32_t ASM_DEFER_STAY = 64;   // gaps of a main launch that stay in it although their graph will not fit its share of the LDS (see `defer`)
constexpr uint32_t ASM_LDS_MAX_WORDS = 38 * 1024;  // 152 KiB of dynamic LDS per CU: staged reads, then tables and per-node arrays

struct AsmParams {
    const uint32_t* reads32;   // packed pool reads as little-endian words
    uint64_t n_words;          // whole words of the packed array
    uint32_t tail_bytes;
    const uint32_t* nmask;     // may be null
    const uint64_t* pool_off;  // n_pools + 1, in reads
    uint32_t n_pools;
    uint64_t total_reads;      // rows of the pool array = what the workspace slices are sized for
    uint32_t rb, read_len, k, kv, nmw;
    uint32_t min_count, min_contig;
    unsigned long long* table; // 4 slots per k-mer instance: low 32 = instance id, high 32 = count / node meta
    uint32_t* surv;            // 2 words per instance: slot lists, survivor list, node instance ids, emitted-walk records
    uint32_t* nodes;           // 3 words per instance: per-node meta + succ[2] when they do not fit in LDS
    uint32_t* jump;            // 4 words per instance: the unitig-ranking pairs of the oriented nodes when they do not fit in LDS
    uint32_t simplify;         // rounds of tip clipping + bubble popping (0: raw unitigs)
    uint32_t tiebreak_counts;  // 1: between equal coverage the side with fewer WEAK nodes wins (option asm_tiebreak; 0: that rule is off)
    uint32_t slice_rows;       // > 0: the workspace holds one slice of this many pool rows per workgroup
    uint64_t slice_base;       // ... the first of them starts at this instance offset
    uint32_t* defer_tickets;   // (defer) counter of the gaps that asked to leave because of their GRAPH: the first ASM_DEFER_STAY of them stay
    uint32_t defer;            // 1: a gap whose count table does not fit the LDS is not counted in the global slice but listed (big_list) for the next launch
    uint32_t accept_rows;      // ... and the launch takes the pools of up to this many rows (<= slice_rows), lists the others
    uint64_t slice_stride;     // ... and they lie this many instances apart (slice_rows x the unit of this launch, or of the largest unit of a sweep)
    // pools beyond slice_rows are not refused: the launch lists them (big_list, *n_big) and a second, small launch takes them from
    // that list (gap_list, *n_gap_list) with slices of its own, sized for deep pools (option asm_big_pool_reads)
    uint32_t* big_list;
    uint32_t* n_big;
    const uint32_t* gap_list;
    const uint32_t* n_gap_list;
    gf_contig* contigs;
    uint32_t contig_cap;
    uint32_t* n_contigs;
    char* seq;
    uint64_t seq_cap;
    unsigned long long* seq_len;
    uint32_t* gap_error;       // per pool: 1 = a table or list overflowed (results of that pool incomplete)
    // count-only mode (gf_count_kmers): survivors of pool 0 are written here instead of being assembled
    uint64_t* cnt_keys;        // 2 x u64 per k-mer (hi, lo), or null
    uint32_t* cnt_counts;
    uint32_t cnt_cap;
    uint32_t lds_words;        // dynamic LDS given to the staged pool
    uint32_t* next_gap;        // work counter, zero at launch
    uint32_t keyslot;          // allow the key-in-slot count phase
    uint32_t ranked;           // allow the ranked table behind the pre-count
    uint32_t precount;         // allow the bit-array pre-count (k-mers seen fewer than min_count times never enter the table)
    uint32_t pre_frac8;        // eighths of the LDS region the pre-count's bit arrays may take under an LDS table (option asm_pre_frac8)
    unsigned long long* stats; // or null: [0] += read windows, [1] += k-mers counted exactly, [2] += surviving k-mers, [3] += nodes (the
                               // assembly's algorithmic bytes, SURVEY.md §8d: 38 B x pool reads + 2 x 20 B x distinct k-mers + contig bases)
    unsigned long long* dbg;   // diagnostic runs only: 16 wall-clock stamps per gap (100 MHz), or null
};

// gap_error bits
constexpr uint32_t ASM_ERR_IDS = 1, ASM_ERR_KTABLE = 2, ASM_ERR_NTABLE = 4, ASM_ERR_NLIST = 8, ASM_ERR_WALKS_PAR = 16;

// node meta bits (high word of a table slot in the graph phases)
constexpr uint32_t M_OUT = 0xFu, M_IN = 0xF0u, M_START0 = 1u << 8, M_START1 = 1u << 9, M_MULT_SHIFT = 14;
// WEAK = one of the surviving k-mers the node came from was seen at most min_count + 1 times (the tie-break of the error removal)
constexpr uint32_t M_WEAK = 1u << 13, INST_WEAK = 1u << 31;
// error removal: KILL = the unitig this node heads is removed in this round; DEADMARK -> DEAD = the node is gone
constexpr uint32_t M_KILL = 1u << 10, M_DEADMARK = 1u << 11, M_DEAD = 1u << 12;

// the kernel's dynamic LDS, at file scope so that every helper addresses it as LDS (ds_* instructions) instead of through
// a generic pointer (flat_* instructions, several times the latency)
extern __shared__ uint32_t g_lds[];

struct PoolView {
    uint32_t rb, L;
    bool lds;
    uint64_t first_byte;  // global byte offset of the pool (global view only)
    const uint32_t* g32;  // the whole packed array in global memory
    uint64_t g_words;
    uint32_t g_tail;
};

__device__ __forceinline__ uint32_t asm_word(const uint32_t* g32, uint64_t n_words, uint32_t tail_bytes, uint64_t w) {
    if (w < n_words) return g32[w];
    uint32_t v = 0;
    if (w == n_words) {
        const uint8_t* t = reinterpret_cast<const uint8_t*>(g32 + n_words);
        for (uint32_t i = 0; i < tail_bytes; ++i) v |= (uint32_t)t[i] << (8 * i);
    }
    return v;
}

// 32 bits of the pool's base stream starting at bit `bit` (first base in the top bits)
__device__ __forceinline__ uint32_t pv_stream32(const PoolView& V, uint64_t bit) {
    if (V.lds) {   // staged at the start of the dynamic LDS
        const uint32_t d = (uint32_t)(bit >> 5), sh = (uint32_t)bit & 31;
        const uint64_t v = ((uint64_t)bswap32(g_lds[d]) << 32) | bswap32(g_lds[d + 1]);
        return (uint32_t)((v << sh) >> 32);
    }
    const uint64_t gb = V.first_byte * 8 + bit;
    const uint64_t d = gb >> 5;
    const uint32_t sh = (uint32_t)gb & 31;
    const uint64_t v = ((uint64_t)bswap32(asm_word(V.g32, V.g_words, V.g_tail, d)) << 32) |
                       bswap32(asm_word(V.g32, V.g_words, V.g_tail, d + 1));
    return (uint32_t)((v << sh) >> 32);
}

// instance id of a read window: read-in-pool << 10 | offset (read_len <= 1000; a pool has fewer than 2^22 reads): no division
// when a key is re-derived from an id, and `id + o` is the window o bases further on in the same read
constexpr uint32_t INST_OFF_BITS = 10, INST_OFF_MASK = (1u << INST_OFF_BITS) - 1;
__device__ __forceinline__ uint32_t make_inst(uint32_t r, uint32_t off) { return (r << INST_OFF_BITS) | off; }

// left-aligned `len`-mer at (read r, offset off) of the pool staged at the start of the dynamic LDS: six dwords, five byte
// permutes (unaligned fetch + byte swap in one v_perm each) and a funnel shift by the 0/2/4/6 bits the offset leaves
template <bool W>
__device__ __forceinline__ K128 lds_window(uint32_t rb, uint32_t r, uint32_t off, int len) {
    const uint32_t byte = r * rb + (off >> 2);
    const uint32_t w = byte >> 2;
    const uint32_t sel = 0x00010203u + (byte & 3u) * 0x01010101u;
    const uint32_t sh = 2u * (off & 3u);
    const uint32_t a0 = g_lds[w], a1 = g_lds[w + 1], a2 = g_lds[w + 2], a3 = g_lds[w + 3];
    const uint32_t b0 = __builtin_amdgcn_perm(a1, a0, sel), b1 = __builtin_amdgcn_perm(a2, a1, sel), b2 = __builtin_amdgcn_perm(a3, a2, sel);
    const uint64_t x0 = ((uint64_t)b0 << 32) | b1;
    K128 v;
    if (!W) {
        v.hi = ((x0 << sh) | (((uint64_t)b2 >> 1) >> (31 - sh))) & (~0ull << (64 - 2 * len));
        v.lo = 0;
        return v;
    }
    const uint32_t a4 = g_lds[w + 4], a5 = g_lds[w + 5];
    const uint32_t b3 = __builtin_amdgcn_perm(a4, a3, sel), b4 = __builtin_amdgcn_perm(a5, a4, sel);
    const uint64_t x1 = ((uint64_t)b2 << 32) | b3, x2 = (uint64_t)b4 << 32;
    v.hi = (x0 << sh) | ((x1 >> 1) >> (63 - sh));
    v.lo = (x1 << sh) | ((x2 >> 1) >> (63 - sh));
    return mask_k(v, len);
}

// same for a (read, offset) pair of any pool view
template <bool W>
__device__ __forceinline__ K128 pv_kmer_at(const PoolView& V, uint32_t r, uint32_t off, int len) {
    if (V.lds) return lds_window<W>(V.rb, r, off, len);
    const uint64_t bit = (uint64_t)r * V.rb * 8 + 2ull * off;
    K128 v;
    v.hi = ((uint64_t)pv_stream32(V, bit) << 32) | pv_stream32(V, bit + 32);
    v.lo = 0;
    if (!W) {
        v.hi &= ~0ull << (64 - 2 * len);
        return v;
    }
    if (len > 32) v.lo = ((uint64_t)pv_stream32(V, bit + 64) << 32) | pv_stream32(V, bit + 96);
    return mask_k(v, len);
}
// left-aligned `len`-mer at an instance id
// W = false: the launch has k <= 32, so every key is one 64-bit word (lo == 0) and the 128-bit halves compile away
template <bool W>
__device__ __forceinline__ K128 pv_kmer(const PoolView& V, uint32_t inst, int len) {
    return pv_kmer_at<W>(V, inst >> INST_OFF_BITS, inst & INST_OFF_MASK, len);
}
template <bool W>
__device__ __forceinline__ K128 revcomp_w(K128 v, int len) {
    if (W) return revcomp(v, len);
    K128 r;
    r.hi = revpairs64(~v.hi) << (64 - 2 * len);
    r.lo = 0;
    return r;
}
template <bool W>
__device__ __forceinline__ K128 canonical_w(K128 f, int len) {
    if (W) return canonical(f, len);
    K128 r = revcomp_w<false>(f, len);
    r.hi = r.hi < f.hi ? r.hi : f.hi;
    return r;
}

__device__ __forceinline__ uint64_t hash_of(K128 key) {
    uint64_t x = key.hi ^ (key.lo * 0x9E3779B97F4A7C15ull) ^ (key.lo >> 29);
    x ^= x >> 31;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    return x;
}
__device__ __forceinline__ uint32_t slot_of_hash(uint64_t x, uint32_t cap) { return (uint32_t)(((x & 0xFFFFFFFFull) * cap) >> 32); }
__device__ __forceinline__ uint32_t slot_of(K128 key, uint32_t cap) { return slot_of_hash(hash_of(key), cap); }
// bit of a k-mer in the pre-count bit arrays (2^bits_log2 bits): the hash's high word mixed once more, so that k-mers which share
// a bit do not also share their table neighbourhood (slot = low word) or their fingerprint (bits 34..63)
// hash of the count phase (internal to it: slot from the low word, fingerprint and pre-count bit from the high word): six 32-bit
// multiplies where hash_of spends three 64-bit ones (twelve quarter-rate instructions) — the phase is issue-bound
template <bool W>
__device__ __forceinline__ uint64_t hash_p1(K128 key) {
    const uint32_t a = (uint32_t)(key.hi >> 32), b = (uint32_t)key.hi;
    const uint32_t c = W ? (uint32_t)(key.lo >> 32) : a, d = W ? (uint32_t)key.lo : b;
    const uint32_t u = (a * 0x9E3779B1u) ^ (b * 0x85EBCA77u);
    const uint32_t v = (c * 0xC2B2AE3Du) ^ (d * 0x27D4EB2Fu);
    uint32_t h1 = u ^ ((v << 15) | (v >> 17));
    h1 ^= h1 >> 15; h1 *= 0x2C1B3C6Du; h1 ^= h1 >> 12;
    uint32_t h2 = v ^ ((u << 13) | (u >> 19));
    h2 ^= h2 >> 16; h2 *= 0x297A2D39u; h2 ^= h2 >> 15;
    return ((uint64_t)h2 << 32) | h1;
}

__device__ __forceinline__ uint32_t* slot_id(unsigned long long* t, uint32_t s) { return reinterpret_cast<uint32_t*>(t + s); }
__device__ __forceinline__ uint32_t* slot_meta(unsigned long long* t, uint32_t s) { return reinterpret_cast<uint32_t*>(t + s) + 1; }
__device__ __forceinline__ unsigned long long slot_load(const unsigned long long* t, uint32_t s) {
    return __hip_atomic_load(t + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr unsigned long long EMPTY64 = 0x00000000FFFFFFFFull;  // id EMPTY32, count / meta 0

// A hash table of 8-byte slots (low word = instance id or node index, high word = count / node meta) that lives either in
// the dynamic LDS (word offset `off`) or in global memory (`g`).  Every access branches on the wave-uniform `lds` flag so
// that the LDS side compiles to ds_* instructions.
struct Tab {
    bool lds;
    uint32_t off;             // word offset in g_lds (8-byte aligned)
    unsigned long long* g;
    uint32_t cap;
    __device__ __forceinline__ unsigned long long* l() const { return reinterpret_cast<unsigned long long*>(&g_lds[off]); }
    __device__ __forceinline__ unsigned long long load(uint32_t s) const {
        return lds ? __hip_atomic_load(l() + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                   : __hip_atomic_load(g + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void store(uint32_t s, unsigned long long v) const { if (lds) l()[s] = v; else g[s] = v; }
    __device__ __forceinline__ unsigned long long cas(uint32_t s, unsigned long long e, unsigned long long d) const {
        return lds ? atomicCAS(l() + s, e, d) : atomicCAS(g + s, e, d);
    }
    __device__ __forceinline__ void add(uint32_t s, unsigned long long v) const { if (lds) atomicAdd(l() + s, v); else atomicAdd(g + s, v); }
    __device__ __forceinline__ void or_meta(uint32_t s, uint32_t bits) const {
        if (lds) atomicOr(&g_lds[off + 2 * s + 1], bits); else atomicOr(reinterpret_cast<uint32_t*>(g + s) + 1, bits);
    }
    __device__ __forceinline__ void set_id(uint32_t s, uint32_t id) const {
        if (lds) g_lds[off + 2 * s] = id; else *reinterpret_cast<uint32_t*>(g + s) = id;
    }
    __device__ __forceinline__ uint32_t id(uint32_t s) const { return (uint32_t)load(s); }
};

// a uint32 array in LDS (word offset) or global memory
struct Arr {
    bool lds;
    uint32_t off;
    uint32_t* g;
    __device__ __forceinline__ uint32_t get(uint32_t i) const { return lds ? g_lds[off + i] : __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void set(uint32_t i, uint32_t v) const { if (lds) g_lds[off + i] = v; else g[i] = v; }
    __device__ __forceinline__ void or_(uint32_t i, uint32_t v) const { if (lds) atomicOr(&g_lds[off + i], v); else atomicOr(g + i, v); }
    __device__ __forceinline__ void and_(uint32_t i, uint32_t v) const { if (lds) atomicAnd(&g_lds[off + i], v); else atomicAnd(g + i, v); }
};

// unitig-internal successor of an ORIENTED node o = 2 * node + orientation.  In LDS one 16-bit entry per oriented node (an LDS plan
// holds fewer than 32 767 nodes: an oriented node fits 16 bits, 0xFFFF = none) — half a word per entry instead of one; in global memory
// two 32-bit arrays (orientation 0, orientation 1)
struct SuccArr {
    bool lds;
    uint32_t off;        // LDS: word offset of the 2 * nb 16-bit entries
    uint32_t* g;         // global: succ0 at g, succ1 at g + gstride
    uint32_t gstride;
    __device__ __forceinline__ uint32_t get(uint32_t o) const {
        if (lds) { const uint32_t v = reinterpret_cast<const uint16_t*>(&g_lds[off])[o]; return v == 0xFFFFu ? EMPTY32 : v; }
        return g[(o & 1u) * gstride + (o >> 1)];
    }
    __device__ __forceinline__ void set(uint32_t o, uint32_t v) const {
        if (lds) reinterpret_cast<uint16_t*>(&g_lds[off])[o] = (uint16_t)v;      // (EMPTY32 -> 0xFFFF)
        else g[(o & 1u) * gstride + (o >> 1)] = v;
    }
};

// 8-byte pairs in LDS (word offset, 8-byte aligned) or global memory, read and written whole
// In LDS a pair is ONE 32-bit word {high half << 16 | low half}: both halves of every pair the graph phase keeps there (an oriented
// node and a distance / a rank / a base code) stay below 65 535 because an LDS plan holds fewer than 32 767 nodes.  "No arc" is a value
// that survives the packing.
constexpr unsigned long long NO_ARC = 0x0000FFFF0000FFFFull;
struct Pairs {
    bool lds;
    uint32_t off;
    unsigned long long* g;
    __device__ __forceinline__ unsigned long long load(uint32_t i) const {
        if (!lds) return __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t w = __hip_atomic_load(&g_lds[off + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return ((unsigned long long)(w >> 16) << 32) | (w & 0xFFFFu);
    }
    __device__ __forceinline__ void store(uint32_t i, unsigned long long v) const {
        if (lds) __hip_atomic_store(&g_lds[off + i], ((uint32_t)(v >> 32) << 16) | ((uint32_t)v & 0xFFFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(g + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// Find-or-insert the canonical `len`-mer `key` (instance `inst`) and add `inc` to the slot's high word, with ONE 64-bit
// CAS when the slot is free (id and first count land together) or one 64-bit add when the key is already there.
// Returns the slot, EMPTY32 when the table is full.
template <bool W>
__device__ __forceinline__ uint32_t table_upsert(const Tab& t, const PoolView& V, K128 key, uint32_t inst, int len, uint32_t inc,
                                                 bool* fresh) {
    uint32_t s = slot_of(key, t.cap);
    *fresh = false;
    for (uint32_t probes = 0; probes < t.cap; ++probes) {
        unsigned long long v = t.load(s);
        if ((uint32_t)v == EMPTY32) {
            v = t.cas(s, EMPTY64, ((unsigned long long)inc << 32) | inst);
            if (v == EMPTY64) { *fresh = true; return s; }
        }
        const uint32_t cur = (uint32_t)v;
        if (cur == inst || canonical_w<W>(pv_kmer<W>(V, cur, len), len) == key) {
            t.add(s, (unsigned long long)inc << 32);
            return s;
        }
        s = s + 1 == t.cap ? 0 : s + 1;
    }
    return EMPTY32;
}

// The same for the NODE table (kv-mers), with an 8-bit key fingerprint in the top byte of the slot's high word when `fp_on`: a probe
// that meets another node sees it in the fingerprint and does not re-derive the occupant's kv-mer from the reads (a window fetch + a
// reverse complement per foreign slot on the way).  The multiplicity lives in bits 14..23 then, which bounds it by 1023: the caller
// switches the fingerprint on only when k - kv <= 3 (a kv-mer lies in at most (k - kv + 1) * 4^(k - kv) <= 256 distinct k-mers).
constexpr uint32_t NODE_FP_SHIFT = 24;
__device__ __forceinline__ uint32_t node_fp(uint64_t x) { return (uint32_t)(x >> 56); }
// The probe sequence is walked in TWO nested steps: a cheap scan to the next slot that is empty or carries this key's fingerprint, then —
// outside that scan — the claim or the exact comparison with the occupant's kv-mer (re-derived from the staged reads: ~50 instructions
// and 5 LDS reads).  With the comparison inside the one probe loop every iteration of a wave ran it as soon as any of its 64 lanes
// stood at a matching fingerprint: ~760 instructions per upsert and wave at C4; almost every lane needs it exactly once.
template <bool W>
__device__ __forceinline__ uint32_t node_upsert(const Tab& t, const PoolView& V, K128 key, uint32_t inst, int len, uint32_t inc, bool fp_on,
                                                bool* fresh) {
    const uint64_t x = hash_p1<W>(key);   // (as in the count phase: six 32-bit multiplies instead of three 64-bit ones)
    const uint32_t fp = fp_on ? node_fp(x) : 0u;
    uint32_t s = slot_of_hash(x, t.cap);
    *fresh = false;
    uint32_t probes = 0;
    while (probes < t.cap) {
        unsigned long long v = t.load(s);
        while ((uint32_t)v != EMPTY32 && fp_on && (uint32_t)(v >> (32 + NODE_FP_SHIFT)) != fp) {   // another key's slot: next
            s = s + 1 == t.cap ? 0 : s + 1;
            if (++probes >= t.cap) return EMPTY32;
            v = t.load(s);
        }
        if ((uint32_t)v == EMPTY32) {
            v = t.cas(s, EMPTY64, ((unsigned long long)(inc | (fp << NODE_FP_SHIFT)) << 32) | inst);
            if (v == EMPTY64) { *fresh = true; return s; }
            if (fp_on && (uint32_t)(v >> (32 + NODE_FP_SHIFT)) != fp) {   // claimed meanwhile, by another key
                s = s + 1 == t.cap ? 0 : s + 1;
                ++probes;
                continue;
            }
        }
        const uint32_t cur = (uint32_t)v;
        if (cur == inst || canonical_w<W>(pv_kmer<W>(V, cur, len), len) == key) {
            t.add(s, (unsigned long long)inc << 32);
            return s;
        }
        s = s + 1 == t.cap ? 0 : s + 1;
        ++probes;
    }
    return EMPTY32;
}

__device__ __forceinline__ uint32_t rev4(uint32_t b) {  // bit c -> bit 3-c
    return ((b & 1) << 3) | ((b & 2) << 1) | ((b & 4) >> 1) | ((b & 8) >> 3);
}
__device__ __forceinline__ uint32_t out_bits(uint32_t meta, uint32_t d) { return d ? rev4((meta & M_IN) >> 4) : (meta & M_OUT); }
__device__ __forceinline__ uint32_t in_bits(uint32_t meta, uint32_t d) { return d ? rev4(meta & M_OUT) : ((meta & M_IN) >> 4); }

__device__ __forceinline__ uint32_t kbase(K128 v, int i) {
    return i < 32 ? (uint32_t)(v.hi >> (62 - 2 * i)) & 3u : (uint32_t)(v.lo >> (62 - 2 * (i - 32))) & 3u;
}
// append base c after dropping the first base of a left-aligned len-mer
__device__ __forceinline__ K128 shift_in(K128 v, uint32_t c, int len) {
    K128 r;
    r.hi = (v.hi << 2) | (v.lo >> 62);
    r.lo = v.lo << 2;
    const int i = len - 1;
    if (i < 32) r.hi |= (uint64_t)c << (62 - 2 * i); else r.lo |= (uint64_t)c << (62 - 2 * (i - 32));
    return mask_k(r, len);
}
// prepend base c, dropping the last base
__device__ __forceinline__ K128 shift_in_front(K128 v, uint32_t c, int len) {
    K128 r;
    r.lo = (v.lo >> 2) | (v.hi << 62);
    r.hi = (v.hi >> 2) | ((uint64_t)c << 62);
    return mask_k(r, len);
}

// Phase boundary inside one workgroup whose phases hand data to each other through GLOBAL memory (lists, per-node arrays
// in the global fallback): the vector L1 is write-through, so after every wave's vmcnt(0) (part of the barrier) the stores
// are in L2; one lane then invalidates this CU's L1 (acquire, agent scope = buffer_inv sc1, ~1.7 us) so that lines cached
// before an atomic or a rewrite are not served stale.  Replaces 1024 x __threadfence() (L2 write-back + invalidate each).
// does the window [p, p + len) of a read touch a masked base (N, or the padding behind a short read)?  row = the read's mask words (bit i = base
// i), nmw of them; len <= 64, so the window spans at most three words: one 96-bit shift instead of a loop over the window's bases (the host
// entry points pass masks for every FASTQ-born pool: the loop was 41 global loads per window and pass at k = 41 — 56 ms for a launch that
// takes 2 ms without masks, rocprofv3 of the CLI, profiles/r05_kernel_stats_e2e_c3.csv)
__device__ __forceinline__ bool window_masked(const uint32_t* row, uint32_t nmw, uint32_t p, uint32_t len) {
    const uint32_t w = p >> 5, sh = p & 31;
    const uint64_t lo = (uint64_t)row[w] | ((uint64_t)(w + 1 < nmw ? row[w + 1] : 0u) << 32);
    const uint64_t hi = w + 2 < nmw ? row[w + 2] : 0u;
    const uint64_t a = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
    return (len >= 64 ? a : a & ((1ull << len) - 1)) != 0;
}

__device__ __forceinline__ void wg_phase_sync() {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

#define ASM_STAMP(n) do { if (P.dbg && tid == 0) P.dbg[(uint64_t)g * 16 + (n)] = wall_clock64(); } while (0)

// LDS plan of one gap (dynamic LDS = P.lds_words words):  [ staged pool | region R ]
//   count phase   R = k-mer table (8-B slots) when the distinct k-mers keep it under 3/4 full, else the global slice
//   graph phase   R = node table (2 slots per possible node) + inst_of/meta/succ0/succ1 arrays, else global
// Every pointer below is generic (LDS or global); the code path is the same.
// KC: k as a compile-time constant with kv = k - 2 (the pipeline's pairs 31/29, 41/39, 51/49), or 0: any k, kv of the call.  The
// count and graph loops test a dozen k-dependent uniform conditions per window (which words a window spans, which masks apply); with k
// known they fold away — as run-time values they cost scalar registers the kernel does not have (a quarter of the count loops'
// vector instructions were v_readlane reloads of spilled scalars)
// PIPE: the launch is the pipeline's (no N masks, no count-only output); FIT: every pool fits its share of the LDS (the host
// checked) — the per-window tests of those cases fold away with the rest
// the static LDS of a workgroup (one set whatever the number of (k, kv) bodies a kernel runs one after the other)
template <int NT>
struct AsmShared {
    unsigned long long seq[2];          // [0] bases to emit  [1] global base offset
    unsigned long long arcv[NT / 4];    // error removal: the arc that enters a removed head
    uint32_t cnt[8];                    // [0] survivors [1] emitted walks [2] contig base [3] error [4] distinct k-mers [5] nodes [6] LDS table overflow
    uint32_t gap, cand, narc, jflag[3];
    uint32_t erq[NT / 4], arco[NT / 4]; // error removal: candidate heads of a round, removed heads
    uint32_t scan[NT / 64];
};
template <bool W, int NT, int KC, bool PIPE, bool FIT, bool ONE = false>   // FIT: every pool fits its share of the LDS; ONE: the caller's gap only
__device__ __forceinline__ void assemble_body(const AsmParams& P, AsmShared<NT>& sh, const uint32_t one_gap = 0) {
    static_assert(KC == 0 || (KC > 32) == W, "W = k > 32");
    const uint32_t* const nmask = PIPE ? nullptr : P.nmask;
    uint64_t* const cnt_keys = PIPE ? nullptr : P.cnt_keys;   // (4 waves per SIMD = 16 per CU for every NT: <= 128 VGPRs)
    constexpr uint32_t ASM_THREADS = NT;
    uint32_t (&s_cnt)[8] = sh.cnt;
    unsigned long long (&s_seq)[2] = sh.seq;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t PK = KC ? (uint32_t)KC : P.k, PKV = KC ? (uint32_t)(KC - 2) : P.kv;
    const uint32_t npos = P.read_len - PK + 1;
    const int k = (int)PK, kv = (int)PKV;
    const uint32_t per = PK - PKV + 1;

    // gaps are handed out through one counter: pool sizes differ, a static stride left CUs idle behind the largest gaps
    uint32_t &s_gap = sh.gap, &s_cand = sh.cand, &s_narc = sh.narc;
    constexpr uint32_t ERQ = NT / 4;      // candidate heads / removed heads of an error-removal round kept in LDS (more: the pair workspace)
    uint32_t (&s_erq)[ERQ] = sh.erq;
    uint32_t (&s_arco)[ERQ] = sh.arco;
    uint32_t (&s_jflag)[3] = sh.jflag;
    unsigned long long (&s_arcv)[ERQ] = sh.arcv;
    uint32_t (&s_scan)[ASM_THREADS / 64] = sh.scan;
    for (bool first = true;; first = false) {
        __syncthreads();
        uint32_t g;
        if (ONE) {
            if (!first) break;
            g = one_gap;
        } else {
            if (tid == 0) s_gap = atomicAdd(P.next_gap, 1u);
            __syncthreads();
            g = s_gap;
            if (P.gap_list) {
                if (g >= *P.n_gap_list) break;
                g = P.gap_list[g];
            } else if (g >= P.n_pools) break;
        }
        const uint64_t r0 = P.pool_off[g], r1 = P.pool_off[g + 1];
        if (r1 < r0 || r1 > P.total_reads) {   // pool_off beyond the pool array (an overflowed gf_build_pools_dev): refuse the gap
            if (tid == 0) P.gap_error[g] = ASM_ERR_IDS;
            continue;
        }
        const uint32_t n_r = (uint32_t)(r1 - r0);
        if (n_r == 0) continue;
        // workspace unit: every k-mer AND every kv-mer of the gap is a window of one of its reads, so
        // n_r * (L - kv + 1) bounds the distinct k-mers, the nodes and every list below
        const uint32_t unit = cnt_keys ? npos : P.read_len - PKV + 1;
        const uint64_t n_unit64 = (uint64_t)n_r * unit;
        // workspace slice: per pool row (slice_rows == 0), or one private slice per workgroup that every gap it takes re-uses
        // (kernels leave their slice EMPTY) — sized by the caller's bound on the rows of one pool
        if (P.slice_rows && n_r > P.accept_rows) {
            if (tid == 0) {
                if (P.big_list) P.big_list[atomicAdd(P.n_big, 1u)] = g;   // (at most n_pools entries)
                else P.gap_error[g] |= ASM_ERR_IDS;
            }
            continue;
        }
        const uint64_t inst_off = P.slice_rows ? P.slice_base + (uint64_t)blockIdx.x * P.slice_stride : r0 * unit;
        if (n_unit64 >= (1ull << 30) || n_r >= (1u << (31 - INST_OFF_BITS)) - 1) {  // ids are 31-bit: read << 10 | offset (bit 31: INST_WEAK)
            if (tid == 0) P.gap_error[g] = ASM_ERR_IDS;
            continue;
        }
        const uint32_t n_inst = (uint32_t)((uint64_t)n_r * npos);   // k-mer positions
        const uint32_t n_unit = (uint32_t)n_unit64;
        unsigned long long* gtab = P.table + 4 * inst_off;   // global slice, 4 * n_unit slots, all EMPTY on entry
        const uint32_t gcap = 4 * n_unit;
        uint32_t* surv = P.surv + 2 * inst_off;
        uint32_t* list_a = surv;           // slots of the distinct k-mers, later of the nodes
        uint32_t* list_b = surv + n_unit;  // surviving instances, later node instance ids (global mode) + walk records

        // ---- stage the pool
        PoolView V;
        V.rb = P.rb; V.L = P.read_len;
        V.g32 = P.reads32; V.g_words = P.n_words; V.g_tail = P.tail_bytes;
        V.first_byte = r0 * P.rb;
        const uint64_t pool_bytes = (uint64_t)n_r * P.rb;
        V.lds = FIT ? true : pool_bytes + 32 <= (uint64_t)P.lds_words * 4 / (W ? 2 : 3);  // at most a third of the LDS (half when the count table is global anyway)
        uint32_t pool_words = 0;
        if (V.lds) {
            const uint64_t w0 = V.first_byte >> 2;
            const uint32_t sh = (uint32_t)(V.first_byte & 3) * 8;
            const uint32_t nw = (uint32_t)((pool_bytes + 3) / 4);
            pool_words = (nw + 8) & ~1u;   // keeps region R 8-byte aligned
            for (uint32_t i = tid; i < pool_words; i += ASM_THREADS) {
                uint32_t v = 0;
                if (i < nw) {
                    const uint32_t a = asm_word(P.reads32, P.n_words, P.tail_bytes, w0 + i);
                    const uint32_t b = sh ? asm_word(P.reads32, P.n_words, P.tail_bytes, w0 + i + 1) : 0;
                    v = sh ? (a >> sh) | (b << (32 - sh)) : a;
                    if (i == nw - 1 && (pool_bytes & 3)) v &= (1u << ((pool_bytes & 3) * 8)) - 1;
                }
                g_lds[i] = v;
            }
        }
        uint32_t R = pool_words;   // word offset of region R in the dynamic LDS
        uint32_t r_words = P.lds_words - pool_words;   // (both change once: when the graph phase takes the staged pool's place, below)
        if (tid < 8) s_cnt[tid] = 0;
        if (tid < 2) s_seq[tid] = 0;
        __syncthreads();
        ASM_STAMP(0);

        // ---- P1: count canonical k-mers; remember each distinct k-mer's slot.  Optimistic LDS table first.
        Tab tab;
        tab.g = gtab;
        bool keyslot = false, keyslot_w = false, fpslot_used = false;
        uint32_t* dist_inst = P.nodes + 3 * inst_off;   // key-slot mode: instance id of the q-th distinct k-mer
        // Pre-count (min_count 2): one bit array per occurrence level in LDS; a k-mer's bit climbs one level per occurrence
        // and only k-mers whose bit reached the last level are counted exactly afterwards.  At ~1 % read errors half of all
        // windows are k-mers seen once — they never enter the table, which then fits the LDS at a low load (a shared bit only
        // lets a k-mer through to the exact count: no k-mer is lost).  The last level sits at R, the table behind it; the lower
        // levels are dead once the bits are up and lie under the table.
        bool pre_built = false;
        uint32_t pre_log2 = 0;
        bool deferred = false;
        for (int attempt = 0; attempt < 2; ++attempt) {
            const bool use_lds = attempt == 0 && r_words / 2 >= 1024;
            if (attempt == 0 && !use_lds) continue;
            // no room in this launch's share of the LDS (foreseen below, or the table ran full): with two gaps per CU the gap is handed to
            // the launch that gives it a whole CU's LDS instead of being counted in the global slice (several times as slow; nothing
            // global but the workgroup's own lists was touched)
            if (attempt == 1 && P.defer) { deferred = true; break; }
            tab.lds = use_lds;
            tab.off = R;
            tab.cap = use_lds ? r_words / 2 : gcap;
            // Key-slot mode (k <= 31, min_count <= 2, assembly): a slot holds the 62-bit canonical key itself plus a 2-bit
            // saturating count in the spare low bits.  A repeat occurrence — most instances at
            // sequencing depth — costs one 8-byte read and a compare: no key re-derivation from the reads, and no atomic
            // once the count has saturated.  The instance id of each distinct k-mer goes to a side list (the node arrays
            // are idle here).
            keyslot = P.keyslot && !W && k <= 31 && P.min_count <= 2 && !cnt_keys;   // (min_count <= 2: the 2-bit field tells count <= min_count + 1 from more)
            // wide variant (32 < k <= 63): a 16-byte slot = {hi, ~(lo | count)}, see count_keyslot_wide below.  Only for the
            // global table: in LDS the doubled slot size overflows pools that the 8-byte instance-id slots still hold
            // (measured at k=41, 214-read pools: 161 us against 105 us per gap).
            const bool wide_ok = P.keyslot && W && k > 32 && k <= 62 && P.min_count <= 2 && !cnt_keys;   // (k <= 62: three spare bits in the 16-byte slot)
            keyslot_w = wide_ok && !use_lds;
            // fingerprint slots (32 < k <= 63 in an LDS table over an LDS-staged pool): slot = instance id | 30-bit key fingerprint
            // << 32 | 2-bit saturating count << 62.  A probe that meets another key sees it in the fingerprint (no re-derivation of
            // the occupant's key: at 1 % errors 60 % of the slots fill up and every collision cost a 128-bit fetch + reverse
            // complement); an equal fingerprint is verified EXACTLY by comparing the occupant's raw window with this window and
            // its reverse complement; once the count is saturated a repeat costs no atomic.  (PMC: 900 wave instructions per k-mer
            // with instance-id slots at k = 51.)
            const bool fpslot = wide_ok && use_lds && V.lds;
            fpslot_used = fpslot;
            // an LDS attempt that is bound to overflow is skipped: at ~1 % errors nearly half of all windows are distinct
            const uint32_t levels = P.min_count;
            bool pre = P.precount && (keyslot || keyslot_w || fpslot) && levels == 2;
            // (an LDS table that holds every key of a k <= 31 pool is filled faster without: see count_keyslot_strided)
            if (keyslot && use_lds && n_inst / 4 <= (r_words / 2) - (r_words / 2) / 4) pre = false;
            if (pre && !pre_built) {   // 8 bits per window when they fit: half of the LDS region under an LDS table, all of it otherwise
                uint32_t lg = 11;
                while ((1u << lg) < 8 * n_inst && lg < 22) ++lg;
                const uint32_t maxw = (use_lds ? (uint32_t)((uint64_t)r_words * P.pre_frac8 / 8) : r_words) / levels;
                while (lg > 11 && (1u << (lg - 5)) > maxw) --lg;
                pre = (1u << (lg - 5)) <= maxw;
                pre_log2 = lg;
            }
            const uint32_t pre_words = pre ? 1u << (pre_log2 - 5) : 0u;
            if (pre && use_lds) {
                tab.off = R + pre_words;
                tab.cap = (r_words - pre_words) / 2;
            }
            const uint32_t limit = use_lds ? tab.cap - tab.cap / 4 : 0xFFFFFFFFu;
            const uint32_t limit_k = limit;
            // (pre-counted: true k-mers + the few that share a bit; measured 3 000 of 66 000 windows at 30x, 1 % errors)
            if (use_lds && (keyslot || wide_ok) && n_inst / (pre ? 12 : 4) > limit) continue;
            auto pre_pass = [&](uint64_t x) -> bool {
                const uint32_t b = (uint32_t)(x >> 32) >> (32 - pre_log2);
                return (g_lds[R + (b >> 5)] >> (b & 31)) & 1u;
            };
            // ---- window generator of the three table modes below: a thread takes S consecutive windows of one read, derives the
            //      first from the staged bytes (two unaligned fetches, byte swaps, a 128-bit reverse complement) and ROLLS the others:
            //      one base shifted into the forward strand and its complement into the reverse strand.  (PMC before: 580 vector
            //      instructions per window over the whole kernel, most of them this derivation — the count phase was issue-bound,
            //      not table-bound.)  S is picked per gap so that the items fill whole rounds of the 1024 threads.
            const bool rolled = keyslot || keyslot_w || fpslot;
            uint32_t S = 8;
            if (rolled) {
                uint32_t best = 0xFFFFFFFFu;
                for (uint32_t c = 6; c <= 16; ++c) {
                    const uint32_t rounds = (n_r * ((npos + c - 1) / c) + ASM_THREADS - 1) / ASM_THREADS;
                    const uint32_t cost = rounds * (c + 3);   // the first window of an item costs about three rolled ones
                    if (cost < best) { best = cost; S = c; }
                }
            }
            const uint32_t chunks = (npos + S - 1) / S, n_items = n_r * chunks;
            // body(read, offset, canonical key, forward, reverse complement, hint) -> true = call me again with the same window.
            // hint(canonical key) runs one window AHEAD of body: it issues the first LDS reads of a window's look-up (the word of its pre-count
            // bit, the prefix of that word) before the body of the window in front works through its own chain of dependent LDS accesses —
            // table slot, CAS —, so a window's chain is two round trips shorter (the exact pass was 3.2 x the bit-array pass per window,
            // and all of the difference was waiting: PMC, §4)
            auto for_windows = [&](auto&& body, auto&& hint, auto ahead_c) {
                constexpr bool AHEAD = decltype(ahead_c)::value;   // false: hint and body of a window together (the bit-array pass: its chain is short, the second window's state only costs registers)
                const uint32_t dr = ASM_THREADS / chunks, dc = ASM_THREADS - dr * chunks;
                uint32_t r = tid / chunks, c = tid - r * chunks;
                for (uint32_t item = tid; item < n_items; item += ASM_THREADS, r += dr, c += dc) {
                    if (c >= chunks) { c -= chunks; ++r; }
                    if (use_lds && s_cnt[6]) break;
                    uint32_t p = c * S;
                    const uint32_t pe = p + S < npos ? p + S : npos;
                    K128 fw = pv_kmer_at<W>(V, r, p, k);
                    K128 rc = revcomp_w<W>(fw, k);
                    uint32_t nxt = pv_stream32(V, (uint64_t)r * V.rb * 8 + 2ull * (p + PK));   // the bases behind the window
                    K128 key = rc < fw ? rc : fw;
                    if constexpr (!AHEAD) {
                        for (;;) {
                            bool ok = true;
                            if (nmask) ok = !window_masked(nmask + (r0 + r) * P.nmw, P.nmw, p, PK);
                            if (ok) {
                                key = rc < fw ? rc : fw;
                                const auto h0 = hint(key);
                                while (body(r, p, key, fw, rc, h0)) {}
                            }
                            if (++p >= pe) break;
                            const uint64_t b = nxt >> 30;
                            nxt <<= 2;
                            if constexpr (W) {   // k > 32: the new base lands in the low word
                                const int sh = 128 - 2 * k;
                                fw.hi = (fw.hi << 2) | (fw.lo >> 62);
                                fw.lo = (fw.lo << 2) | (b << sh);
                                rc.lo = ((rc.lo >> 2) | (rc.hi << 62)) & (~0ull << sh);
                                rc.hi = (rc.hi >> 2) | ((3ull - b) << 62);
                            } else {
                                const int sh = 64 - 2 * k;
                                fw.hi = (fw.hi << 2) | (b << sh);
                                rc.hi = ((rc.hi >> 2) | ((3ull - b) << 62)) & (~0ull << sh);
                            }
                        }
                        continue;
                    }
                    auto h = hint(key);
                    for (;;) {
                        // the next window: rolled, canonical, hinted
                        const bool more = p + 1 < pe;
                        K128 fw1 = fw, rc1 = rc;
                        {
                            const uint64_t b = nxt >> 30;
                            nxt <<= 2;
                            if constexpr (W) {   // k > 32: the new base lands in the low word
                                const int sh = 128 - 2 * k;
                                fw1.hi = (fw.hi << 2) | (fw.lo >> 62);
                                fw1.lo = (fw.lo << 2) | (b << sh);
                                rc1.lo = ((rc.lo >> 2) | (rc.hi << 62)) & (~0ull << sh);
                                rc1.hi = (rc.hi >> 2) | ((3ull - b) << 62);
                            } else {
                                const int sh = 64 - 2 * k;
                                fw1.hi = (fw.hi << 2) | (b << sh);
                                rc1.hi = ((rc.hi >> 2) | ((3ull - b) << 62)) & (~0ull << sh);
                            }
                        }
                        const K128 key1 = rc1 < fw1 ? rc1 : fw1;
                        auto h1 = h;
                        if (more) h1 = hint(key1);
                        bool ok = true;
                        if (nmask) ok = !window_masked(nmask + (r0 + r) * P.nmw, P.nmw, p, PK);
                        if (ok) {
                            while (body(r, p, key, fw, rc, h)) {}
                        }
                        if (!more) break;
                        ++p;
                        fw = fw1; rc = rc1; key = key1; h = h1;
                    }
                }
            };
            struct WinHint { uint64_t x; uint32_t word, pfx; };   // the window's hash (computed once, here), then what was read ahead
            auto no_hint = [](const K128& key) -> WinHint { return WinHint{hash_p1<W>(key), 0u, 0u}; };
            // (word of the k-mer's pre-count bit in the last level, and — ranked tables — the prefix of that word)
            auto fin_hint = [&](const K128& key) -> WinHint {
                const uint64_t x = hash_p1<W>(key);
                const uint32_t b = (uint32_t)(x >> 32) >> (32 - pre_log2);
                return WinHint{x, g_lds[R + (b >> 5)], 0u};
            };
            auto rank_hint = [&](const K128& key) -> WinHint {
                const uint64_t x = hash_p1<W>(key);
                const uint32_t b = (uint32_t)(x >> 32) >> (32 - pre_log2);
                return WinHint{x, g_lds[R + (b >> 5)], (uint32_t)reinterpret_cast<const uint16_t*>(&g_lds[R + pre_words])[b >> 5]};
            };
            if (pre && !pre_built) {
                for (uint32_t i = tid; i < levels * pre_words; i += ASM_THREADS) g_lds[R + i] = 0;
                __syncthreads();
                uint32_t* fin = &g_lds[R];
                uint32_t* up1 = fin + pre_words;
                uint32_t* up2 = up1 + pre_words;
                for_windows([&](uint32_t, uint32_t, const K128&, const K128&, const K128&, const WinHint& hw) -> bool {
                    const uint32_t b = (uint32_t)(hw.x >> 32) >> (32 - pre_log2);
                    const uint32_t w = b >> 5, m = 1u << (b & 31);
                    if (hw.word & m) return false;                      // already up: most windows at sequencing depth (a bit that went up since the hint was taken only costs the steps below: they are idempotent)
                    if (!(atomicOr(&up1[w], m) & m)) return false;      // first occurrence
                    if (levels == 3 && !(atomicOr(&up2[w], m) & m)) return false;
                    atomicOr(&fin[w], m);
                    return false;
                }, fin_hint, std::false_type{});
                __syncthreads();
                ASM_STAMP(8);
                pre_built = true;
            }
            // Ranked table (fingerprint slots behind a pre-count): the set bits of the last level ARE a perfect hash of the k-mers that
            // get counted, up to the few that share a bit — a k-mer's slot is the rank of its bit (prefix of set bits per word + a
            // popcount), so a look-up is ONE slot, no probe sequence (with linear probing a second probe happened in 7 % of the
            // look-ups but in 99 % of the 64-lane steps, and every extra round costs the whole wave an LDS round trip).  A k-mer that
            // finds another one on its bit goes to a small hashed overflow region behind the ranked slots.
            bool ranked = false;
            uint32_t n_set = 0, ovf_cap = 0;
            if ((fpslot || keyslot) && pre && use_lds && P.ranked) {
                uint32_t* fin = &g_lds[R];
                uint16_t* pfx = reinterpret_cast<uint16_t*>(fin + pre_words);   // over the lower levels (dead now); 16 bits: < 65 536 counted k-mers
                const uint32_t cw = (pre_words + ASM_THREADS - 1) / ASM_THREADS;
                const uint32_t w0 = tid * cw < pre_words ? tid * cw : pre_words, w1 = w0 + cw < pre_words ? w0 + cw : pre_words;
                uint32_t local = 0;
                for (uint32_t w = w0; w < w1; ++w) local += (uint32_t)__popc(fin[w]);
                uint32_t inc = local;
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t y = __shfl_up(inc, d);
                    if ((int)lane >= d) inc += y;
                }
                if (lane == 63) s_scan[tid >> 6] = inc;
                __syncthreads();
                uint32_t wave_off = 0, total = 0;
                for (uint32_t v = 0; v < ASM_THREADS / 64; ++v) { const uint32_t t = s_scan[v]; if (v < (tid >> 6)) wave_off += t; total += t; }
                uint32_t run = wave_off + inc - local;
                for (uint32_t w = w0; w < w1; ++w) { const uint32_t c = (uint32_t)__popc(fin[w]); pfx[w] = (uint16_t)run; run += c; }
                n_set = total;
                // overflow region: as many slots as ranked ones when they fit (deep pools leave ~4 bits per window: a quarter of the
                // counted k-mers share a bit), at least half as many
                const uint64_t head = (uint64_t)pre_words + pre_words / 2;   // last level + prefixes
                ovf_cap = n_set > 256 ? n_set : 256;
                if (head + 2 * ((uint64_t)n_set + ovf_cap) > r_words) ovf_cap = n_set / 2 > 256 ? n_set / 2 : 256;
                ranked = n_set < 65536 && head + 2 * ((uint64_t)n_set + ovf_cap) <= r_words;
                if (ranked) {
                    tab.off = R + pre_words + pre_words / 2;
                    tab.cap = n_set + ovf_cap;
                }
                __syncthreads();
                // The set bits also say how many nodes the graph will have (k-mers that share a bit and k-mers that reach a bit without
                // reaching min_count about cancel: the survivors are 0.9-1.0 of the set bits): when no LDS plan of this launch holds that many
                // (plan 2: 5.5 words per node) the gap would build, link, clean and rank its graph in the global slice — three to six times
                // as slow per phase, and its traffic slows the neighbours — so it goes to the launch with a whole CU's LDS now, a tenth of
                // its work done.
                // (The first ASM_DEFER_STAY such gaps of a launch stay: a handful of them — C4 has eight — is not worth a further launch, which
                //  lasts as long as its slowest gap at least.)
                if (P.defer && n_set + n_set / 32 + 64 > 2 * r_words / 11) {
                    if (tid == 0) s_cand = atomicAdd(P.defer_tickets, 1u);
                    __syncthreads();
                    const bool leave = s_cand >= ASM_DEFER_STAY;
                    __syncthreads();
                    if (leave) { deferred = true; break; }
                }
            }
#ifndef GF_KS_COMPLEMENT_LDS
#define GF_KS_COMPLEMENT_LDS 0
#endif
            // the global slice must keep its EMPTY64 pattern, so key|count is stored complemented there; an LDS table is
            // private to the gap and stores it plain with an all-ones EMPTY
            const unsigned long long xm = (!use_lds || GF_KS_COMPLEMENT_LDS) ? ~0ull : 0ull;
            const unsigned long long kempty = xm ? EMPTY64 : ~0ull;
            if (use_lds) {
                const unsigned long long e = keyslot ? kempty : EMPTY64;   // (the wide variant keeps the EMPTY64 words)
                for (uint32_t i = tid; i < tab.cap; i += ASM_THREADS) tab.store(i, e);
                __syncthreads();
                ASM_STAMP(9);
            }
            if (keyslot) {
                // one instantiation per home of the table, so that every access compiles to ds_* or global_* without a branch
                auto count_keyslot = [&](auto lds_c) {
                constexpr bool LDS = decltype(lds_c)::value;
                Tab t = tab;
                t.lds = LDS;
                auto keyslot_body = [&](uint32_t r, uint32_t p, const K128& key, const K128&, const K128&, const WinHint& hw) -> bool {
                    const unsigned long long keyhi = key.hi;
                    const uint64_t x = hw.x;
                    uint32_t lo_sl = 0, n_sl = t.cap;                                  // probe region [lo_sl, lo_sl + n_sl)
                    if (LDS && ranked) {   // the slot of the k-mer's pre-count bit first (see `ranked`), the hashed overflow region behind it
                        const uint32_t b = (uint32_t)(x >> 32) >> (32 - pre_log2);
                        const uint32_t word = hw.word;
                        if (!((word >> (b & 31)) & 1u)) return false;
                        const uint32_t rnk = hw.pfx + (uint32_t)__popc(word & ((1u << (b & 31)) - 1u));
                        unsigned long long v = t.load(rnk);
                        if (v == kempty) {
                            v = t.cas(rnk, kempty, keyhi ^ xm);
                            if (v == kempty) {   // first occurrence
                                const uint32_t q = atomicAdd(&s_cnt[4], 1u);
                                if (q < n_unit) { list_a[q] = rnk; dist_inst[q] = make_inst(r, p); }
                                return false;
                            }
                        }
                        if (((v ^ xm) & ~3ull) == keyhi) {
                            while (((v ^ xm) & 3ull) != 3ull) {   // saturating increment (the field holds count - 1: saturates at 4 occurrences)
                                const unsigned long long o = t.cas(rnk, v, ((v ^ xm) + 1) ^ xm);
                                if (o == v) break;
                                v = o;
                            }
                            return false;
                        }
                        lo_sl = n_set; n_sl = ovf_cap;
                    } else if (pre && !pre_pass(x)) return false;
                    uint32_t sl = lo_sl + slot_of_hash(x, n_sl);
                    // insert / count: every decision is confirmed by a CAS, whose result replaces the loaded value
                    bool placed = false;
                    for (uint32_t probes = 0; probes < n_sl; ++probes) {
                        unsigned long long v = t.load(sl);
                        if (v == kempty) {
                            v = t.cas(sl, kempty, keyhi ^ xm);
                            if (v == kempty) {   // first occurrence
                                const uint32_t q = atomicAdd(&s_cnt[4], 1u);
                                if (LDS && ranked) { if (atomicAdd(&s_cnt[7], 1u) >= ovf_cap - ovf_cap / 4) s_cnt[6] = 1; }
                                else if (q >= limit) s_cnt[6] = 1;
                                if (q < n_unit) { list_a[q] = sl; dist_inst[q] = make_inst(r, p); }
                                placed = true;
                                break;
                            }
                        }
                        if (((v ^ xm) & ~3ull) == keyhi) {
                            while (((v ^ xm) & 3ull) != 3ull) {   // saturating increment (the field holds count - 1: saturates at 4 occurrences)
                                const unsigned long long o = t.cas(sl, v, ((v ^ xm) + 1) ^ xm);
                                if (o == v) break;
                                v = o;
                            }
                            placed = true;
                            break;
                        }
                        sl = sl + 1 == lo_sl + n_sl ? lo_sl : sl + 1;
                    }
                    if (!placed) { if (LDS) s_cnt[6] = 1; else atomicOr(&s_cnt[3], ASM_ERR_KTABLE); }
                    return false;
                };
                if (LDS && ranked) for_windows(keyslot_body, rank_hint, std::true_type{}); else for_windows(keyslot_body, no_hint, std::false_type{});
                };
                auto count_keyslot_strided = [&](auto lds_c) {
                constexpr bool LDS = decltype(lds_c)::value;
                Tab t = tab;
                t.lds = LDS;
                // positions are strided over the threads; (read, offset) advance incrementally (no division per k-mer).
                // (Measured and dropped: rolling the k-mers along per-thread runs — fewer instructions per k-mer but longer
                // runs per thread; 59 vs 48 us per gap, the phase is bound by the dependent LDS table accesses.)
                const uint32_t dr = ASM_THREADS / npos, dp = ASM_THREADS - dr * npos;
                uint32_t r = tid / npos, p = tid - r * npos;
                // the k-mer at (read rr, offset pp), or valid = false when the N mask covers it
                auto kmer_of = [&](uint32_t rr, uint32_t pp, bool& valid) -> unsigned long long {
                    const unsigned long long fw = pv_kmer_at<false>(V, rr, pp, k).hi;
                    const unsigned long long rc = revcomp_w<false>(K128{fw, 0}, k).hi;
                    valid = true;
                    if (nmask) valid = !window_masked(nmask + (r0 + rr) * P.nmw, P.nmw, pp, PK);
                    return fw < rc ? fw : rc;
                };
                // insert / count one k-mer; v = the slot's content as loaded by the caller (may be stale: every decision below is
                // confirmed by a CAS, whose result replaces it)
                auto upsert = [&](unsigned long long keyhi, uint32_t sl, unsigned long long v, uint32_t inst) {
                    bool placed = false;
                    for (uint32_t probes = 0; probes < t.cap; ++probes) {
                        if (probes) v = t.load(sl);
                        if (v == kempty) {
                            v = t.cas(sl, kempty, keyhi ^ xm);
                            if (v == kempty) {   // first occurrence
                                const uint32_t q = atomicAdd(&s_cnt[4], 1u);
                                if (q >= limit) s_cnt[6] = 1;
                                if (q < n_unit) { list_a[q] = sl; dist_inst[q] = inst; }
                                placed = true;
                                break;
                            }
                        }
                        if (((v ^ xm) & ~3ull) == keyhi) {
                            while (((v ^ xm) & 3ull) != 3ull) {   // saturating increment (the field holds count - 1: saturates at 4 occurrences)
                                const unsigned long long o = t.cas(sl, v, ((v ^ xm) + 1) ^ xm);
                                if (o == v) break;
                                v = o;
                            }
                            placed = true;
                            break;
                        }
                        sl = sl + 1 == t.cap ? 0 : sl + 1;
                    }
                    if (!placed) { if (LDS) s_cnt[6] = 1; else atomicOr(&s_cnt[3], ASM_ERR_KTABLE); }
                };
                // KU k-mers per thread and round: their first probes are in flight together (the phase is bound by the latency of
                // its dependent table accesses, not by their number)
                constexpr int KU = 2;   // four measured 36.0 vs 37.3 us per gap but spills registers
                for (uint32_t inst_i = tid; inst_i < n_inst; inst_i += KU * ASM_THREADS) {
                    if (LDS && s_cnt[6]) break;
                    unsigned long long kk[KU], vv[KU];
                    uint32_t ss[KU], ii[KU];
                    bool ok[KU];
#pragma unroll
                    for (int u = 0; u < KU; ++u) {
                        if (p >= npos) { p -= npos; ++r; }
                        ok[u] = false;
                        kk[u] = 0;
                        if (inst_i + u * ASM_THREADS < n_inst) kk[u] = kmer_of(r, p, ok[u]);
                        ii[u] = make_inst(r, p);
                        ss[u] = slot_of(K128{kk[u], 0}, t.cap);
                        r += dr; p += dp;
                    }
#pragma unroll
                    for (int u = 0; u < KU; ++u) vv[u] = t.load(ss[u]);
#pragma unroll
                    for (int u = 0; u < KU; ++u)
                        if (ok[u]) upsert(kk[u], ss[u], vv[u], ii[u]);
                }
                };
                // an LDS table holds every key of a k <= 31 pool that gets here: strided windows with two table probes in flight beat
                // the rolled generator there (58 against 80 us per 320-read pool at k = 31); the global table is the other way round
                if (use_lds && !pre) count_keyslot_strided(std::true_type{});
                else if (use_lds) count_keyslot(std::true_type{});
                else count_keyslot(std::false_type{});
            } else if (keyslot_w) {
                // Wide key-slot: word 0 = key.hi, word 1 = ~(key.lo | count) (count in the >= 2 spare low bits of lo; the
                // complement keeps data words apart from EMPTY64 and from the LOCK word, whose low two bits are 11).  A slot is
                // claimed by CAS(word 1: EMPTY64 -> LOCK), then hi is written, then word 1 is published.
                auto count_keyslot_wide = [&](auto lds_c) {
                constexpr bool LDS = decltype(lds_c)::value;
                constexpr unsigned long long LOCK = 0x80000000FFFFFFFFull;
                Tab t = tab;
                t.lds = LDS;
                const uint32_t wcap = t.cap / 2;
                auto put = [&](uint32_t word, unsigned long long v) {
                    if (LDS) __hip_atomic_store(t.l() + word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_store(t.g + word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                };
                // A thread that meets a LOCKed slot does NOT spin: the owner's publishing stores sit on the exit path of the
                // probe loop, which the SIMT control flow runs only after every lane of the wave has left that loop — a lane
                // spinning inside it would wait for an owner in its own wave forever.  It leaves the probe loop and is called
                // again with the same window, by which time an owner in its wave has published.
                for_windows([&](uint32_t r, uint32_t p, const K128& key, const K128&, const K128&, const WinHint& hw) -> bool {
                    const uint64_t x = hw.x;
                    if (pre && !pre_pass(x)) return false;
                    const uint32_t inst = make_inst(r, p);
                    bool placed = false, retry = false;
                    uint32_t sl = slot_of_hash(x, wcap);
                    for (uint32_t probes = 0; probes < wcap; ++probes) {
                        // global table: both words of the slot in ONE 16-byte request (word 0 is in L2 before word 1 is
                        // published, so a load that sees word 1 published sees word 0 too)
                        unsigned long long b, a_pre = 0;
                        bool a_valid = false;
                        if (LDS) b = t.load(2 * sl + 1);
                        else {   // sc0 sc1: from L2, like the atomics (the CU's L1 may hold the line as it was before a CAS)
                            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                            u64x2 q;
                            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(t.g + 2 * sl) : "memory");
                            a_pre = q.x; b = q.y;
                            a_valid = true;
                        }
                        if (b == EMPTY64) {
                            a_valid = false;   // b will come from the CAS: word 0 must be read again
                            b = t.cas(2 * sl + 1, EMPTY64, LOCK);
                            if (b == EMPTY64) {   // first occurrence: this thread owns the slot
                                put(2 * sl, key.hi);
                                // word 0 must be in place before word 1 says so.  LDS executes a wave's operations in order; a
                                // global store is acknowledged (vmcnt) once it is in L2, where every access of this table
                                // goes — a full agent-scope release (L2 write-back) here cost 5x the whole phase
                                if (LDS) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                put(2 * sl + 1, ~(key.lo | 1ull));   // (3-bit count field, 1..4)
                                const uint32_t q = atomicAdd(&s_cnt[4], 1u);
                                if (q >= limit_k) s_cnt[6] = 1;
                                if (q < n_unit) { list_a[q] = sl; dist_inst[q] = inst; }
                                placed = true;
                                break;
                            }
                        }
                        if (b == LOCK) { retry = true; break; }
                        asm volatile("" ::: "memory");          // word 0 is read after word 1 was seen published
                        const unsigned long long a = a_valid ? a_pre : t.load(2 * sl);
                        if (a == key.hi && (~b & ~7ull) == key.lo) {
                            while ((~b & 7ull) != 4ull) {   // saturating increment of the complemented count
                                const unsigned long long o = t.cas(2 * sl + 1, b, b - 1);
                                if (o == b) break;
                                b = o;
                            }
                            placed = true;
                            break;
                        }
                        sl = sl + 1 == wcap ? 0 : sl + 1;
                    }
                    if (retry) return true;
                    if (!placed) { if (LDS) s_cnt[6] = 1; else atomicOr(&s_cnt[3], ASM_ERR_KTABLE); }
                    return false;
                }, no_hint, std::false_type{});
                };
                count_keyslot_wide(std::false_type{});   // (keyslot_w implies the global table: no LDS instantiation — the kernel is at the edge of its registers and every unused path costs the used ones)
            } else if (fpslot) {
                auto fpslot_body = [&](uint32_t r, uint32_t p, const K128&, const K128& fw, const K128& rc, const WinHint& hw) -> bool {
                    const uint64_t x = hw.x;
                    const unsigned long long fp = (x >> 34) << 32;                    // 30 bits, in place
                    const unsigned long long mine = fp | make_inst(r, p);   // (count - 1 = 0 in bits 62..63)
                    uint32_t sl, lo_sl = 0, n_sl = tab.cap;                            // probe region [lo_sl, lo_sl + n_sl)
                    if (ranked) {
                        const uint32_t b = (uint32_t)(x >> 32) >> (32 - pre_log2);
                        const uint32_t word = hw.word;
                        if (!((word >> (b & 31)) & 1u)) return false;
                        const uint32_t rnk = hw.pfx + (uint32_t)__popc(word & ((1u << (b & 31)) - 1u));
                        unsigned long long v = tab.load(rnk);
                        if ((uint32_t)v == EMPTY32) {
                            v = tab.cas(rnk, EMPTY64, mine);
                            if (v == EMPTY64) {   // first occurrence
                                const uint32_t q = atomicAdd(&s_cnt[4], 1u);
                                if (q < n_unit) list_a[q] = rnk;
                                return false;
                            }
                        }
                        if ((v & 0x3FFFFFFF00000000ull) == fp) {
                            const uint32_t oi = (uint32_t)v;
                            const K128 ow = lds_window<true>(V.rb, oi >> INST_OFF_BITS, oi & INST_OFF_MASK, k);
                            if (ow == fw || ow == rc) {   // the same canonical k-mer
                                while ((v >> 62) != 3ull) {   // saturating increment
                                    const unsigned long long o = tab.cas(rnk, v, v + (1ull << 62));
                                    if (o == v) break;
                                    v = o;
                                }
                                return false;
                            }
                        }
                        lo_sl = n_set; n_sl = ovf_cap;   // another k-mer owns the bit's slot: hashed overflow region
                    } else {
                        if (pre && !pre_pass(x)) return false;
                    }
                    sl = lo_sl + slot_of_hash(x, n_sl);
                    bool placed = false;
                    for (uint32_t probes = 0; probes < n_sl; ++probes) {
                        unsigned long long v = tab.load(sl);
                        if ((uint32_t)v == EMPTY32) {
                            v = tab.cas(sl, EMPTY64, mine);
                            if (v == EMPTY64) {   // first occurrence
                                const uint32_t q = atomicAdd(&s_cnt[4], 1u);
                                if (ranked) { if (atomicAdd(&s_cnt[7], 1u) >= ovf_cap - ovf_cap / 4) s_cnt[6] = 1; }
                                else if (q >= limit) s_cnt[6] = 1;
                                if (q < n_unit) list_a[q] = sl;
                                placed = true;
                                break;
                            }
                        }
                        if ((v & 0x3FFFFFFF00000000ull) == fp) {
                            const uint32_t oi = (uint32_t)v;
                            const K128 ow = lds_window<true>(V.rb, oi >> INST_OFF_BITS, oi & INST_OFF_MASK, k);
                            if (ow == fw || ow == rc) {   // the same canonical k-mer
                                while ((v >> 62) != 3ull) {   // saturating increment
                                    const unsigned long long o = tab.cas(sl, v, v + (1ull << 62));
                                    if (o == v) break;
                                    v = o;
                                }
                                placed = true;
                                break;
                            }
                        }
                        sl = sl + 1 == lo_sl + n_sl ? lo_sl : sl + 1;
                    }
                    if (!placed) s_cnt[6] = 1;
                    return false;
                };
                if (ranked) for_windows(fpslot_body, rank_hint, std::true_type{}); else for_windows(fpslot_body, no_hint, std::false_type{});
            } else
            for (uint32_t inst_i = tid; inst_i < n_inst; inst_i += ASM_THREADS) {
                if (use_lds && s_cnt[6]) break;
                const uint32_t r = inst_i / npos, p = inst_i - r * npos;
                if (nmask && window_masked(nmask + (r0 + r) * P.nmw, P.nmw, p, PK)) continue;
                const uint32_t inst = make_inst(r, p);
                const K128 key = canonical_w<W>(pv_kmer<W>(V, inst, k), k);
                bool fresh;
                const uint32_t s = table_upsert<W>(tab, V, key, inst, k, 1u, &fresh);
                if (s == EMPTY32) { if (use_lds) s_cnt[6] = 1; else atomicOr(&s_cnt[3], ASM_ERR_KTABLE); continue; }
                if (fresh) {
                    const uint32_t q = atomicAdd(&s_cnt[4], 1u);
                    if (q >= limit) s_cnt[6] = 1;
                    if (q < n_unit) list_a[q] = s;
                }
            }
            wg_phase_sync();
            if (!(use_lds && s_cnt[6])) break;
            // the LDS table got too full: start over in the global slice (nothing global was touched yet)
            __syncthreads();
            if (tid == 0) { s_cnt[4] = 0; s_cnt[6] = 0; }
            __syncthreads();
        }
        if (deferred) {
            if (tid == 0) P.big_list[atomicAdd(P.n_big, 1u)] = g;
            continue;
        }
        const bool tab_global = !tab.lds;
        const uint32_t n_dist = s_cnt[4] < n_unit ? s_cnt[4] : n_unit;
        ASM_STAMP(1);

        // ---- P2: survivors (count >= min_count) -> list_b; a global table gets its used slots reset.  Four items per thread
        //      and round with their (dependent: list -> slot) loads issued together: in the global table this phase is a
        //      chain of L2/fabric round trips
        for (uint32_t i0 = 0; i0 < n_dist; i0 += 4 * ASM_THREADS) {
            uint32_t sl[4], idv[4];
            unsigned long long v[4];
            bool in[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + u * ASM_THREADS + tid;
                in[u] = i < n_dist;
                sl[u] = in[u] ? list_a[i] : 0;
                idv[u] = (in[u] && (keyslot || keyslot_w)) ? dist_inst[i] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = in[u] ? tab.load(keyslot_w ? 2 * sl[u] + 1 : sl[u]) : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                bool keep = false;
                uint32_t id = 0, c = 0;
                if (in[u]) {
                    id = (keyslot || keyslot_w) ? idv[u] : (uint32_t)v[u];
                    // occurrences, saturating at 4 in the slot forms (their 2-bit fields hold count - 1, the wide slot 1..4)
                    c = fpslot_used ? (uint32_t)(v[u] >> 62) + 1u
                        : keyslot_w ? (uint32_t)(~v[u] & 7ull)
                        : keyslot ? (uint32_t)((tab_global ? ~v[u] : (GF_KS_COMPLEMENT_LDS ? ~v[u] : v[u])) & 3ull) + 1u : (uint32_t)(v[u] >> 32);
                    keep = c >= P.min_count;
                    // WEAK: seen at most min_count + 1 times.  c is the true count wherever that matters: the slot forms saturate at 4 and
                    // are used at min_count <= 2 only (4 > min_count + 1), the instance-id slots count in 32 bits.  tiebreak_counts == 0:
                    // the rule is off — nothing Velvet could not have known (cvtFaToFq drops the counts, assemble_gaps.py:56-79)
                    if (!cnt_keys && P.tiebreak_counts && c <= P.min_count + 1) id |= INST_WEAK;
                    if (tab_global) {
                        if (keyslot_w) { tab.store(2 * sl[u], EMPTY64); tab.store(2 * sl[u] + 1, EMPTY64); }
                        else tab.store(sl[u], EMPTY64);
                    }
                }
                const unsigned long long bal = __ballot(keep);
                if (bal) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&s_cnt[0], (uint32_t)__popcll(bal));
                    base = __shfl(base, 0);
                    if (keep) {
                        const uint32_t o = base + __popcll(bal & ((1ull << lane) - 1));
                        list_b[o] = id;
                        if (cnt_keys && o < P.cnt_cap) {
                            const K128 key = canonical_w<W>(pv_kmer<W>(V, id, k), k);
                            cnt_keys[2 * (uint64_t)o] = key.hi;
                            cnt_keys[2 * (uint64_t)o + 1] = key.lo;
                            P.cnt_counts[o] = c > 10000000u ? 10000000u : c;  // kmc -cs10000000 (assemble_gaps.py:96)
                        }
                    }
                }
            }
        }
        wg_phase_sync();
        if (cnt_keys) {
            if (tid == 0) { *P.n_contigs = s_cnt[0]; if (s_cnt[3]) P.gap_error[g] |= s_cnt[3]; }
            __syncthreads();
            continue;
        }
        const uint32_t n_surv = s_cnt[0];
        ASM_STAMP(2);
        if (P.dbg && tid == 0) {   // diagnostics: distinct k-mers counted exactly, table home, survivors, windows
            P.dbg[(uint64_t)g * 16 + 10] = n_dist; P.dbg[(uint64_t)g * 16 + 11] = tab_global; P.dbg[(uint64_t)g * 16 + 12] = n_surv; P.dbg[(uint64_t)g * 16 + 13] = n_inst;
        }

        // ---- graph-phase memory: node table + 4 arrays + the unitig-ranking pairs.  Optimistic LDS plan first: room for `nb`
        //      nodes (11 words each: 4/3 table slots + inst_of/meta/succ0/succ1 + one 8-byte {ancestor, distance} pair per
        //      oriented node); if the gap has more nodes the phase is redone in global memory.  The node table stays alive to the
        //      end: error removal looks neighbours up again after every round.
        const uint64_t node_bound = (uint64_t)per * n_surv < n_unit ? (uint64_t)per * n_surv : n_unit;
        // Three plans: 0 = everything in LDS (9 words per node: 4 array words, 3/2 table slots, one 32-bit pair per oriented node);
        // 1 = node table + arrays in LDS, the pairs in the global slice until error removal is over, then over the node table
        // (7 words per node: gaps with up to 4 900 nodes beside a 720-read pool — C5's pools, whose graph phases all ran in the
        // global slice before: 910 us per gap at k = 51); 2 = everything in the global slice.  The survivors
        // say where to start (a gap has about as many nodes as surviving k-mers); a plan that overflows falls through to the next.
        bool graph_lds = false, j_lds = false;
        // Successor slots: while the nodes are built every edge is at hand with BOTH its slots, so it leaves "a successor of (slot,
        // orientation)" in the pair workspace (idle until the error removal) — for an oriented node with ONE arc out that is the
        // successor, and the links phase reads it instead of finding the shifted kv-mer in the node table again (derive, reverse
        // complement, hash, probe, verify: 34 of a C4 gap's 314 us).  Equal values race when the out-degree is 1; otherwise unused.
        bool cand_on = false;
        uint16_t* const scand = reinterpret_cast<uint16_t*>(P.jump + 4 * inst_off);   // 16 bits per entry: the table of an LDS plan has fewer than 32 768 slots
        const bool node_fp_on = PK - PKV <= 3;   // see node_upsert
        uint32_t nb = 0, n_nodes = 0;
        Tab ntab;
        ntab.g = gtab;
        const uint32_t want = n_surv + n_surv / 32 + 64;   // (measured: 4 129 nodes for 4 089 survivors)
        // (plan 2 = plan 1 with 5/4 instead of 3/2 table slots per node: 6.5 words)
        // (words per node: inst_of + meta + the two successors of 16 bits each = 3, + the table's 3/2 (plan 2: 5/4) slots of two words, + in
        //  plan 0 one 32-bit pair per oriented node: 8 / 6 / 5.5.  Round 3 kept the successors as two 32-bit words — 9 / 7 / 6.5 —, and a
        //  tenth of C5's gaps fell out of every LDS plan, taking three to four times as long as the others)
        // No LDS plan holds this graph beside the staged pool, but one does in its place (C5 at k = 31: 7 % of the gaps, up to 8 100 nodes
        // beside 900 reads): the graph gets the pool's LDS and the reads are fetched from global memory from here on — the pool is a few
        // dozen KiB in L2, the graph phases touch each node's window a few times; the alternative, every table and array of the graph in the
        // global slice, cost such gaps three to four times the time of the others.  (The count table and the bit arrays are dead by now.)
        // Only with one gap per CU: with two (NT = 512) the gap goes to the launch that has a whole CU for it (`defer`), and the kernels whose
        // pools all fit the LDS keep "the reads are in LDS" as a compile-time fact in their graph phases (C4: 10.4 against 10.6 ms).
        if (NT == 1024 && V.lds && want > 2 * r_words / 11 && (uint64_t)want * 11 <= 2 * (uint64_t)P.lds_words) {
            V.lds = false;
            R = 0;
            r_words = P.lds_words;
        }
        for (int attempt = want <= r_words / 8 ? 0 : want <= r_words / 6 ? 1 : want <= 2 * r_words / 11 ? 2 : 3; attempt < 4; ++attempt) {
            graph_lds = false;
            j_lds = false;
            if (attempt < 3) {
                const uint32_t room = (attempt == 0 ? r_words / 8 : attempt == 1 ? r_words / 6 : 2 * r_words / 11) & ~1u;   // (even: the table behind the arrays stays 8-byte aligned)
                nb = (uint32_t)(node_bound < room ? node_bound + 1 : room) & ~1u;
                if (nb < n_surv || nb < 64) continue;       // cannot even hold one node per survivor
                graph_lds = true;
                j_lds = attempt == 0;
            }
            ntab.lds = graph_lds;
            ntab.off = R + 3 * nb;
            ntab.cap = graph_lds ? ((r_words - (j_lds ? 5 : 3) * nb) / 2) : gcap;
            cand_on = graph_lds && 2ull * ntab.cap <= 4ull * n_unit && ntab.cap <= 32768;   // (the slice of the pair workspace holds two words per slot; an entry is slot << 1 | orientation in 16 bits)
            if (graph_lds) {
                for (uint32_t i = tid; i < ntab.cap; i += ASM_THREADS) ntab.store(i, EMPTY64);
                __syncthreads();
            }
            // ---- P3: nodes + edges; remember each node's slot
            for (uint32_t j = tid; j < n_surv; j += ASM_THREADS) {
                if (graph_lds && s_cnt[6]) break;
                const uint32_t inst_w = list_b[j];
                const uint32_t inst = inst_w & ~INST_WEAK;
                const K128 tf = pv_kmer<W>(V, inst, k);
                K128 t = tf;
                {   // use the canonical k-mer string (what kmc_dump lists); either strand yields the same graph
                    const K128 rc = revcomp_w<W>(t, k);
                    if (rc < t) t = rc;
                }
                const bool fwd = tf == t;
                uint32_t ps = EMPTY32, pd = 0;
                for (uint32_t o = 0; o < per; ++o) {
                    K128 a;  // kv-mer at offset o of t
                    {
                        const int sh = 2 * (int)o;
                        a.hi = sh ? (t.hi << sh) | (t.lo >> (64 - sh)) : t.hi;
                        a.lo = sh ? (t.lo << sh) : t.lo;
                        a = mask_k(a, kv);
                    }
                    const K128 rc = revcomp_w<W>(a, kv);
                    const uint32_t d = rc < a ? 1u : 0u;
                    const K128 A = d ? rc : a;
                    const uint32_t ninst = fwd ? inst + o : inst + (per - 1 - o);  // same read, shifted offset
                    bool fresh;
                    const uint32_t sl = node_upsert<W>(ntab, V, A, ninst, kv, 1u << M_MULT_SHIFT, node_fp_on, &fresh);
                    if (sl == EMPTY32) { if (graph_lds) s_cnt[6] = 1; else atomicOr(&s_cnt[3], ASM_ERR_NTABLE); break; }
                    if (inst_w & INST_WEAK) ntab.or_meta(sl, M_WEAK);   // (a few per cent of the survivors)
                    if (fresh) {
                        const uint32_t q = atomicAdd(&s_cnt[5], 1u);
                        if (graph_lds && q >= nb) { s_cnt[6] = 1; break; }
                        if (q < n_unit) list_a[q] = sl; else atomicOr(&s_cnt[3], ASM_ERR_NLIST);
                    }
                    if (ps != EMPTY32) {  // edge prev -> this
                        const uint32_t c_out = kbase(t, (int)(o - 1) + kv), c_in = kbase(t, (int)o - 1);
                        ntab.or_meta(ps, pd ? (1u << (4 + (3 - c_out))) : (1u << c_out));
                        ntab.or_meta(sl, d ? (1u << (3 - c_in)) : (1u << (4 + c_in)));
                        if (cand_on) {
                            scand[2 * ps + pd] = (uint16_t)((sl << 1) | d);                  // (prev, pd) -> (this, d)
                            scand[2 * sl + (d ^ 1u)] = (uint16_t)((ps << 1) | (pd ^ 1u));    // and the same edge read from the other strand
                        }
                    }
                    ps = sl; pd = d;
                }
            }
            wg_phase_sync();
            if (!(graph_lds && s_cnt[6])) break;
            __syncthreads();
            if (tid == 0) { s_cnt[5] = 0; s_cnt[6] = 0; }   // too many nodes for this plan: redo with the next one
            __syncthreads();
        }
        n_nodes = s_cnt[5] < n_unit ? s_cnt[5] : n_unit;
        if (P.stats && tid < 4) atomicAdd(P.stats + tid, (unsigned long long)(tid == 0 ? n_inst : tid == 1 ? n_dist : tid == 2 ? n_surv : n_nodes));
        const uint32_t astride = graph_lds ? nb : n_nodes;
        uint32_t* garr = P.nodes + 3 * inst_off;
        ASM_STAMP(3);

        // ---- P3.5: dense node indices.  slot.id <- node index; inst_of / meta / succ arrays
        // LDS: [inst_of | meta | succ (16 bits per oriented node)] at R, pairs behind the table;  global: inst_of = list_b, [meta | succ0 | succ1] in
        // the node workspace, pairs in the jump workspace
        const Arr inst_of{graph_lds, R, list_b};
        const Arr nmeta{graph_lds, R + astride, garr};
        const SuccArr succ{graph_lds, R + 2 * astride, garr + astride, astride};
        uint32_t Joff = j_lds ? ntab.off + 2 * ntab.cap : 0;
        Pairs J{j_lds, Joff, reinterpret_cast<unsigned long long*>(P.jump + 4 * inst_off)};
        uint32_t* rec = list_b + (graph_lds ? 0 : n_nodes);           // emitted-walk records, 4 words each
        const uint32_t rec_cap = (n_unit - (graph_lds ? 0 : n_nodes)) / 2;
        for (uint32_t ni = tid; ni < n_nodes; ni += ASM_THREADS) {
            const uint32_t sl = list_a[ni];
            const unsigned long long v = ntab.load(sl);
            inst_of.set(ni, (uint32_t)v);
            nmeta.set(ni, (uint32_t)(v >> 32) & (node_fp_on ? (1u << NODE_FP_SHIFT) - 1u : 0xFFFFFFFFu));   // (without the fingerprint)
            succ.set(2 * ni, EMPTY32);
            succ.set(2 * ni + 1, EMPTY32);
            ntab.set_id(sl, ni);
        }
        wg_phase_sync();

        // ---- neighbourhood helpers (all of them read the adjacency bits, the node table and the staged reads only)
        const uint32_t n_or = 2 * n_nodes;
        auto node_seq = [&](uint32_t o) -> K128 {   // oriented kv-mer of oriented node o
            const K128 x = canonical_w<W>(pv_kmer<W>(V, inst_of.get(o >> 1), kv), kv);
            return (o & 1) ? revcomp_w<W>(x, kv) : x;
        };
        auto find_oriented = [&](K128 y) -> uint32_t {   // oriented node with oriented sequence y, or EMPTY32
            const K128 yr = revcomp_w<W>(y, kv);
            const uint32_t dy = yr < y ? 1u : 0u;
            const K128 Y = dy ? yr : y;
            const uint64_t x = hash_p1<W>(Y);
            const uint32_t fp = node_fp(x);
            uint32_t sl = slot_of_hash(x, ntab.cap);
            uint32_t probes = 0;
            while (probes < ntab.cap) {   // (scan to the next slot with this fingerprint, compare outside the scan: see node_upsert)
                unsigned long long v = ntab.load(sl);
                while ((uint32_t)v != EMPTY32 && node_fp_on && (uint32_t)(v >> (32 + NODE_FP_SHIFT)) != fp) {
                    sl = sl + 1 == ntab.cap ? 0 : sl + 1;
                    if (++probes >= ntab.cap) return EMPTY32;
                    v = ntab.load(sl);
                }
                const uint32_t cand = (uint32_t)v;
                if (cand == EMPTY32) break;
                if (cand < n_nodes && canonical_w<W>(pv_kmer<W>(V, inst_of.get(cand), kv), kv) == Y) return (cand << 1) | dy;
                sl = sl + 1 == ntab.cap ? 0 : sl + 1;
                ++probes;
            }
            return EMPTY32;
        };
        auto adj = [&](uint32_t o) -> uint32_t { return nmeta.get(o >> 1) & 0xFFu; };
        auto outb = [&](uint32_t o) -> uint32_t { return out_bits(adj(o), o & 1); };
        auto inb = [&](uint32_t o) -> uint32_t { return in_bits(adj(o), o & 1); };
        auto has_pred = [&](uint32_t o) { return (nmeta.get(o >> 1) & ((o & 1) ? M_START1 : M_START0)) != 0; };   // internal predecessor
        auto is_dead = [&](uint32_t o) { return (nmeta.get(o >> 1) & M_DEAD) != 0; };
        auto succ_get = [&](uint32_t o) -> uint32_t { return succ.get(o); };

        uint32_t n_emit = 0;
        bool cacc_lds = false;
        uint32_t* cacc = nullptr;
        uint32_t cacc_cap = 0;
        // between phases that hand data over through the graph arrays only: a plain barrier when those live in LDS (the L1
        // invalidate of wg_phase_sync is for hand-overs through global memory)
        auto graph_sync = [&]() { if (j_lds) __syncthreads(); else wg_phase_sync(); };   // (j_lds implies graph_lds)
        // ---- P4: unitig-internal edges.  (x,d) -> (y,dy) is internal iff out-degree(x,d) == 1 and in-degree(y,dy) == 1;
        //      an oriented node that no internal edge enters is a unitig START.
        if (cand_on) {
            for (uint32_t ni = tid; ni < n_nodes; ni += ASM_THREADS) {
                const uint32_t meta = nmeta.get(ni);
                const uint32_t o0 = out_bits(meta & 0xFFu, 0), o1 = out_bits(meta & 0xFFu, 1);
                if (__popc(o0) != 1 && __popc(o1) != 1) continue;
                const uint32_t sl = list_a[ni];
                // (written by other waves two phase barriers ago; read from L2 like every hand-over through global memory)
                const uint32_t cw = __hip_atomic_load(reinterpret_cast<const uint32_t*>(scand) + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // both orientations' entries
                const uint32_t c0 = cw & 0xFFFFu, c1 = cw >> 16;
                for (uint32_t d = 0; d < 2; ++d) {
                    if (__popc(d ? o1 : o0) != 1) continue;
                    const uint32_t c = d ? c1 : c0;
                    const uint32_t y = (ntab.id(c >> 1) << 1) | (c & 1u);
                    if (__popc(inb(y)) != 1) continue;
                    succ.set(2 * ni + d, y);
                    nmeta.or_(y >> 1, (y & 1) ? M_START1 : M_START0);  // here the flag means "has an internal predecessor"
                }
            }
        } else
        for (uint32_t ni = tid; ni < n_nodes; ni += ASM_THREADS) {
            const uint32_t meta = nmeta.get(ni);
            const K128 x = canonical_w<W>(pv_kmer<W>(V, inst_of.get(ni), kv), kv);
            for (uint32_t d = 0; d < 2; ++d) {
                const uint32_t ob = out_bits(meta & 0xFFu, d);
                if (__popc(ob) != 1) continue;
                const K128 cur = d ? revcomp_w<W>(x, kv) : x;
                const uint32_t y = find_oriented(shift_in(cur, __ffs(ob) - 1, kv));
                if (y == EMPTY32) continue;
                if (__popc(inb(y)) != 1) continue;
                succ.set(2 * ni + d, y);
                nmeta.or_(y >> 1, (y & 1) ? M_START1 : M_START0);  // here the flag means "has an internal predecessor"
            }
        }
        graph_sync();
        ASM_STAMP(4);

        // ---- error removal (semantics: oracle/gp_oracle.c simplify_round): rounds on snapshots of the graph.  Every
        //      head decides for its unitig X — TIP (dead end, <= kv nodes, hangs on a junction where another branch beats it) or
        //      BUBBLE (<= 2 kv nodes between two junctions that an alternative path of the same length also joins).  Candidates are
        //      short, so heads WALK the links (no ranking needed yet): a decision only sets the KILL bit of X's end nodes; after the
        //      barrier the arcs into X are cleared at their sources, X's nodes die, and the junctions that lost a branch re-link.
        // unitig headed by h: tail and node count, false when it has more than `limit` nodes
        // (cov = coverage sum << 32 | nodes that are not weak: "more coverage, then fewer weak nodes" is one 64-bit comparison
        //  between unitigs of the same node count — the only ones whose coverage is ever compared)
        auto cov_of = [&](uint32_t m) -> unsigned long long {
            return ((unsigned long long)(m >> M_MULT_SHIFT) << 32) | ((m & M_WEAK) ? 0u : 1u);   // (nmeta carries no fingerprint)
        };
        // (a round has a dozen candidates per gap — 9 at C4, 38 at C5's k = 31 — and lasts as long as the slowest one's chain of walks and
        //  look-ups: a step of the walk asks for the node's successor and its coverage TOGETHER, one LDS round trip instead of two)
        auto walk = [&](uint32_t h, uint32_t limit, uint32_t& tail, uint32_t& n, unsigned long long& cov) -> bool {
            uint32_t cur = h;
            n = 1;
            cov = 0;                             // the coverage sum rides along: no second walk when two unitigs are compared
            for (;;) {
                const uint32_t nx = succ_get(cur);
                const uint32_t m = nmeta.get(cur >> 1);
                cov += cov_of(m);
                if (nx == EMPTY32) break;
                if (n == limit) return false;
                cur = nx;
                ++n;
            }
            tail = cur;
            return true;
        };
        // (always_inline: left as calls, uni_key / beats keep the pool view, the node arrays and two closures in SCRATCH memory — every window of
        // the count phase then reads the view from there)
        auto uni_key = [&](uint32_t h, uint32_t t) __attribute__((always_inline)) -> K128 {
            const K128 a = node_seq(h), b = node_seq(t ^ 1u);
            return b < a ? b : a;
        };
        // does the unitig headed by y (tail ty, ny nodes, coverage cy) beat the one headed by x?
        auto beats = [&](uint32_t y, uint32_t ty, uint32_t ny, unsigned long long cy, uint32_t x, uint32_t tx, uint32_t nx, unsigned long long cx, bool with_len) __attribute__((always_inline)) -> bool {
            if (with_len && ny != nx) return ny > nx;
            if (cy != cx) return cy > cx;     // equal node counts here: more coverage, then fewer weak nodes
            return uni_key(y, ty) < uni_key(x, tx);
        };
        for (uint32_t round = 0; round < P.simplify; ++round) {
            if (tid == 0) { s_cnt[7] = 0; s_cand = 0; s_narc = 0; }
            __syncthreads();
            // candidate heads first (live, no internal predecessor, exactly one arc in) into a queue — the pairs are idle before the
            // ranking —, then one candidate per thread: the evaluation is a serial chain of walks and look-ups (3-40 us each), and
            // a thread that found three candidates in its own stride set the pace of the whole phase
            // (Plan 1 keeps the pairs in the GLOBAL slice until the ranking, and a phase that hands data over through global memory ends
            //  with an L1 invalidate — 1.7 us, eighteen of them in three rounds.  The rounds therefore keep their hand-overs in LDS while they
            //  fit: the queue in s_erq, the removed heads and their arcs in s_arco / s_arcv.)
            for (uint32_t o = tid; o < n_or; o += ASM_THREADS) {
                if (is_dead(o) || has_pred(o) || __popc(inb(o)) != 1) continue;
                const uint32_t qi = atomicAdd(&s_cand, 1u);
                if (qi < ERQ) s_erq[qi] = o; else J.store(qi, o);
            }
            __syncthreads();
            const uint32_t n_cand = s_cand;
            const bool er_lds = graph_lds && n_cand <= ERQ;      // the queue went through LDS
            if (!er_lds) wg_phase_sync();
            // dealt round-robin over the WAVES (candidate q -> wave q mod 16): every wave gets as few divergent lanes as possible and
            // all waves' latency chains overlap (packing them into the first waves measured slower than no queue at all)
            for (uint32_t qi = (tid & 63) * (ASM_THREADS / 64) + (tid >> 6); qi < n_cand; qi += ASM_THREADS) {
                const uint32_t o = qi < ERQ ? s_erq[qi] : (uint32_t)J.load(qi);
                const uint32_t ib = inb(o);
                uint32_t t, n;
                unsigned long long cx;
                if (!walk(o, 2 * PKV, t, n, cx)) continue;
                const K128 hs = node_seq(o);
                const uint32_t p = find_oriented(shift_in_front(hs, __ffs(ib) - 1, kv));
                if (p == EMPTY32) continue;
                const uint32_t pb = outb(p);
                if (__popc(pb) < 2) continue;
                const uint32_t tb = outb(t);
                bool go = false;
                if (tb == 0 && n <= PKV) {                                     // TIP
                    const K128 ps = node_seq(p);
                    for (uint32_t c = 0; c < 4 && !go; ++c) {
                        if (!((pb >> c) & 1u)) continue;
                        const uint32_t y = find_oriented(shift_in(ps, c, kv));
                        if (y == EMPTY32 || y == o || has_pred(y) || y == (t ^ 1u)) continue;
                        uint32_t ty, ny;
                        unsigned long long cy;
                        const bool tip_shaped = walk(y, PKV, ty, ny, cy) && outb(ty) == 0 && __popc(inb(y)) == 1;
                        if (!tip_shaped || beats(y, ty, ny, cy, o, t, n, cx, true)) go = true;
                    }
                } else if (__popc(tb) == 1) {                                   // BUBBLE (n <= 2 kv by the walk)
                    const uint32_t s = find_oriented(shift_in(node_seq(t), __ffs(tb) - 1, kv));
                    if (s != EMPTY32 && __popc(inb(s)) >= 2) {
                        // alternative paths p -> A1 .. Am -> s of whole unitigs, m <= 4, exactly n nodes, none of them X or its reverse
                        uint32_t st_q[4], st_rem[4], st_c[4];
                        int sp = 0;
                        st_q[0] = p; st_rem[0] = n; st_c[0] = 0;
                        while (sp >= 0 && !go) {
                            if (st_c[sp] == 4) { --sp; continue; }
                            const uint32_t c = st_c[sp]++;
                            const uint32_t q = st_q[sp], rem = st_rem[sp];
                            if (!((outb(q) >> c) & 1u)) continue;
                            const uint32_t y = find_oriented(shift_in(node_seq(q), c, kv));
                            if (y == EMPTY32 || has_pred(y) || y == o || y == (t ^ 1u)) continue;
                            uint32_t ty, ny;
                            unsigned long long cy;
                            if (!walk(y, rem, ty, ny, cy)) continue;             // more nodes than remain
                            if (ny == rem) {
                                const uint32_t yb = outb(ty);
                                bool reaches = false;
                                const K128 tys = node_seq(ty);
                                for (uint32_t c2 = 0; c2 < 4; ++c2)
                                    if (((yb >> c2) & 1u) && find_oriented(shift_in(tys, c2, kv)) == s) reaches = true;
                                if (!reaches) continue;
                                if (sp >= 1 || beats(y, ty, ny, cy, o, t, n, cx, false)) go = true;
                            } else if (sp + 1 < 4) {
                                ++sp;
                                st_q[sp] = ty; st_rem[sp] = rem - ny; st_c[sp] = 0;
                            }
                        }
                    }
                }
                if (go) {   // both end nodes: the reverse orientation of a tip does not qualify by itself (its head has no predecessor)
                    nmeta.or_(o >> 1, M_KILL);
                    nmeta.or_(t >> 1, M_KILL);
                    atomicAdd(&s_cnt[7], 2u);      // (at most two heads — X and its reverse — per removed unitig)
                }
            }
            if (er_lds) __syncthreads(); else wg_phase_sync();
            const uint32_t kill_bound = s_cnt[7];
            if (!kill_bound) break;                                             // nothing to remove
            // The removed heads remember the arc that enters them — on the snapshot, before any arc is cleared: in an LDS list when they
            // are few (then the passes below visit the list, not every oriented node), in the pair workspace otherwise.
            const bool arcs_lds = kill_bound <= ERQ;
            const bool round_lds = er_lds && arcs_lds;
            auto er_sync = [&]() { if (round_lds) __syncthreads(); else wg_phase_sync(); };
            for (uint32_t o = tid; o < n_or; o += ASM_THREADS) {
                if (is_dead(o) || has_pred(o) || !(nmeta.get(o >> 1) & M_KILL)) continue;
                const uint32_t ib = inb(o);
                unsigned long long arc = NO_ARC;
                if (__popc(ib) == 1) {
                    const K128 hs = node_seq(o);
                    const uint32_t p = find_oriented(shift_in_front(hs, __ffs(ib) - 1, kv));
                    if (p != EMPTY32) arc = ((unsigned long long)kbase(hs, kv - 1) << 32) | p;
                }
                if (arcs_lds) {
                    const uint32_t qi = atomicAdd(&s_narc, 1u);
                    s_arco[qi] = o;
                    s_arcv[qi] = arc;
                } else J.store(o, arc);
            }
            er_sync();
            const uint32_t n_arc = arcs_lds ? s_narc : n_or;
            // arcs into the removed unitigs are cleared at their sources; the heads walk their unitigs and mark the nodes dead
            for (uint32_t i = tid; i < n_arc; i += ASM_THREADS) {
                const uint32_t o = arcs_lds ? s_arco[i] : i;
                if (!arcs_lds && (has_pred(o) || !(nmeta.get(o >> 1) & M_KILL) || (nmeta.get(o >> 1) & M_DEAD))) continue;
                const unsigned long long arc = arcs_lds ? s_arcv[i] : J.load(o);
                if (arc != NO_ARC) {
                    const uint32_t p = (uint32_t)arc, c = (uint32_t)(arc >> 32);
                    nmeta.and_(p >> 1, ~((p & 1) ? (1u << (4 + (3 - c))) : (1u << c)));
                }
                for (uint32_t cur = o; cur != EMPTY32; cur = succ_get(cur)) nmeta.or_(cur >> 1, M_DEADMARK);
            }
            er_sync();
            // junctions that lost a branch: with one successor left, the edge to it may have become unitig-internal (both directions)
            auto is_gone = [&](uint32_t o) { return (nmeta.get(o >> 1) & (M_DEAD | M_DEADMARK)) != 0; };
            for (uint32_t i = tid; i < n_arc; i += ASM_THREADS) {
                const uint32_t o = arcs_lds ? s_arco[i] : i;
                if (!arcs_lds) {
                    const uint32_t m = nmeta.get(o >> 1);
                    if (has_pred(o) || !(m & M_KILL) || (m & M_DEAD)) continue;  // the heads removed in this round hold their arc
                }
                const unsigned long long arc = arcs_lds ? s_arcv[i] : J.load(o);
                if (arc == NO_ARC) continue;
                const uint32_t p = (uint32_t)arc;
                if (is_gone(p)) continue;
                const uint32_t pb = outb(p);
                if (__popc(pb) != 1) continue;
                const uint32_t y = find_oriented(shift_in(node_seq(p), __ffs(pb) - 1, kv));
                if (y == EMPTY32 || is_gone(y) || __popc(inb(y)) != 1) continue;
                succ.set(p, y);
                nmeta.or_(y >> 1, (y & 1) ? M_START1 : M_START0);
                succ.set(y ^ 1u, p ^ 1u);                                       // the reverse link y' -> p'
                nmeta.or_(p >> 1, (p & 1) ? M_START0 : M_START1);
            }
            er_sync();
            for (uint32_t ni = tid; ni < n_nodes; ni += ASM_THREADS) {
                const uint32_t m = nmeta.get(ni);
                if ((m & M_DEADMARK) && !(m & M_DEAD)) nmeta.or_(ni, M_DEAD);
                if (m & M_KILL) nmeta.and_(ni, ~M_KILL);
            }
            er_sync();
        }

        ASM_STAMP(7);
        // plan 1: nothing looks a node up any more — the ranking pairs move over the node table when they fit there
        if (P.dbg && tid == 0) { P.dbg[(uint64_t)g * 16 + 14] = n_nodes; P.dbg[(uint64_t)g * 16 + 15] = graph_lds ? (j_lds ? 0 : 1) : 2; }   // diagnostics: nodes, plan (0 LDS, 1 LDS + global pairs, 2 global)
        if (graph_lds && !j_lds && 2 * (uint64_t)n_nodes + 8 <= 2 * (uint64_t)ntab.cap) {
            wg_phase_sync();
            Joff = ntab.off;
            J = Pairs{true, Joff, J.g};
            j_lds = true;
        }
        // ---- P5: unitigs ranked by pointer jumping: every oriented node learns its unitig's head and its rank in
        //      ~log2(longest unitig) rounds.  Pairs are read and written as single 64-bit accesses, so the asynchronous
        //      in-place update keeps the invariant "ancestor at that distance".
        for (uint32_t o = tid; o < n_or; o += ASM_THREADS) {
            unsigned long long pr = o;                                      // heads (and dead nodes): {self, 0}
            if (!is_dead(o) && has_pred(o)) {
                const uint32_t back = succ_get(o ^ 1u);                     // walking the other way from this node
                pr = (1ull << 32) | (back ^ 1u);                            // its internal predecessor, one step away
            }
            J.store(o, pr);
        }
        graph_sync();
        // (one barrier per round: three "something changed" flags in rotation — the flag of round jr + 2 is cleared behind the barrier of
        //  round jr, a whole round before anybody sets it; the pairs themselves need no barrier between rounds, see above)
        if (tid < 3) s_jflag[tid] = 0;
        __syncthreads();
        for (int jr = 0; jr < 24; ++jr) {
            bool changed = false;
            for (uint32_t o = tid; o < n_or; o += ASM_THREADS) {
                const unsigned long long a0 = J.load(o);
                const uint32_t pa = (uint32_t)a0;
                if (pa == o) continue;
                const unsigned long long a1 = J.load(pa);
                const uint32_t pb = (uint32_t)a1;
                if (pb == pa) continue;                                     // parent is a head (or this is a finished cycle hop)
                // three jumps per round (the ancestor's ancestor, and its ancestor, as well): half the rounds — each a barrier — for the same number of loads
                uint32_t anc = pb, dist = (uint32_t)(a0 >> 32) + (uint32_t)(a1 >> 32);
#pragma unroll
                for (int hop = 0; hop < 2; ++hop) {
                    const unsigned long long ax = J.load(anc);
                    const uint32_t px = (uint32_t)ax;
                    if (px == anc) break;
                    dist += (uint32_t)(ax >> 32);
                    anc = px;
                }
                J.store(o, ((unsigned long long)dist << 32) | anc);
                changed = true;
            }
            if (changed) s_jflag[jr % 3] = 1;
            __syncthreads();
            if (tid == 0) s_jflag[(jr + 2) % 3] = 0;
            if (!s_jflag[jr % 3]) break;
        }
        graph_sync();
        // tails publish {tail, length} in their head's pair (heads are not read as ancestors any more)
        for (uint32_t o = tid; o < n_or; o += ASM_THREADS) {
            if (is_dead(o) || succ_get(o) != EMPTY32) continue;
            const bool hp = has_pred(o);
            const unsigned long long a0 = hp ? J.load(o) : 0ull;
            const uint32_t h = hp ? (uint32_t)a0 : o;
            if (has_pred(h)) continue;                                      // part of an isolated cycle: never reported
            const uint32_t rank = hp ? (uint32_t)(a0 >> 32) : 0;
            J.store(h, ((unsigned long long)(rank + 1) << 32) | o);
        }
        graph_sync();

        // ---- emission: heads decide, then every node of an emitted unitig writes its own base
        {
            // per-contig coverage sums: in the LDS behind the pairs when the graph lives there, else in the global records
            if (j_lds) {
                const uint32_t used = Joff + 2 * n_nodes;
                cacc = &g_lds[used];
                cacc_cap = P.lds_words > used ? P.lds_words - used : 0;
            }
            for (uint32_t o = tid; o < n_or; o += ASM_THREADS) {
                if (is_dead(o) || has_pred(o)) continue;
                const unsigned long long jo = J.load(o);
                const uint32_t tail = (uint32_t)jo, len = (uint32_t)(jo >> 32);
                uint32_t q = EMPTY32;
                if (len && len + PKV - 1 >= P.min_contig) {
                    const K128 first = node_seq(o);
                    const K128 opp = node_seq(tail ^ 1u);
                    if (!(opp < first)) {
                        q = atomicAdd(&s_cnt[1], 1u);
                        if (q >= rec_cap / 2) { atomicOr(&s_cnt[3], ASM_ERR_WALKS_PAR); q = EMPTY32; }
                        else {
                            rec[4 * q] = o;
                            rec[4 * q + 1] = len;
                            rec[4 * q + 3] = 0;              // coverage sum when the LDS accumulators do not reach this far
                            if (q < cacc_cap) cacc[q] = 0;
                            atomicAdd(&s_seq[0], (unsigned long long)(len + PKV - 1));
                        }
                    }
                }
                J.store(o, q);                               // heads now carry their contig number (EMPTY32: not emitted)
            }
            wg_phase_sync();
            n_emit = s_cnt[1] < rec_cap / 2 ? s_cnt[1] : rec_cap / 2;
            cacc_lds = n_emit <= cacc_cap;   // else (LDS nearly full of nodes): sums accumulate in the global records
            if (tid == 0) {
                s_cnt[2] = n_emit ? atomicAdd(P.n_contigs, n_emit) : 0;
                s_seq[1] = s_seq[0] ? atomicAdd(P.seq_len, s_seq[0]) : 0;
                s_seq[0] = 0;
                if (s_cnt[3]) P.gap_error[g] |= s_cnt[3];
            }
            __syncthreads();
            ASM_STAMP(5);
            for (uint32_t q = tid; q < n_emit; q += ASM_THREADS)                // relative offsets of the contigs
                rec[4 * q + 2] = (uint32_t)atomicAdd(&s_seq[0], (unsigned long long)(rec[4 * q + 1] + PKV - 1));
            wg_phase_sync();
            for (uint32_t o = tid; o < n_or; o += ASM_THREADS) {
                if (is_dead(o)) continue;
                uint32_t h = o, rank = 0;
                if (has_pred(o)) {
                    const unsigned long long a0 = J.load(o);
                    h = (uint32_t)a0;
                    rank = (uint32_t)(a0 >> 32);
                    if (has_pred(h)) continue;
                }
                const uint32_t q = (uint32_t)J.load(h);
                if (q == EMPTY32 || q >= n_emit) continue;
                const unsigned long long off = s_seq[1] + rec[4 * q + 2];
                const uint32_t len = rec[4 * q + 1] + PKV - 1;
                if (cacc_lds) atomicAdd(&cacc[q], nmeta.get(o >> 1) >> M_MULT_SHIFT);
                else atomicAdd(&rec[4 * q + 3], nmeta.get(o >> 1) >> M_MULT_SHIFT);
                if (off + len > P.seq_cap) continue;
                const K128 ok = node_seq(o);
                if (rank == 0) {
                    for (int bq = 0; bq < kv; ++bq) P.seq[off + bq] = "ACGT"[kbase(ok, bq)];
                } else {
                    P.seq[off + kv - 1 + rank] = "ACGT"[kbase(ok, kv - 1)];
                }
            }
            wg_phase_sync();
            for (uint32_t q = tid; q < n_emit; q += ASM_THREADS) {
                const uint32_t ci = s_cnt[2] + q;
                if (ci < P.contig_cap) {
                    gf_contig ct;
                    ct.gap = g; ct.k = (uint16_t)PK; ct.kv = (uint16_t)PKV; ct.n_nodes = rec[4 * q + 1];
                    ct.length = rec[4 * q + 1] + PKV - 1; ct.cov_sum = cacc_lds ? cacc[q] : __hip_atomic_load(&rec[4 * q + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // from L2, where the atomics landed
                    ct.reserved = 0;
                    ct.seq_off = s_seq[1] + rec[4 * q + 2];
                    P.contigs[ci] = ct;
                }
            }
            __syncthreads();
        }
        ASM_STAMP(6);
        // leave the global slice EMPTY for the next gap / launch
        if (!graph_lds) {
            for (uint32_t ni = tid; ni < n_nodes; ni += ASM_THREADS) gtab[list_a[ni]] = EMPTY64;
        }
        if (s_cnt[3])  // an overflow may have left slots outside the lists: clear the whole slice
            for (uint32_t i = tid; i < gcap; i += ASM_THREADS) gtab[i] = EMPTY64;
        wg_phase_sync();
    }
}

template <bool W, int NT, int KC, bool PIPE, bool FIT>
__global__ __launch_bounds__(NT, 4) void assemble_kernel(AsmParams P) {
    __shared__ AsmShared<NT> sh;
    assemble_body<W, NT, KC, PIPE, FIT>(P, sh);
}
// The (k, k_velvet) loop of run_assembly (assemble_gaps.py:87-122) inside ONE launch, for the pipeline's sweep 31/29, 41/39, 51/49: a
// workgroup takes a gap and assembles it three times, then takes the next one.  One tail instead of three (a launch ends when its
// slowest workgroup does), one walk over the gap list, and a gap's contigs stay in (k, kv) order in the contig list — the order the
// reference writes them to the merged contig file (assemble_gaps.py:124-133), which breaks ties between equal picks.  The bodies share
// the workgroup's static LDS and its workspace slice (strided for the largest unit of the three).
template <int NT>
__global__ __launch_bounds__(NT, 4) void assemble_sweep_kernel(AsmParams P31, AsmParams P41, AsmParams P51) {
    __shared__ AsmShared<NT> sh;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) sh.gap = atomicAdd(P31.next_gap, 1u);
        __syncthreads();
        uint32_t g = sh.gap;
        if (P31.gap_list) {
            if (g >= *P31.n_gap_list) break;
            g = P31.gap_list[g];
        } else if (g >= P31.n_pools) break;
        assemble_body<false, NT, 31, true, true, true>(P31, sh, g);
        assemble_body<true, NT, 41, true, true, true>(P41, sh, g);
        assemble_body<true, NT, 51, true, true, true>(P51, sh, g);
    }
}

__global__ void fill_empty_kernel(unsigned long long* t, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        t[i] = 0x00000000FFFFFFFFull;
}

// Deepest pools first (longest-processing-time order): the workgroups take gaps from one counter, and a gap of three times the average
// depth handed out last keeps its CU busy for a millisecond after the others have run dry — per launch.  One workgroup sorts the gaps
// into 256 depth classes (a counting sort; the order inside a class does not matter) and the launch walks that list.
__global__ __launch_bounds__(1024) void asm_order_kernel(const uint64_t* pool_off, uint32_t n_pools, uint32_t* order, uint32_t* n_order) {
    __shared__ uint32_t s_max, s_bin[256];
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { s_max = 1; *n_order = n_pools; }
    if (tid < 256) s_bin[tid] = 0;
    __syncthreads();
    uint32_t mx = 0;
    for (uint32_t g = tid; g < n_pools; g += 1024) {
        const uint64_t n = pool_off[g + 1] - pool_off[g];
        mx = n > mx ? (uint32_t)(n < 0xFFFFFFu ? n : 0xFFFFFFu) : mx;
    }
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t y = __shfl_xor(mx, d); mx = y > mx ? y : mx; }
    if ((tid & 63) == 0) atomicMax(&s_max, mx);
    __syncthreads();
    const uint32_t top = s_max;
    auto cls = [&](uint32_t g) -> uint32_t {
        uint64_t n = pool_off[g + 1] - pool_off[g];
        if (n > top) n = top;                       // (also a pool_off that runs backwards: the kernel refuses that gap itself)
        return 255u - (uint32_t)(n * 255u / top);   // deepest -> class 0
    };
    for (uint32_t g = tid; g < n_pools; g += 1024) atomicAdd(&s_bin[cls(g)], 1u);
    __syncthreads();
    if (tid < 64) {   // exclusive scan of the 256 classes, four per lane
        uint32_t v[4], run = 0;
        for (int q = 0; q < 4; ++q) { v[q] = s_bin[tid * 4 + q]; run += v[q]; }
        uint32_t inc = run;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d); if ((int)tid >= d) inc += y; }
        uint32_t base = inc - run;
        for (int q = 0; q < 4; ++q) { s_bin[tid * 4 + q] = base; base += v[q]; }
    }
    __syncthreads();
    for (uint32_t g = tid; g < n_pools; g += 1024) order[atomicAdd(&s_bin[cls(g)], 1u)] = g;
}

constexpr unsigned ASM_BIG_WGS = 8;   // workgroups (= workspace slices) of the launch that takes the pools beyond asm_max_pool_reads

namespace {
// threads per gap, grid and workspace slices of an assembly launch
struct AsmGeom {
    int nt;
    bool auto_nt;                        // the thread count was chosen here, not by option asm_threads
    uint32_t deep_rows;                  // > 0: pools beyond this many rows leave the main launch (512 threads per gap) for one with 1 024
    unsigned per_cu, grid;
    uint32_t unit;                       // workspace instances per pool row (of the largest unit, for a sweep)
    uint64_t slice_rows, big_rows, big_base, n_inst;
};
// The deepest pool whose count phase still runs in LDS with `lds_words` of dynamic LDS (the kernel's own test before its LDS attempt:
// staged pool, pre-count bit arrays, a table for a twelfth of the windows at 3/4 load).  Beyond it the count table of a gap lives in the
// global slice and the gap takes several times as long — measured on C4's layout at 512 threads per gap (76 KiB): pools of 313 / 421 /
// 530 reads on average 10.4 / 13.7 / 34.2 ms per 19 840 gaps, against 12.7 / 15.5 / 18.4 ms at 1 024 threads (152 KiB).
uint32_t asm_lds_count_rows(const gf_ctx* ctx, uint32_t lds_words, int read_len, int k, int min_count) {
    const uint32_t rb = (uint32_t)((read_len + 3) / 4), npos = (uint32_t)(read_len - k + 1);
    const uint32_t levels = (uint32_t)(min_count < 1 ? 1 : min_count);
    auto fits = [&](uint32_t n_r) {
        const uint32_t pool_words = (((uint32_t)(((uint64_t)n_r * rb + 3) / 4)) + 8) & ~1u;
        if (pool_words + 2048 > lds_words) return false;
        const uint32_t r_words = lds_words - pool_words;
        const uint64_t n_inst = (uint64_t)n_r * npos;
        bool pre = ctx->asm_precount && levels == 2 && k <= 62;
        uint32_t pre_words = 0;
        if (pre) {
            uint32_t lg = 11;
            while ((1ull << lg) < 8 * n_inst && lg < 22) ++lg;
            const uint32_t maxw = (uint32_t)((uint64_t)r_words * (uint32_t)ctx->asm_pre_frac8 / 8) / levels;
            while (lg > 11 && (1u << (lg - 5)) > maxw) --lg;
            pre = (1u << (lg - 5)) <= maxw;
            pre_words = pre ? 1u << (lg - 5) : 0u;
        }
        const uint32_t cap = (r_words - pre_words) / 2, limit = cap - cap / 4;
        return n_inst / (pre ? 12 : 4) <= limit;
    };
    uint32_t lo = 0, hi = 1u << 16;   // fits(lo), !fits(hi)
    if (!fits(1)) return 0;
    lo = 1;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (fits(mid)) lo = mid; else hi = mid; }
    return lo;
}
AsmGeom asm_geometry(const gf_ctx* ctx, size_t n_pools, size_t total_reads, int read_len, uint32_t unit, uint32_t deep_rows_512) {
    AsmGeom G;
    // threads per gap: option asm_threads (1024 / 512 / 256), or automatic: two gaps per CU (512) when the caller's bound on the
    // largest pool (asm_max_pool_reads) leaves room for the tables in half of a CU's LDS and there are gaps enough to keep every CU
    // busy with pairs of them.  Measured on MI355X (us per gap and CU at 1024 / 512 threads): C4's 313-read pools at k = 51: 221 /
    // 386 -> 18.0 / 15.6 ms per 19 840 gaps; every phase takes about twice as long with half the threads — the CU is throughput-bound
    // on its LDS pipeline, not short of ready waves —, the gain is what the barriers and the serial error-removal walks left idle.
    // C2 (1 000 gaps = 4 per CU: latency per gap counts) 0.59 / 0.82 ms; C5's 722-read pools fall out of the LDS plans at 76 KiB (142
    // / 268 ms); four gaps per CU (256 threads, 38 KiB) push C4's graph phases into the global slice: 37 ms.
    G.nt = ctx->asm_threads;
    G.auto_nt = G.nt != 1024 && G.nt != 512 && G.nt != 256;
    // ... and the count phase of MOST pools still runs in the LDS of half a CU (deep_rows_512: asm_lds_count_rows at 76 KiB; the pools
    // beyond it are then listed for a launch with 1 024 threads per gap, see launch_assemble).  What the host knows of the pools is the
    // caller's bound, their maximum or 99th percentile — 1.3-1.4 x their mean in every workload measured (C4 442 / 313, C5 916 / 722,
    // C4 with 1.5 G reads 703 / 530): "the mean pool fits" is read as bound <= 4/3 x deep_rows_512.
    if (G.auto_nt)
        G.nt = (ctx->asm_max_pool_reads > 0 && n_pools >= 8 * (size_t)ctx->n_cu && deep_rows_512 > 0 &&
                (uint64_t)ctx->asm_max_pool_reads * 3 <= (uint64_t)deep_rows_512 * 4) ? 512 : 1024;
    G.deep_rows = G.auto_nt && G.nt == 512 ? deep_rows_512 : 0;
    G.per_cu = 1024u / (unsigned)G.nt;
    G.grid = (unsigned)std::max<size_t>(1, std::min<size_t>(n_pools, (size_t)ctx->n_cu * G.per_cu));
    G.unit = unit;
    // workspace units (see the kernel): one slice per pool row, or — when the caller bounds the rows of one pool (option
    // asm_max_pool_reads; the host entry points know their pools) — one slice of that many rows per workgroup
    G.slice_rows = ctx->asm_max_pool_reads > 0 && (uint64_t)ctx->asm_max_pool_reads * G.grid < total_reads ? (uint64_t)ctx->asm_max_pool_reads : 0;
    // pools beyond that bound (a flank inside a repeat recruits thousands of reads; the reference runs KMC and Velvet on whatever
    // the pool holds) go to a second launch of ASM_BIG_WGS workgroups with slices of asm_big_pool_reads rows; only a pool beyond
    // THAT sets its gap_error
    G.big_rows = G.slice_rows ? std::min<uint64_t>(std::max<uint64_t>((uint64_t)std::max(0l, ctx->asm_big_pool_reads), G.slice_rows), total_reads) : 0;
    G.big_base = G.slice_rows * G.grid * unit;
    G.n_inst = G.slice_rows ? G.big_base + G.big_rows * ASM_BIG_WGS * unit : (uint64_t)total_reads * unit;
    return G;
}
int asm_workspace(gf_ctx* ctx, const AsmGeom& G, bool jump) {
    int rc;
    {   // the table must be all-EMPTY (id 0xFFFFFFFF, meta 0) on entry; kernels leave it so, a fresh buffer is filled once
        // (a re-created buffer is recognised by its SIZE: hipFree + hipMalloc of a larger one may hand back the same base address)
        const void* before = ctx->asm_table.p;
        const size_t bytes_before = ctx->asm_table.bytes;
        if ((rc = ensure(ctx, ctx->asm_table, std::max<uint64_t>(G.n_inst, 1) * 4 * 8))) return rc;
        if (ctx->asm_table.p != before || ctx->asm_table.bytes != bytes_before) {
            const size_t words = ctx->asm_table.bytes / 8;
            hipLaunchKernelGGL(fill_empty_kernel, dim3(ctx->n_cu * 8), dim3(256), 0, ctx->stream,
                               (unsigned long long*)ctx->asm_table.p, (uint64_t)words);
        }
    }
    if ((rc = ensure(ctx, ctx->asm_surv, std::max<uint64_t>(G.n_inst, 1) * 2 * 4))) return rc;
    if ((rc = ensure(ctx, ctx->asm_nodes, std::max<uint64_t>(G.n_inst, 1) * 3 * 4))) return rc;
    if (jump && (rc = ensure(ctx, ctx->asm_jump, std::max<uint64_t>(G.n_inst, 1) * 4 * 4))) return rc;
    return GF_OK;
}
// the caller's buffers of an assembly call
struct AsmIO {
    const void *d_pool, *d_nmask, *d_pool_off;
    size_t n_pools, total_reads;
    int read_len, min_count, min_contig;
    void* d_contigs; size_t contig_cap; void* d_n_contigs;
    void* d_seq; size_t seq_cap; void* d_seq_len;
    void* d_gap_error;
    void *d_cnt_keys, *d_cnt_counts; size_t cnt_cap;
};
AsmParams asm_params(const gf_ctx* ctx, const AsmGeom& G, const AsmIO& io, int k, int kv) {
    AsmParams P;
    P.slice_base = 0;
    P.slice_stride = G.slice_rows * G.unit;
    P.accept_rows = (uint32_t)G.slice_rows;
    P.defer = 0;
    P.defer_tickets = nullptr;
    P.keyslot = (uint32_t)ctx->asm_keyslot;
    P.precount = (uint32_t)ctx->asm_precount;
    P.pre_frac8 = (uint32_t)ctx->asm_pre_frac8;
    P.ranked = (uint32_t)ctx->asm_ranked;
    const uint32_t rb = (uint32_t)((io.read_len + 3) / 4);
    P.reads32 = (const uint32_t*)io.d_pool;
    P.n_words = ((uint64_t)io.total_reads * rb) / 4;
    P.tail_bytes = (uint32_t)(((uint64_t)io.total_reads * rb) & 3);
    P.nmask = (const uint32_t*)io.d_nmask;
    P.pool_off = (const uint64_t*)io.d_pool_off;
    P.n_pools = (uint32_t)io.n_pools;
    P.total_reads = io.total_reads;
    P.rb = rb; P.read_len = io.read_len; P.k = k; P.kv = kv; P.nmw = (io.read_len + 31) / 32;
    P.min_count = io.min_count < 1 ? 1 : io.min_count;
    P.min_contig = io.min_contig < 0 ? 0 : io.min_contig;
    P.table = (unsigned long long*)ctx->asm_table.p;
    P.surv = (uint32_t*)ctx->asm_surv.p;
    P.nodes = (uint32_t*)ctx->asm_nodes.p;
    P.jump = (uint32_t*)ctx->asm_jump.p;
    P.simplify = (uint32_t)std::max(0, ctx->asm_simplify);
    P.tiebreak_counts = ctx->asm_tiebreak ? 1u : 0u;
    P.slice_rows = (uint32_t)G.slice_rows;
    P.contigs = (gf_contig*)io.d_contigs;
    P.contig_cap = (uint32_t)io.contig_cap;
    P.n_contigs = (uint32_t*)io.d_n_contigs;
    P.seq = (char*)io.d_seq;
    P.seq_cap = io.seq_cap;
    P.seq_len = (unsigned long long*)io.d_seq_len;
    P.gap_error = (uint32_t*)io.d_gap_error;
    P.cnt_keys = (uint64_t*)io.d_cnt_keys;
    P.cnt_counts = (uint32_t*)io.d_cnt_counts;
    P.cnt_cap = (uint32_t)std::min<size_t>(io.cnt_cap, 0xFFFFFFFFu);
    P.dbg = (unsigned long long*)ctx->asm_dbg;
    P.stats = (unsigned long long*)ctx->asm_stats;
    // dynamic LDS (option asm_lds_kb, default all 152 KiB): the gap's packed reads, then meta + succ[2] of its nodes;
    // whatever does not fit is read from / kept in global memory
    P.lds_words = std::min<uint32_t>(ASM_LDS_MAX_WORDS / G.per_cu, (uint32_t)std::max(4, ctx->asm_lds_pool_kb) * 256);
    return P;
}
// every pool of the main launch within its share of the LDS
bool asm_pools_fit(const AsmGeom& G, const AsmParams& P, int k) {
    return G.slice_rows > 0 && (uint64_t)P.accept_rows * P.rb + 32 <= (uint64_t)P.lds_words * 4 / (k > 32 ? 2 : 3);
}
// the launch for the pools the main launch listed (none, as a rule: its workgroups leave at once)
void asm_launch_big(gf_ctx* ctx, const AsmGeom& G, const AsmParams& P, uint32_t* next, const uint32_t* list, const uint32_t* n_list) {
    AsmParams B = P;
    B.next_gap = next;
    B.slice_rows = (uint32_t)G.big_rows;
    B.accept_rows = (uint32_t)G.big_rows;
    B.slice_stride = G.big_rows * G.unit;
    B.slice_base = G.big_base;
    B.big_list = nullptr;
    B.gap_list = list;
    B.n_gap_list = n_list;
    B.lds_words = std::min<uint32_t>(ASM_LDS_MAX_WORDS, (uint32_t)std::max(4, ctx->asm_lds_pool_kb) * 256);
    hipLaunchKernelGGL((P.k <= 32 ? assemble_kernel<false, 1024, 0, false, false> : assemble_kernel<true, 1024, 0, false, false>),
                       dim3(ASM_BIG_WGS), dim3(1024), (size_t)B.lds_words * 4, ctx->stream, B);
}
}  // namespace

int launch_assemble(gf_ctx* ctx, const void* d_pool, const void* d_nmask, const void* d_pool_off, size_t n_pools,
                    size_t total_reads, int read_len, int k, int kv, int min_count, int min_contig, void* d_contigs,
                    size_t contig_cap, void* d_n_contigs, void* d_seq, size_t seq_cap, void* d_seq_len, void* d_gap_error,
                    void* d_cnt_keys, void* d_cnt_counts, size_t cnt_cap, bool append) {
    if (k < 16 || k > 64 || read_len < k || read_len > 1000) return GF_E_UNSUPPORTED;
    if (!d_cnt_keys && (kv < 15 || kv >= k || !(kv & 1))) return GF_E_UNSUPPORTED;
    if (n_pools >= 0xFFFFFFFFull || contig_cap > 0xFFFFFFFFull) return GF_E_INVAL;
    const AsmIO io{d_pool, d_nmask, d_pool_off, n_pools, total_reads, read_len, min_count, min_contig, d_contigs, contig_cap, d_n_contigs,
                   d_seq, seq_cap, d_seq_len, d_gap_error, d_cnt_keys, d_cnt_counts, cnt_cap};
    const uint32_t lds_half = std::min<uint32_t>(ASM_LDS_MAX_WORDS / 2, (uint32_t)std::max(4, ctx->asm_lds_pool_kb) * 256);
    const uint32_t deep512 = d_cnt_keys ? 0u : asm_lds_count_rows(ctx, lds_half, read_len, k, min_count);
    const AsmGeom G = asm_geometry(ctx, n_pools, total_reads, read_len, d_cnt_keys ? read_len - k + 1 : read_len - kv + 1, deep512);
    int rc;
    if ((rc = asm_workspace(ctx, G, !d_cnt_keys))) return rc;
    if (n_pools == 0 && !append) {
        GF_HIP(ctx, hipMemsetAsync(d_n_contigs, 0, 4, ctx->stream));
        GF_HIP(ctx, hipMemsetAsync(d_seq_len, 0, 8, ctx->stream));
    }
    if (n_pools == 0) return GF_OK;
    if ((rc = ensure(ctx, ctx->counters, GF_COUNTER_BYTES))) return rc;
    // [8] work counter, [9] the last launch's, [10] pools listed by the main launch, [11] gaps in the launch order, [12] the middle launch's
    // work counter, [13] pools listed by the middle launch, [14] tickets of the gaps that would leave the main launch for their graph's sake
    uint32_t* d_next = (uint32_t*)ctx->counters.p + 8;
    if ((rc = ensure(ctx, ctx->asm_big, 3 * n_pools * 4 + 64))) return rc;   // [listed by the main launch][gaps in launch order][listed by the middle launch]
    // append: a further (k, kv) pair of the same call adds to the contig list and keeps the error flags of the earlier pairs
    if (append) zero_regions(ctx, ZeroList{{d_next, d_next + 4, nullptr, nullptr}, {3, 3, 0, 0}});
    else zero_regions(ctx, ZeroList{{(uint32_t*)d_n_contigs, (uint32_t*)d_seq_len, (uint32_t*)d_gap_error, d_next}, {1, 2, (uint32_t)n_pools, 7}});
    // launch order: the deepest pools first.  The list of an earlier (k, kv) pair of the same call is still good (`append`: same pools).
    uint32_t* d_list1 = (uint32_t*)ctx->asm_big.p;
    uint32_t* d_order = d_list1 + n_pools;
    uint32_t* d_list2 = d_list1 + 2 * n_pools;
    const bool ordered = n_pools >= 4 * (size_t)ctx->n_cu && !d_cnt_keys;
    if (ordered && !append)
        hipLaunchKernelGGL(asm_order_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint64_t*)d_pool_off, (uint32_t)n_pools, d_order, d_next + 3);
    // the kernel of a launch: the generic one, or — the pipeline's launches: k / kv one of its pairs, no N masks, no count-only output —
    // the instantiation for that k (and for "every pool within its share of the LDS")
    auto kernel_of = [&](const AsmGeom& g, const AsmParams& q) -> void (*)(AsmParams) {
        const int nt = g.nt;
        void (*kern)(AsmParams) = nt == 1024 ? (k <= 32 ? assemble_kernel<false, 1024, 0, false, false> : assemble_kernel<true, 1024, 0, false, false>)
                                  : nt == 512 ? (k <= 32 ? assemble_kernel<false, 512, 0, false, false> : assemble_kernel<true, 512, 0, false, false>)
                                              : (k <= 32 ? assemble_kernel<false, 256, 0, false, false> : assemble_kernel<true, 256, 0, false, false>);
        const bool pools_fit = asm_pools_fit(g, q, k);
        if (kv == k - 2 && nt >= 512 && !d_nmask && !d_cnt_keys) {
            if (k == 51 && pools_fit) kern = nt == 1024 ? assemble_kernel<true, 1024, 51, true, true> : assemble_kernel<true, 512, 51, true, true>;
            if (k == 41 && pools_fit) kern = nt == 1024 ? assemble_kernel<true, 1024, 41, true, true> : assemble_kernel<true, 512, 41, true, true>;
            if (k == 31) kern = pools_fit ? (nt == 1024 ? assemble_kernel<false, 1024, 31, true, true> : assemble_kernel<false, 512, 31, true, true>)
                                          : (nt == 1024 ? assemble_kernel<false, 1024, 31, true, false> : assemble_kernel<false, 512, 31, true, false>);
        }
        return kern;
    };
    AsmParams P = asm_params(ctx, G, io, k, kv);
    P.next_gap = d_next;
    P.big_list = G.slice_rows ? d_list1 : nullptr;
    P.n_big = d_next + 2;
    P.gap_list = ordered ? d_order : nullptr;
    P.n_gap_list = ordered ? d_next + 3 : nullptr;
    // Three launches at most.  Main: every workgroup slot of the chip (two gaps per CU at 512 threads).  With two gaps per CU a gap has 76 KiB
    // of LDS, and a pool whose count table does not stay there takes several times as long: the pools too deep by the kernel's own estimate
    // (asm_lds_count_rows) are listed unseen, those whose table runs full (shallow coverage of a long region: many distinct k-mers per window)
    // when it happens, and the list goes to a MIDDLE launch with 1 024 threads and a whole CU's LDS per gap.  Last: the pools beyond the caller's bound (asm_launch_big).
    const bool split = G.slice_rows > 0 && G.deep_rows > 0;
    if (split) { P.accept_rows = (uint32_t)std::min<uint64_t>(G.deep_rows, G.slice_rows); P.defer = 1; P.defer_tickets = d_next + 6; }
    ctx->asm_last_threads = G.nt;
    ctx->asm_last_split = split;
    {
        LaunchTimer tm(ctx, GF_KERNEL_ASSEMBLE);
        hipLaunchKernelGGL(kernel_of(G, P), dim3(G.grid), dim3(G.nt), (size_t)P.lds_words * 4, ctx->stream, P);
        if (split) {
            AsmGeom G2 = G;
            G2.nt = 1024; G2.per_cu = 1;
            G2.grid = (unsigned)std::max<size_t>(1, std::min<size_t>(n_pools, (size_t)ctx->n_cu));   // (<= the main launch's grid: its slices are re-used)
            AsmParams D = asm_params(ctx, G2, io, k, kv);
            D.next_gap = d_next + 4;
            D.gap_list = d_list1;
            D.n_gap_list = d_next + 2;
            D.big_list = d_list2;
            D.n_big = d_next + 5;
            hipLaunchKernelGGL(kernel_of(G2, D), dim3(G2.grid), dim3(1024), (size_t)D.lds_words * 4, ctx->stream, D);
            asm_launch_big(ctx, G, D, d_next + 1, d_list2, d_next + 5);
        } else if (G.slice_rows) asm_launch_big(ctx, G, P, d_next + 1, d_list1, d_next + 2);
    }
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

// The pipeline's sweep (k, kv) = (31, 29), (41, 39), (51, 49) in one launch (assemble_sweep_kernel).  Returns GF_E_UNSUPPORTED, before
// it has touched anything, when the call is not of that shape — the caller then launches per (k, kv).
int launch_assemble_sweep(gf_ctx* ctx, const void* d_pool, const void* d_pool_off, size_t n_pools, size_t total_reads, int read_len,
                          int min_count, int min_contig, void* d_contigs, size_t contig_cap, void* d_n_contigs, void* d_seq,
                          size_t seq_cap, void* d_seq_len, void* d_gap_error) {
    static const int KS[3] = {31, 41, 51};
    if (!ctx->asm_sweep || read_len < 51 || read_len > 1000 || n_pools == 0 || n_pools >= 0xFFFFFFFFull || contig_cap > 0xFFFFFFFFull)
        return GF_E_UNSUPPORTED;
    const AsmIO io{d_pool, nullptr, d_pool_off, n_pools, total_reads, read_len, min_count, min_contig, d_contigs, contig_cap, d_n_contigs,
                   d_seq, seq_cap, d_seq_len, d_gap_error, nullptr, nullptr, 0};
    const uint32_t lds_half = std::min<uint32_t>(ASM_LDS_MAX_WORDS / 2, (uint32_t)std::max(4, ctx->asm_lds_pool_kb) * 256);
    uint32_t deep512 = 0xFFFFFFFFu;   // (no middle launch here: two gaps per CU only when every pool's count phase fits 76 KiB at every k)
    for (int i = 0; i < 3; ++i) deep512 = std::min(deep512, asm_lds_count_rows(ctx, lds_half, read_len, KS[i], min_count));
    if (ctx->asm_max_pool_reads > 0 && (uint64_t)ctx->asm_max_pool_reads > deep512) deep512 = 0;
    const AsmGeom G = asm_geometry(ctx, n_pools, total_reads, read_len, (uint32_t)read_len - 29 + 1, deep512);
    if (G.nt < 512) return GF_E_UNSUPPORTED;
    AsmParams P[3];
    for (int i = 0; i < 3; ++i) {
        P[i] = asm_params(ctx, G, io, KS[i], KS[i] - 2);
        if (!asm_pools_fit(G, P[i], KS[i])) return GF_E_UNSUPPORTED;
    }
    int rc;
    if ((rc = asm_workspace(ctx, G, true))) return rc;
    if ((rc = ensure(ctx, ctx->counters, GF_COUNTER_BYTES))) return rc;
    // [16] work counter, [17] gaps in the launch order, [18 + i] the work counter of pair i's launch for listed pools, [21 + i] pools listed
    uint32_t* d_next = (uint32_t*)ctx->counters.p + 16;
    if ((rc = ensure(ctx, ctx->asm_big, 4 * n_pools * 4 + 64))) return rc;   // [listed by pair 0][pair 1][pair 2][gaps in launch order]
    zero_regions(ctx, ZeroList{{(uint32_t*)d_n_contigs, (uint32_t*)d_seq_len, (uint32_t*)d_gap_error, d_next}, {1, 2, (uint32_t)n_pools, 8}});
    uint32_t* d_order = (uint32_t*)ctx->asm_big.p + 3 * n_pools;
    const bool ordered = n_pools >= 4 * (size_t)ctx->n_cu;
    if (ordered)
        hipLaunchKernelGGL(asm_order_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint64_t*)d_pool_off, (uint32_t)n_pools, d_order, d_next + 1);
    for (int i = 0; i < 3; ++i) {
        P[i] = asm_params(ctx, G, io, KS[i], KS[i] - 2);   // (again: the workspace pointers)
        P[i].next_gap = d_next;
        P[i].big_list = (uint32_t*)ctx->asm_big.p + (size_t)i * n_pools;
        P[i].n_big = d_next + 5 + i;
        P[i].gap_list = ordered ? d_order : nullptr;
        P[i].n_gap_list = ordered ? d_next + 1 : nullptr;
    }
    {
        LaunchTimer tm(ctx, GF_KERNEL_ASSEMBLE);
        if (G.nt == 1024) hipLaunchKernelGGL(assemble_sweep_kernel<1024>, dim3(G.grid), dim3(1024), (size_t)P[0].lds_words * 4, ctx->stream, P[0], P[1], P[2]);
        else hipLaunchKernelGGL(assemble_sweep_kernel<512>, dim3(G.grid), dim3(512), (size_t)P[0].lds_words * 4, ctx->stream, P[0], P[1], P[2]);
        for (int i = 0; i < 3; ++i) asm_launch_big(ctx, G, P[i], d_next + 2 + i, P[i].big_list, P[i].n_big);
    }
    // gf_assemble_last_launch describes a per-pair launch group (its counters are counters[8..13]): after a sweep it reports zeros
    ctx->asm_last_threads = 0;
    ctx->asm_last_split = false;
    GF_HIP(ctx, hipGetLastError());
    return GF_OK;
}

}  // namespace gf
