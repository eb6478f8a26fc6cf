// kernels.hip.h -- hand-written gfx950 (CDNA4, wave64) kernels of the counting path.
//
// Replaces the reference's per-window loop (src/run.rs:526-571) and its DashMap upsert:
//   count_direct_kernel   a1-a7 of SURVEY.md section 8: encode + validity/quality mask +
//                         sliding-window canonicalisation + open-addressing upsert
//   table_* kernels       a8 (into_hashmap) as device compaction, histogram, lookup, rehash
//   owner_* kernels       multi-GPU key-partitioned export
//   synth_reads_kernel    deterministic synthetic reads
//
// All integer work; HBM/atomic bound; no MFMA (there is no contraction on this path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmer_bits.h"

// A kernel that is not a template: every translation unit that includes its header gets a copy of its own (the units that do
// not launch it drop theirs), so that the headers can be shared between units.
#define KH_GLOBAL [[maybe_unused]] static __global__

// The wave's condition mask as it stands.  (__ballot / __any of the HIP headers take an int: the compiler turns the predicate
// into 0 / 1 and compares that again -- a v_cndmask and a v_cmp per call, eight calls per payload round in the region pass.)
__device__ __forceinline__ unsigned long long kh_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool kh_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

namespace kh {

typedef unsigned long long u64;

// One table entry.  key == KH_EMPTY_KEY marks a free slot (count is then 0).
struct alignas(16) Slot {
    u64 key;
    u64 count;
};

// Device-resident running counters (one instance per context).
struct Counters {
    u64 kmers;     // valid windows counted so far (== sum of counts added by count kernels)
    u64 distinct;  // slots claimed so far
    u64 failed;    // upserts that found no free slot (must stay 0)
    u64 cursor;    // scratch cursor for compaction / counting kernels
    u64 big;       // scratch cursor for the histogram's big-count list
    u64 part_failed;  // regions that overflowed in region_count_kernel (re-inserted after growth)
    u64 heads_wide;   // a count too large for 32-bit exchange heads was seen (region_count_kernel32)
    u64 narrow_ovf;   // overflow-list entries whose count would not fit the 8-byte table image (left in the list: ovf_insert_kernel)
    u64 hot;          // buckets the region pass left to hot_buckets_kernel (hot_list_kernel counts them; partition.hip.h)
    u64 hot_total;    // ... and the payloads in them
};

constexpr int BLOCK = 256;           // 4 waves of 64
constexpr int CHUNK = 16;            // bases per thread per tile (one 16-byte load)
constexpr int TILE = BLOCK * CHUNK;  // 4096 positions per workgroup iteration

// ---------------------------------------------------------------------------------------------
// wave helpers (wave = 64 lanes on gfx950)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// number of set bits of `mask` below this lane
__device__ __forceinline__ uint32_t mbcnt(u64 mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ u64 wave_sum(u64 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid in lane 0
}

// ---------------------------------------------------------------------------------------------
// open-addressing table, laid out in REGIONS of REGION_SLOTS slots (64 KiB each):
//   H      = kh_table_hash(key, k)             (bijective 2k-bit hash, left-aligned in 64 bits)
//   region, start = see TableGeom below        (the top bits of H; a fast-range over any multiple of 1024 regions)
// Linear probing wraps INSIDE the region, so a region is self-contained: the direct path updates
// it in HBM with device-scope atomics, the partitioned path rebuilds it in LDS with no global
// atomics at all.  capacity = REGION_SLOTS * regions.
// ---------------------------------------------------------------------------------------------
#ifndef KH_REGION_BITS
#define KH_REGION_BITS 12
#endif
constexpr uint32_t REGION_BITS = KH_REGION_BITS;
constexpr uint32_t REGION_SLOTS = 1u << REGION_BITS;
constexpr uint32_t REGION_MASK = REGION_SLOTS - 1;
// A key's probe sequence inside its region starts at an EVEN slot (and goes on slot by slot from there): the region pass
// then sees the two slots a key most likely sits in with ONE 8-byte LDS read (region_count_kernel32: at load 0.5 a key is in
// its home slot two times in three, in its home pair more than four times in five).  Everything that probes uses start_of /
// narrow_start (or this mask), so the layout is one decision.  Measured (S100M, k = 21; 125 M reads at load 0.61), region
// pass: groups of 1 / 2 / 4 slots 24.5 / 21.7 / 22.5 ms and 43.0 / 36.9 / 36.6 ms -- a 16-byte read costs the LDS twice the
// cycles of an 8-byte one, and eight of them in flight do not fit the registers of two workgroups per CU.
#ifndef KH_REGION_GROUP
#define KH_REGION_GROUP 2  // 1, 2 or 4 (A/B builds: make VARIANT=_g4 EXTRA=-DKH_REGION_GROUP=4)
#endif
constexpr uint32_t REGION_GROUP = KH_REGION_GROUP;
constexpr uint32_t REGION_START_MASK = REGION_MASK & ~(REGION_GROUP - 1);

// Table geometry (round 4: any multiple of 1024 regions, not only powers of two).  A table has NR = b2 << p1_bits regions:
//   p1     = the top p1_bits bits of H                 (the level-1 partition digit of the partitioned path: <= 10 bits)
//   x      = the next 32 bits of H
//   region = p1 * b2 + ((x * b2) >> 32)                (a FAST-RANGE of x over the b2 buckets of partition p1)
//   start  = the top REGION_BITS bits of the product's low word, lowest one cleared (REGION_START_MASK)
// For b2 = 2^j this is exactly "region = the top p1_bits + j bits of H, start = the next REGION_BITS": rounds 1-3's layout,
// and any split (p1_bits, j) of the same sum describes the same table.  A table of <= 1024 regions is always a power of
// two with b2 = 1; a larger one has p1_bits = 10 and ANY b2 (kh_geom_of_regions: the geometry is a function of the region
// count), so that a table can be sized to the load the region pass likes (~0.5) whatever the number of keys -- with powers
// of two the load jumped between 0.3 and 0.6 (125 M reads at 0.61 took 36 ms where 100 M at 0.51 took 19).
// (x, b) <-> (b, frac) is a bijection (x * b2 = b << 32 | frac, an exact division back), so everything that carries "the
// hash bits below the region index" still identifies the key: see kh_below_region / kh_x_of_below.
struct TableGeom {
    Slot *table;
    uint32_t p1_bits;      // regions = b2 << p1_bits
    uint32_t b2;           // buckets (= regions) per level-1 partition
    uint32_t k;
    uint32_t shard_shift;  // 0 = the table covers the whole hash space; n = it is shard `shard_index` of 2^n:
    uint32_t shard_index;  //     every key it holds has shard_index in the top n hash bits, placement uses H << n
};

// ---- geometry arithmetic shared by tables (TableGeom) and partition passes (PartGeom) ------------------------------------
struct RegionGeom {  // (p1_bits, b2) of either
    uint32_t p1_bits, b2;
};
__host__ __device__ inline u64 kh_regions_of(RegionGeom g) { return (u64)g.b2 << g.p1_bits; }
// the geometry of a table with `nregions` regions: a power of two up to 1024 (p1_bits = log2, b2 = 1), a multiple of 1024 beyond
__host__ __device__ inline RegionGeom kh_geom_of_regions(u64 nregions) {
    RegionGeom g;
    if (nregions <= 1024) {
        g.p1_bits = 0;
        while ((2ull << g.p1_bits) <= nregions) ++g.p1_bits;
        g.b2 = 1;
    } else {
        g.p1_bits = 10;
        g.b2 = (uint32_t)(nregions >> 10);
    }
    return g;
}
__host__ __device__ inline bool kh_regions_valid(u64 nregions) {
    return nregions >= 1 && (nregions <= 1024 ? (nregions & (nregions - 1)) == 0 : (nregions & 1023) == 0 && (nregions >> 10) <= (1u << 20));
}
__host__ __device__ inline uint32_t kh_floor_log2(uint32_t v) {
    uint32_t b = 0;
    while ((2u << b) <= v && b < 31) ++b;
    return b;
}
// x (the 32 hash bits behind the level-1 digit) of the placement hash
__host__ __device__ __forceinline__ uint32_t kh_x_of(u64 H, uint32_t p1_bits) { return (uint32_t)((H << p1_bits) >> 32); }
__host__ __device__ __forceinline__ uint32_t kh_p1_of(u64 H, uint32_t p1_bits) { return p1_bits ? (uint32_t)(H >> (64 - p1_bits)) : 0u; }
// bucket of x among b2, and its in-region start
__host__ __device__ __forceinline__ uint32_t kh_bucket_of_x(uint32_t x, uint32_t b2) { return (uint32_t)(((u64)x * b2) >> 32); }
__host__ __device__ __forceinline__ uint32_t kh_start_of_x(uint32_t x, uint32_t b2) {
    return ((uint32_t)(x * b2) >> (32 - REGION_BITS)) & REGION_START_MASK;
}
// smallest x of bucket b (b <= b2: b = b2 gives 2^32): ceil(b * 2^32 / b2)
__host__ __device__ __forceinline__ u64 kh_xlo(uint32_t b, uint32_t b2) { return (((u64)b << 32) + b2 - 1) / b2; }
// ... among the x a k-mer table can hold: with 2k < p1_bits + 32 hash bits the low zs = p1_bits + 32 - 2k bits of every x are
// zero, so the smallest x of the bucket is kh_xlo rounded UP to a multiple of 2^zs -- and x minus THAT keeps its low zs bits
// zero (the exchange units count on it: their count field lives there).  zs = kh_x_zero_bits(k, p1_bits).
__host__ __device__ __forceinline__ uint32_t kh_x_zero_bits(uint32_t k, uint32_t p1_bits) {
    const int z = (int)p1_bits + 32 - 2 * (int)k;
    return z <= 0 ? 0u : (z >= 32 ? 31u : (uint32_t)z);
}
__host__ __device__ __forceinline__ uint32_t kh_xlo_k(uint32_t b, uint32_t b2, uint32_t zs) {
    const u64 m = (1ull << zs) - 1;
    return (uint32_t)((kh_xlo(b, b2) + m) & ~m);
}
// "The hash bits below the region index" as a 32-bit window -- what the exchange units (shard.hip.h) carry: the top
// w = 32 - floor(log2 b2) bits hold x - xlo(bucket) (< 2^w), the bits below them are the hash bits that follow x.
// For b2 = 2^j: bits [p1_bits + j, p1_bits + j + 32) of H, as in rounds 1-3.
__host__ __device__ __forceinline__ uint32_t kh_below_w(uint32_t b2) { return 32u - kh_floor_log2(b2); }
__host__ __device__ __forceinline__ uint32_t kh_below_region(u64 H, RegionGeom g, uint32_t k) {
    const uint32_t x = kh_x_of(H, g.p1_bits), b = kh_bucket_of_x(x, g.b2), w = kh_below_w(g.b2);
    const uint32_t xoff = x - kh_xlo_k(b, g.b2, kh_x_zero_bits(k, g.p1_bits));
    const uint32_t z = w < 32 ? (uint32_t)((H << (g.p1_bits + 32)) >> (32 + w)) : 0u;  // the 32 - w hash bits behind x
    return (w < 32 ? xoff << (32 - w) : xoff) | z;
}
// the placement hash back from (region, window)
__host__ __device__ __forceinline__ u64 kh_hash_of_below(u64 region, uint32_t low, RegionGeom g, uint32_t k) {
    const uint32_t p1 = (uint32_t)(region / g.b2), b = (uint32_t)(region % g.b2), w = kh_below_w(g.b2);
    const uint32_t x = kh_xlo_k(b, g.b2, kh_x_zero_bits(k, g.p1_bits)) + (w < 32 ? low >> (32 - w) : low);
    const u64 z = w < 32 ? (u64)(low << w) : 0ull;  // left-aligned in 32 bits
    u64 H = ((u64)x << 32) | z;                    // x and what follows it, left-aligned in 64 bits ...
    H >>= g.p1_bits;                               // ... behind the level-1 digit
    if (g.p1_bits) H |= (u64)p1 << (64 - g.p1_bits);
    return H;
}
// significant bits of that window for a k-mer table: 2k minus the bits the region index stands for
__host__ __device__ inline int kh_below_bits(uint32_t k, uint32_t shard_shift, RegionGeom g) {
    return 2 * (int)k - (int)shard_shift - (int)g.p1_bits - (int)kh_floor_log2(g.b2);
}

__device__ __forceinline__ RegionGeom rgeom(const TableGeom &tg) { return RegionGeom{tg.p1_bits, tg.b2}; }
// placement hash of a key in this table
__device__ __forceinline__ u64 table_hash(const TableGeom &tg, u64 key) { return kh_table_hash(key, tg.k) << tg.shard_shift; }
__device__ __forceinline__ u64 region_of(const TableGeom &tg, u64 H) {
    return (u64)kh_p1_of(H, tg.p1_bits) * tg.b2 + kh_bucket_of_x(kh_x_of(H, tg.p1_bits), tg.b2);
}
__device__ __forceinline__ uint32_t start_of(const TableGeom &tg, u64 H) { return kh_start_of_x(kh_x_of(H, tg.p1_bits), tg.b2); }

__device__ __forceinline__ void count_add(Slot *s, u64 addend) {
    // fire-and-forget device-scope add (result unused -> no-return global_atomic_add_x2)
    (void)__hip_atomic_fetch_add(&s->count, addend, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Upsert into region `reg` (pointer to its first slot) starting at in-region offset `off`, whose
// key was observed (by a plain, possibly stale load) as `cur`.  A stale observation is harmless: a
// slot only ever goes EMPTY -> K once, so a non-empty value is final, and a stale EMPTY is
// corrected by the value the CAS returns.
__device__ __forceinline__ void upsert_from(Slot *reg, u64 key, uint32_t off, u64 cur, u64 addend,
                                            uint32_t &ndistinct, uint32_t &nfailed) {
    for (uint32_t probes = 0; probes < REGION_SLOTS; ++probes) {
        if (cur == KH_EMPTY_KEY) {
            u64 expected = KH_EMPTY_KEY;
            if (__hip_atomic_compare_exchange_strong(&reg[off].key, &expected, (u64)key, __ATOMIC_RELAXED,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                ++ndistinct;
                cur = key;
            } else {
                cur = expected;  // someone else claimed it first; may be our key
            }
        }
        if (cur == key) {
            count_add(&reg[off], addend);
            return;
        }
        off = (off + 1) & REGION_MASK;
        cur = reg[off].key;
    }
    ++nfailed;  // region full: the host keeps the load factor far below this
}

__device__ __forceinline__ void upsert(const TableGeom &tg, u64 key, u64 addend, uint32_t &ndistinct,
                                       uint32_t &nfailed) {
    const u64 H = table_hash(tg, key);
    Slot *reg = tg.table + region_of(tg, H) * REGION_SLOTS;
    const uint32_t off = start_of(tg, H);
    upsert_from(reg, key, off, reg[off].key, addend, ndistinct, nfailed);
}

// ---------------------------------------------------------------------------------------------
// G1+G2: encode + mask + canonicalise + upsert, straight into the HBM table
// ---------------------------------------------------------------------------------------------
// Positions are "virtual": abase is the 16-byte-aligned address at or below the caller's pointer,
// real data occupies [vbeg, vend), and only windows ENDING at >= wlo are counted (wlo > vbeg when
// the bytes before it are a halo that an earlier launch already counted).
//
// Per tile of 4096 positions a workgroup: (1) loads 16 bases (+16 quals) per lane with one
// coalesced 16-byte load each, (2) turns them into a 32-bit 2-bit-code word and a 16-bit
// validity word, staged in LDS so every lane can read the two words in front of its own (the
// k-1 <= 31 base look-back), (3) extracts the 16 windows ending in its chunk by funnel shifts,
// canonicalises them and (4) upserts them eight at a time (eight independent slot loads in
// flight per lane before the first atomic).
__device__ __forceinline__ uint32_t byte_of(const uint4 &w, int j) {
    uint32_t d = (j < 4) ? w.x : (j < 8) ? w.y : (j < 12) ? w.z : w.w;
    return (d >> (8 * (j & 3))) & 0xFFu;
}

// Raw 16-byte chunk(s) of one lane, as loaded from HBM (zeros outside the data).
struct RawChunk {
    uint4 w, q;
    int64_t p0;
    bool live;
};

template <bool QUAL>
__device__ __forceinline__ RawChunk load_raw(const uint8_t *__restrict__ abase, const uint8_t *__restrict__ qbase,
                                             int qaligned, int64_t p0, u64 vbeg, u64 vend) {
    RawChunk r;
    r.p0 = p0;
    r.w = make_uint4(0, 0, 0, 0);
    r.q = make_uint4(0, 0, 0, 0);
    r.live = !(p0 < 0 || (u64)p0 >= vend);
    // The 16-byte loads are UNCONDITIONAL (a chunk outside the data reads the data's first chunk instead and is
    // ignored by encode_raw): behind a branch the loaded registers reach the caller through copies at the join,
    // the compiler waits for the load right there, and the "prefetch" of the next tile stalls every wave for a
    // memory round trip per tile (s_waitcnt vmcnt(0) straight after the global_load in every level-1 kernel).
    const int64_t pa = r.live ? p0 : (int64_t)(vbeg & ~(u64)15);
    r.w = *reinterpret_cast<const uint4 *>(abase + pa);
    if (QUAL) {
        if (qaligned) {
            r.q = *reinterpret_cast<const uint4 *>(qbase + pa);
        } else if (r.live) {
            uint32_t qq[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < CHUNK; ++j) {
                u64 pos = (u64)p0 + j;
                uint32_t b = (pos >= vbeg && pos < vend) ? qbase[pos] : 0u;
                qq[j >> 2] |= b << (8 * (j & 3));
            }
            r.q = make_uint4(qq[0], qq[1], qq[2], qq[3]);
        }
    }
    return r;
}

// ---- SWAR helpers: four bases per 32-bit word -------------------------------------------------
// 0x80 in every byte of x that is zero (exact: the per-byte add cannot carry into its neighbour)
__device__ __forceinline__ uint32_t swar_zero_bytes(uint32_t x) {
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}
// 0x80 in every byte where x >= y (unsigned bytes)
__device__ __forceinline__ uint32_t swar_ge_bytes(uint32_t x, uint32_t y) {
    const uint32_t z = (x | 0x80808080u) - (y & 0x7F7F7F7Fu);  // bit 7: low 7 bits of x >= low 7 bits of y
    return ((x & ~y) | (~(x ^ y) & z)) & 0x80808080u;
}
// bytes b0..b3 (b0 = lowest address = FIRST base) -> 8 bits, first base most significant.
// Round 4: the four 2-bit fields are gathered by ONE multiply -- field i sits at bit 8 i, the factor 2^30 + 2^20 + 2^10 + 1
// moves it to bit 30 - 2 i of the product's top byte, and no other partial product lands in that byte (they fall below bit
// 24 or beyond bit 31) -- where rounds 1-3 took three shifts and three ORs (the encoder was 110 of the ~945 vector
// instructions a level-1 wave spends on a tile; ISA of part1_bins_kernel).  swar_codes4_top leaves the byte where the
// multiply puts it (bits 24..31; junk below), for encode_raw to pick with v_perm_b32.
__device__ __forceinline__ uint32_t swar_codes4_top(uint32_t w) {
    const uint32_t t = ((w >> 1) ^ (w >> 2)) & 0x03030303u;  // A,C,G,T -> 0,1,2,3 in every byte (either case)
    return t * 0x40100401u;
}
__device__ __forceinline__ uint32_t swar_codes4(uint32_t w) { return swar_codes4_top(w) >> 24; }
// 0x80-per-byte flags -> 4 bits, first base most significant (the same gather: flag i at bit 8 i after the shift, factor
// 2^27 + 2^18 + 2^9 + 1, result in bits 24..27 of the product, nothing above)
__device__ __forceinline__ uint32_t swar_flags4(uint32_t m) { return ((m >> 7) * 0x08040201u) >> 24; }
__device__ __forceinline__ uint32_t swar_valid4(uint32_t w, uint32_t q, uint32_t thr4, bool qual) {
    // Accepted bytes are exactly ACGTacgt (src/kmer.rs:271-273).  Every byte has SOME 2-bit code; a
    // byte is a base iff, case folded, it equals the letter of its own code.  The four letters come
    // from one byte-select (v_perm_b32 with the codes as selectors) instead of four SWAR compares.
    const uint32_t u = w & 0xDFDFDFDFu;
    const uint32_t t = ((w >> 1) ^ (w >> 2)) & 0x03030303u;               // as in swar_codes4
    const uint32_t letters = __builtin_amdgcn_perm(0u, 0x54474341u, t);  // selector 0..3 -> 'A','C','G','T'
    uint32_t m = swar_zero_bytes(u ^ letters);
    if (qual) m &= swar_ge_bytes(q, thr4);  // run.rs:545: skip iff qv < threshold
    return swar_flags4(m);
}

// 16 bases -> 32-bit code word (first base in the top two bits) + 16-bit validity word (bit 15-j =
// base j is countable: in ACGTacgt, inside [vbeg, vend), quality >= thr).
template <bool QUAL>
__device__ __forceinline__ void encode_raw(const RawChunk &r, u64 vbeg, u64 vend, uint32_t thr, uint32_t &code,
                                           uint32_t &val) {
    code = 0;
    val = 0;
    if (!r.live) return;
    const uint32_t thr4 = thr * 0x01010101u;
    // (the top bytes of the four products, first word's first: two byte-selects and an OR)
    code = __builtin_amdgcn_perm(swar_codes4_top(r.w.x), swar_codes4_top(r.w.y), 0x07030c0cu) |
           __builtin_amdgcn_perm(swar_codes4_top(r.w.z), swar_codes4_top(r.w.w), 0x0c0c0703u);
    val = (swar_valid4(r.w.x, r.q.x, thr4, QUAL) << 12) | (swar_valid4(r.w.y, r.q.y, thr4, QUAL) << 8) |
          (swar_valid4(r.w.z, r.q.z, thr4, QUAL) << 4) | swar_valid4(r.w.w, r.q.w, thr4, QUAL);
    // data bounds, once per chunk instead of two 64-bit compares per base
    const u64 p0 = (u64)r.p0;
    if (p0 < vbeg) {
        const u64 d = vbeg - p0;  // first d bases lie before the data
        val = d >= 16 ? 0u : (val & (0xFFFFu >> d));
    }
    if (p0 + CHUNK > vend) {
        const u64 e = p0 + CHUNK - vend;  // last e bases lie past the data (1..15 here: the chunk is live)
        val &= ~((1u << e) - 1u);
    }
}

template <bool QUAL>
__device__ __forceinline__ void encode_chunk(const uint8_t *__restrict__ abase, const uint8_t *__restrict__ qbase,
                                             int qaligned, int64_t p0, u64 vbeg, u64 vend, uint32_t thr,
                                             uint32_t &code, uint32_t &val) {
    const RawChunk r = load_raw<QUAL>(abase, qbase, qaligned, p0, vbeg, vend);
    encode_raw<QUAL>(r, vbeg, vend, thr, code, val);
}

// One staged tile as seen by one lane: its own 16 codes (low half of lo64), the 32 codes in front
// of them (high half of lo64, hi) and 48 validity bits (bit 15-j = own base j, older bases higher).
struct WinCtx {
    u64 lo64;
    u64 V;
    u64 p0;  // virtual position of the lane's first base
    uint32_t hi;
};

// Stages tile t (NT*16 positions) for a workgroup of NT lanes.  s_code/s_val are [2][NT+2] LDS
// arrays, double-buffered by `buf`; `first` = this is the workgroup's first tile (its look-back
// words are then encoded from memory, later tiles get them carried from the previous tile).
// Contains the tile's only barrier.
// Position of lane tid's chunk in tile t.
template <int NT>
__device__ __forceinline__ int64_t chunk_pos(u64 t, int tid) { return (int64_t)(t * (u64)(NT * CHUNK) + (u64)tid * CHUNK); }

template <bool QUAL, int NT>
__device__ __forceinline__ WinCtx stage_tile_raw(uint32_t (*s_code)[NT + 2], uint16_t (*s_val)[NT + 2], int buf, bool first,
                                                 int tid, const RawChunk &raw, const uint8_t *__restrict__ abase,
                                                 const uint8_t *__restrict__ qbase, int qaligned, u64 t, u64 vbeg,
                                                 u64 vend, uint32_t thr) {
    constexpr int TILE_N = NT * CHUNK;
    const int64_t p0 = raw.p0;
    uint32_t code, val;
    encode_raw<QUAL>(raw, vbeg, vend, thr, code, val);
    s_code[buf][tid + 2] = code;
    s_val[buf][tid + 2] = (uint16_t)val;
    if (first && tid < 2) {
        uint32_t hc, hv;
        encode_chunk<QUAL>(abase, qbase, qaligned, (int64_t)(t * TILE_N) - (int64_t)(2 - tid) * CHUNK, vbeg, vend, thr,
                           hc, hv);
        s_code[buf][tid] = hc;
        s_val[buf][tid] = (uint16_t)hv;
    }
    __syncthreads();
    // Carry the last two words to the next tile's look-back slots.  Done AFTER the barrier: the
    // other buffer's slots [0,1] were last read in the previous iteration, and every reader is
    // past those reads once it has arrived here.
    if (tid >= NT - 2) {
        s_code[buf ^ 1][tid - (NT - 2)] = code;
        s_val[buf ^ 1][tid - (NT - 2)] = (uint16_t)val;
    }
    WinCtx w;
    w.hi = s_code[buf][tid];
    w.lo64 = ((u64)s_code[buf][tid + 1] << 32) | code;
    w.V = ((u64)s_val[buf][tid] << 32) | ((u64)s_val[buf][tid + 1] << 16) | (u64)val;
    w.p0 = (u64)p0;
    return w;
}

// The same in two halves, for a kernel that encodes tile t + 1 while it still works on tile t (so that the wait for
// the prefetched bases is not also a wait for the global stores the kernel has just issued -- one counter, vmcnt,
// covers both): stage_encode() is everything before the barrier, stage_collect() everything after it.
template <bool QUAL, int NT>
__device__ __forceinline__ void stage_encode(uint32_t (*s_code)[NT + 2], uint16_t (*s_val)[NT + 2], int buf, bool first, int tid,
                                             const RawChunk &raw, const uint8_t *__restrict__ abase,
                                             const uint8_t *__restrict__ qbase, int qaligned, u64 t, u64 vbeg, u64 vend,
                                             uint32_t thr) {
    constexpr int TILE_N = NT * CHUNK;
    uint32_t code, val;
    encode_raw<QUAL>(raw, vbeg, vend, thr, code, val);
    s_code[buf][tid + 2] = code;
    s_val[buf][tid + 2] = (uint16_t)val;
    if (first && tid < 2) {
        uint32_t hc, hv;
        encode_chunk<QUAL>(abase, qbase, qaligned, (int64_t)(t * TILE_N) - (int64_t)(2 - tid) * CHUNK, vbeg, vend, thr,
                           hc, hv);
        s_code[buf][tid] = hc;
        s_val[buf][tid] = (uint16_t)hv;
    }
}
template <int NT>
__device__ __forceinline__ WinCtx stage_collect(uint32_t (*s_code)[NT + 2], uint16_t (*s_val)[NT + 2], int buf, int tid, u64 t) {
    const uint32_t code = s_code[buf][tid + 2];
    const uint32_t val = s_val[buf][tid + 2];
    if (tid >= NT - 2) {  // the last two words are the next tile's look-back (that buffer's slots [0,1] are free: see stage_tile_raw)
        s_code[buf ^ 1][tid - (NT - 2)] = code;
        s_val[buf ^ 1][tid - (NT - 2)] = (uint16_t)val;
    }
    WinCtx w;
    w.hi = s_code[buf][tid];
    w.lo64 = ((u64)s_code[buf][tid + 1] << 32) | code;
    w.V = ((u64)s_val[buf][tid] << 32) | ((u64)s_val[buf][tid + 1] << 16) | (u64)val;
    w.p0 = (u64)chunk_pos<NT>(t, tid);
    return w;
}

template <bool QUAL, int NT>
__device__ __forceinline__ WinCtx stage_tile(uint32_t (*s_code)[NT + 2], uint16_t (*s_val)[NT + 2], int buf, bool first,
                                             int tid, const uint8_t *__restrict__ abase,
                                             const uint8_t *__restrict__ qbase, int qaligned, u64 t, u64 vbeg, u64 vend,
                                             uint32_t thr) {
    const RawChunk raw = load_raw<QUAL>(abase, qbase, qaligned, chunk_pos<NT>(t, tid), vbeg, vend);
    return stage_tile_raw<QUAL, NT>(s_code, s_val, buf, first, tid, raw, abase, qbase, qaligned, t, vbeg, vend, thr);
}

// Rolling canonical-window state of one lane over its 16 bases.  All 32-bit arithmetic:
//   f  = the last 32 bases as a 64-bit shift register (two words), newest base in the low bits
//   rc = reverse complement of the last k bases in the low 2k bits
//   good = bit (15-j): the window ENDING at own base j is countable (k valid bases, inside the
//          data, ending at or after wlo) -- computed once per lane by eroding the 48 validity bits
// next(j) must be called for j = 0, 1, ..., 15 in order.
// bit (15-j): the window ENDING at the lane's own base j is countable (k valid bases, inside the data, ending at
// or after wlo): G = AND_{i<k} (V >> i) over the lane's 48 validity bits
__device__ __forceinline__ uint32_t window_good(const WinCtx &w, uint32_t k, u64 wlo) {
    u64 g = w.V;
    uint32_t L = 1;
#pragma unroll
    for (uint32_t sft = 1; sft <= 16; sft <<= 1)
        if (2 * sft <= k) {
            g &= g >> sft;
            L = 2 * sft;
        }
    if (k > L) g &= g >> (k - L);
    uint32_t good = (uint32_t)g & 0xFFFFu;
    if (w.p0 < wlo) {  // windows ending before wlo belong to an earlier launch
        const u64 d = wlo - w.p0;
        good = d >= 16 ? 0u : (good & (0xFFFFu >> d));
    }
    return good;
}

struct Roller {
    uint32_t flo, fhi, rlo, rhi, code, good;
    uint32_t kmlo, kmhi;  // kh_kmask(k)
    uint32_t ins_sh;      // (2k-2) mod 32: where the complemented new base enters rc ...
    uint32_t ins_mlo, ins_mhi;  // ... in the high word (k > 16) or the low word: all-ones mask of that word

    __device__ __forceinline__ void init(const WinCtx &w, uint32_t k, u64 wlo) {
        const u64 km = kh_kmask(k);
        kmlo = (uint32_t)km;
        kmhi = (uint32_t)(km >> 32);
        ins_sh = (2 * k - 2) & 31;
        ins_mhi = k > 16 ? 0xFFFFFFFFu : 0u;
        ins_mlo = ~ins_mhi;
        code = (uint32_t)w.lo64;
        flo = (uint32_t)(w.lo64 >> 32);  // chunk t-1: bases -16..-1
        fhi = w.hi;                      // chunk t-2: bases -32..-17
        const u64 rc = kh_revcomp((((u64)fhi << 32) | flo) & km, k);
        rlo = (uint32_t)rc;
        rhi = (uint32_t)(rc >> 32);
        good = window_good(w, k, wlo);
    }

    __device__ __forceinline__ bool next(int j, u64 &key) {
        const uint32_t c = (code >> (30 - 2 * j)) & 3u;
        fhi = __builtin_amdgcn_alignbit(fhi, flo, 30);  // (f << 2) high word
        flo = (flo << 2) | c;
        rlo = __builtin_amdgcn_alignbit(rhi, rlo, 2);   // rc >> 2
        rhi >>= 2;
        const uint32_t ins = (c ^ 3u) << ins_sh;         // complement enters at the top of the 2k bits
        rlo |= ins & ins_mlo;                            // (uniform masks, one v_and_or each: no 64-bit shift, no selects)
        rhi |= ins & ins_mhi;
        const u64 fwd = ((u64)(fhi & kmhi) << 32) | (flo & kmlo);
        const u64 rc = ((u64)rhi << 32) | rlo;
        key = fwd < rc ? fwd : rc;  // integer min == the reference's lexicographic choice (kmer.rs:348-365)
        return (good >> (15 - j)) & 1u;
    }
};

template <bool QUAL>
__global__ __launch_bounds__(BLOCK) void count_direct_kernel(
    const uint8_t *__restrict__ abase, const uint8_t *__restrict__ qbase, int qaligned, u64 vbeg, u64 vend,
    u64 wlo, u64 tile0, u64 ntiles, uint32_t tiles_per_block, uint32_t k, uint32_t thr, TableGeom tg, Counters *ctr) {
    __shared__ uint32_t s_code[2][BLOCK + 2];
    __shared__ uint16_t s_val[2][BLOCK + 2];

    const int tid = threadIdx.x;
    const u64 tb = tile0 + (u64)blockIdx.x * tiles_per_block;
    u64 te = tb + tiles_per_block;
    if (te > tile0 + ntiles) te = tile0 + ntiles;

    uint32_t nk = 0, nd = 0, nf = 0;

    int buf = 0;
    for (u64 t = tb; t < te; ++t, buf ^= 1) {
        const WinCtx w = stage_tile<QUAL, BLOCK>(s_code, s_val, buf, t == tb, tid, abase, qbase, qaligned, t, vbeg, vend, thr);
        Roller roll;
        roll.init(w, k, wlo);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u64 key[8];
            Slot *reg[8];
            uint32_t off[8];
            u64 cur[8];
            uint32_t ok = 0;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) ok |= (uint32_t)roll.next(half * 8 + jj, key[jj]) << jj;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const u64 H = table_hash(tg, key[jj]);
                reg[jj] = tg.table + region_of(tg, H) * REGION_SLOTS;
                off[jj] = start_of(tg, H);
                cur[jj] = KH_EMPTY_KEY;
                if (ok & (1u << jj)) cur[jj] = reg[jj][off[jj]].key;
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                // Skew guard (all lanes): a same-address device atomic costs ~6-17 ns, so a homopolymer
                // run (every window of every lane the same key) would serialise the whole wave.  Lanes
                // holding the first valid lane's key hand it their increment.
                bool mine = ok & (1u << jj);
                nk += mine;
                u64 weight = 1;
                const u64 vmask = kh_ballot(mine);
                if (vmask) {
                    const int first = __builtin_ctzll(vmask);
                    const u64 lead = __shfl(key[jj], first, 64);
                    const bool same = mine && key[jj] == lead;
                    const u64 smask = kh_ballot(same);
                    if (__builtin_popcountll(smask) > 1) {
                        if ((int)lane_id() == first) weight = (u64)__builtin_popcountll(smask);
                        else if (same) mine = false;
                    }
                }
                if (mine) {
                    if (cur[jj] == key[jj]) count_add(&reg[jj][off[jj]], weight);
                    else upsert_from(reg[jj], key[jj], off[jj], cur[jj], weight, nd, nf);
                }
            }
        }
    }

    // wave-aggregated statistics: one atomic per wave per counter
    u64 s = wave_sum((u64)nk);
    u64 d = wave_sum((u64)nd);
    u64 f = wave_sum((u64)nf);
    if (lane_id() == 0) {
        if (s) atomicAdd(&ctr->kmers, s);
        if (d) atomicAdd(&ctr->distinct, d);
        if (f) atomicAdd(&ctr->failed, f);
    }
}

// How many windows of a range survive masking (N, soft-masked, quality below the threshold)?  Counted over every
// `stride`-th 4096-position tile: the host sizes the partition buffers of a quality-masked range from this instead of
// from "every window" (kmerhip.hip, count_device_range; a wrong estimate costs a retry, never a result).
template <bool QUAL>
__global__ __launch_bounds__(BLOCK) void survival_sample_kernel(
    const uint8_t *__restrict__ abase, const uint8_t *__restrict__ qbase, int qaligned, u64 vbeg, u64 vend,
    u64 wlo, u64 tile0, u64 ntiles, u64 stride, uint32_t k, uint32_t thr, u64 *__restrict__ out) {
    __shared__ uint32_t s_code[2][BLOCK + 2];
    __shared__ uint16_t s_val[2][BLOCK + 2];
    // (a workgroup per sampled tile would be 4 same-address atomics each: 230 k of them at ~10 ns took 2.8 ms on S100M; a
    //  fixed grid, every workgroup over its share of the sampled tiles, one atomic per wave at the end: 0.1 ms)
    uint32_t n = 0;
    int buf = 0;
    for (u64 t = tile0 + (u64)blockIdx.x * stride; t < tile0 + ntiles; t += (u64)gridDim.x * stride, buf ^= 1) {
        const WinCtx w = stage_tile<QUAL, BLOCK>(s_code, s_val, buf, true, (int)threadIdx.x, abase, qbase, qaligned, t, vbeg, vend, thr);
        n += (uint32_t)__builtin_popcount(window_good(w, k, wlo));
        __syncthreads();  // (stage_tile carries this tile's last words into the other buffer's look-back slots, which the next
                          //  -- not consecutive -- tile fills from memory: keep the two writes apart)
    }
    const u64 s = wave_sum((u64)n);
    if (lane_id() == 0 && s) atomicAdd(out, s);
}

#ifndef KH_HELPERS_ONLY  // (the level-1 translation units take the helpers above, not the kernels below: one definition each)
// ---------------------------------------------------------------------------------------------
// table maintenance
// ---------------------------------------------------------------------------------------------
KH_GLOBAL __launch_bounds__(BLOCK) void table_init_kernel(Slot *table, u64 cap) {
    const u64 stride = (u64)gridDim.x * BLOCK;
    uint4 e = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < cap; i += stride)
        *reinterpret_cast<uint4 *>(&table[i]) = e;
}

// ctr->cursor += number of live slots with count >= min_count
KH_GLOBAL __launch_bounds__(BLOCK) void table_count_kernel(const Slot *table, u64 cap, u64 min_count, Counters *ctr) {
    const u64 stride = (u64)gridDim.x * BLOCK;
    u64 n = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < cap; i += stride) {
        const Slot s = table[i];
        n += (s.key != KH_EMPTY_KEY && s.count >= min_count) ? 1 : 0;
    }
    n = wave_sum(n);
    if (lane_id() == 0 && n) atomicAdd(&ctr->cursor, n);
}

// Stream compaction of live slots with ONE cursor atomic per tile of COMPACT_PER x BLOCK slots.  (Round 1-3a had one per
// WAVE: 33 M atomics on one word for a 2^31-slot table, ~10 ns each -- 0.3 of the 0.7 s of a kh_result_copy of S100M.)
// load(i, key, count) -> is slot i live; pairs past out_cap are not written, ctr->cursor still counts them.
constexpr int COMPACT_PER = 16;
template <typename LOAD>
__device__ __forceinline__ void compact_tiles(u64 cap, u64 *__restrict__ keys, u64 *__restrict__ counts, u64 out_cap, Counters *ctr, LOAD load) {
    __shared__ uint32_t s_wave[BLOCK / 64];
    __shared__ u64 s_base;
    const int tid = threadIdx.x, wave = tid >> 6;
    const u64 TILE = (u64)COMPACT_PER * BLOCK;
    const u64 ntiles = (cap + TILE - 1) / TILE;
    for (u64 t = blockIdx.x; t < ntiles; t += gridDim.x) {  // (uniform per workgroup: barriers inside)
        u64 k[COMPACT_PER], c[COMPACT_PER];
        uint32_t off[COMPACT_PER], live = 0, wtotal = 0;
#pragma unroll
        for (int j = 0; j < COMPACT_PER; ++j) {
            const u64 i = t * TILE + (u64)j * BLOCK + tid;
            k[j] = 0;
            c[j] = 0;
            const bool lv = i < cap && load(i, k[j], c[j]);
            const u64 m = kh_ballot(lv);
            off[j] = wtotal + mbcnt(m);
            wtotal += (uint32_t)__builtin_popcountll(m);
            live |= (lv ? 1u : 0u) << j;
        }
        if ((tid & 63) == 0) s_wave[wave] = wtotal;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) {
            const uint32_t x = s_wave[w];
            before += w < wave ? x : 0u;
            total += x;
        }
        if (tid == 0 && total) s_base = atomicAdd(&ctr->cursor, (u64)total);
        __syncthreads();
        if (total) {
            const u64 base = s_base + before;
#pragma unroll
            for (int j = 0; j < COMPACT_PER; ++j)
                if ((live >> j) & 1u) {
                    const u64 o = base + off[j];
                    if (o < out_cap) {
                        keys[o] = k[j];
                        counts[o] = c[j];
                    }
                }
        }
        __syncthreads();  // (s_wave / s_base are reused by the next tile)
    }
}
KH_GLOBAL __launch_bounds__(BLOCK) void table_compact_kernel(const Slot *table, u64 cap, u64 min_count, u64 *keys,
                                                              u64 *counts, u64 out_cap, Counters *ctr) {
    compact_tiles(cap, keys, counts, out_cap, ctr, [&](u64 i, u64 &key, u64 &count) {
        const uint4 v = *reinterpret_cast<const uint4 *>(&table[i]);
        key = ((u64)v.y << 32) | v.x;
        count = ((u64)v.w << 32) | v.z;
        return key != KH_EMPTY_KEY && count >= min_count;
    });
}

// count[key] += addend for n (key, addend) pairs: rehash-free merge of another table's pairs.
KH_GLOBAL __launch_bounds__(BLOCK) void table_merge_pairs_kernel(TableGeom tg, const u64 *keys,
                                                                  const u64 *counts, u64 n, Counters *ctr) {
    const u64 stride = (u64)gridDim.x * BLOCK;
    uint32_t nd = 0, nf = 0;
    u64 ad = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const u64 key = keys[i];
        const u64 c = counts[i];
        if (key != KH_EMPTY_KEY && c != 0) {
            upsert(tg, key, c, nd, nf);
            ad += c;
        }
    }
    u64 d = wave_sum((u64)nd), f = wave_sum((u64)nf);
    ad = wave_sum(ad);
    if (lane_id() == 0) {
        if (d) atomicAdd(&ctr->distinct, d);
        if (f) atomicAdd(&ctr->failed, f);
        if (ad) atomicAdd(&ctr->kmers, ad);  // (kmers == the sum of all counts in the table, merges included: round 5)
    }
}

// ---- dense form for small k (2k <= 26 bits of key space): element-wise reducible ------------------
// dense[key] = count of every live pair (dense[] is zeroed by the host first; keys are unique).
KH_GLOBAL __launch_bounds__(BLOCK) void table_to_dense_kernel(const Slot *__restrict__ table, u64 cap, u64 *__restrict__ dense,
                                                               u64 n_entries) {
    const u64 stride = (u64)gridDim.x * BLOCK;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < cap; i += stride) {
        const Slot s = table[i];
        if (s.key != KH_EMPTY_KEY && s.key < n_entries) dense[s.key] = s.count;
    }
}

// count[key] += dense[key] for every key with a non-zero entry that shard `owner` of `nparts` owns.
KH_GLOBAL __launch_bounds__(BLOCK) void table_merge_dense_kernel(TableGeom tg, const u64 *__restrict__ dense, u64 n_entries,
                                                                  uint32_t owner, uint32_t nparts, Counters *ctr) {
    const u64 stride = (u64)gridDim.x * BLOCK;
    uint32_t nd = 0, nf = 0;
    u64 ad = 0;
    for (u64 key = (u64)blockIdx.x * BLOCK + threadIdx.x; key < n_entries; key += stride) {
        const u64 c = dense[key];
        if (c != 0 && kh_owner_of(key, tg.k, nparts) == owner) {
            upsert(tg, key, c, nd, nf);
            ad += c;
        }
    }
    u64 d = wave_sum((u64)nd), f = wave_sum((u64)nf);
    ad = wave_sum(ad);
    if (lane_id() == 0) {
        if (d) atomicAdd(&ctr->distinct, d);
        if (f) atomicAdd(&ctr->failed, f);
        if (ad) atomicAdd(&ctr->kmers, ad);
    }
}

// Move every live pair of `old` into `nt` (table growth).
KH_GLOBAL __launch_bounds__(BLOCK) void table_rehash_kernel(const Slot *old, u64 oldcap, TableGeom tg,
                                                             Counters *ctr) {
    const u64 stride = (u64)gridDim.x * BLOCK;
    uint32_t nd = 0, nf = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < oldcap; i += stride) {
        const Slot s = old[i];
        if (s.key != KH_EMPTY_KEY) upsert(tg, s.key, s.count, nd, nf);
    }
    u64 f = wave_sum((u64)nf);
    if (lane_id() == 0 && f) atomicAdd(&ctr->failed, f);
}

KH_GLOBAL __launch_bounds__(BLOCK) void table_lookup_kernel(TableGeom tg, const u64 *keys, u64 n, u64 *out) {
    const u64 stride = (u64)gridDim.x * BLOCK;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const u64 key = keys[i];
        const u64 H = table_hash(tg, key);
        const Slot *reg = tg.table + region_of(tg, H) * REGION_SLOTS;
        uint32_t off = start_of(tg, H);
        u64 res = 0;
        for (uint32_t probes = 0; probes < REGION_SLOTS; ++probes) {
            const Slot s = reg[off];
            if (s.key == key) { res = s.count; break; }
            if (s.key == KH_EMPTY_KEY) break;
            off = (off + 1) & REGION_MASK;
        }
        out[i] = res;
    }
}

// Count-of-counts.  Counts below HIST_LDS are binned in LDS per workgroup and flushed once;
// counts below HIST_DENSE go to a dense global array; the (rare) rest are appended to `big`.
constexpr uint32_t HIST_LDS = 2048;
constexpr uint32_t HIST_DENSE = 1u << 16;

KH_GLOBAL __launch_bounds__(BLOCK) void table_hist_kernel(const Slot *table, u64 cap, u64 min_count, u64 *dense,
                                                           u64 *big, u64 big_cap, Counters *ctr) {
    __shared__ uint32_t s_bins[HIST_LDS];
    for (uint32_t i = threadIdx.x; i < HIST_LDS; i += BLOCK) s_bins[i] = 0;
    __syncthreads();
    // (counts 1, 2, 3 -- what nearly every key of a real table has -- by ballot and popcount in the wave's registers instead of
    //  sixty-four lanes' atomics on one LDS word: partition.hip.h ntable_hist_kernel)
    const u64 stride = (u64)gridDim.x * BLOCK;
    uint32_t w1 = 0, w2 = 0, w3 = 0;
    const u64 rounds = (cap + stride - 1) / stride;
    for (u64 r = 0; r < rounds; ++r) {  // (uniform trip count: every lane takes part in the ballots)
        const u64 i = r * stride + (u64)blockIdx.x * BLOCK + threadIdx.x;
        Slot s;
        s.key = KH_EMPTY_KEY;
        s.count = 0;
        if (i < cap) s = table[i];
        const bool live = s.key != KH_EMPTY_KEY && s.count >= min_count && s.count != 0;
        w1 += (uint32_t)__builtin_popcountll(kh_ballot(live && s.count == 1));
        w2 += (uint32_t)__builtin_popcountll(kh_ballot(live && s.count == 2));
        w3 += (uint32_t)__builtin_popcountll(kh_ballot(live && s.count == 3));
        if (!live || s.count <= 3) continue;
        if (s.count < HIST_LDS) {
            atomicAdd(&s_bins[(uint32_t)s.count], 1u);
        } else if (s.count < HIST_DENSE) {
            atomicAdd(&dense[s.count], 1ull);
        } else {
            const u64 o = atomicAdd(&ctr->big, 1ull);
            if (o < big_cap) big[o] = s.count;
        }
    }
    if (lane_id() == 0) {
        if (w1) atomicAdd(&s_bins[1], w1);
        if (w2) atomicAdd(&s_bins[2], w2);
        if (w3) atomicAdd(&s_bins[3], w3);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < HIST_LDS; i += BLOCK) {
        const uint32_t v = s_bins[i];
        if (v) atomicAdd(&dense[i], (u64)v);
    }
}

// ---------------------------------------------------------------------------------------------
// multi-GPU export: live pairs grouped by owner shard
// ---------------------------------------------------------------------------------------------
constexpr uint32_t MAX_PARTS = 256;

KH_GLOBAL __launch_bounds__(BLOCK) void owner_count_kernel(const Slot *table, u64 cap, uint32_t k, uint32_t nparts,
                                                            u64 *part_counts) {
    __shared__ uint32_t s_cnt[MAX_PARTS];
    for (uint32_t i = threadIdx.x; i < nparts; i += BLOCK) s_cnt[i] = 0;
    __syncthreads();
    const u64 stride = (u64)gridDim.x * BLOCK;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < cap; i += stride) {
        const Slot s = table[i];
        if (s.key != KH_EMPTY_KEY) atomicAdd(&s_cnt[kh_owner_of(s.key, k, nparts)], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nparts; i += BLOCK)
        if (s_cnt[i]) atomicAdd(&part_counts[i], (u64)s_cnt[i]);
}

// cursors[p] starts at the exclusive prefix of part_counts; wave-aggregated per owner.
KH_GLOBAL __launch_bounds__(BLOCK) void owner_scatter_kernel(const Slot *table, u64 cap, uint32_t k, uint32_t nparts,
                                                              u64 *cursors, u64 *keys, u64 *counts, u64 out_cap) {
    const u64 stride = (u64)gridDim.x * BLOCK;
    const u64 first = (u64)blockIdx.x * BLOCK + threadIdx.x;
    const u64 rounds = (cap + stride - 1) / stride;
    for (u64 r = 0; r < rounds; ++r) {
        const u64 i = first + r * stride;
        Slot s;
        s.key = KH_EMPTY_KEY;
        s.count = 0;
        if (i < cap) s = table[i];
        const bool live = (s.key != KH_EMPTY_KEY);
        const uint32_t own = live ? kh_owner_of(s.key, k, nparts) : 0xFFFFFFFFu;
        u64 todo = kh_ballot(live);
        while (todo) {  // one round per distinct owner present in the wave
            const int leader = __builtin_ctzll(todo);
            const uint32_t p = (uint32_t)__shfl((int)own, leader, 64);
            const u64 m = kh_ballot(live && own == p);
            u64 base = 0;
            if ((int)lane_id() == leader) base = atomicAdd(&cursors[p], (u64)__builtin_popcountll(m));
            base = __shfl(base, leader, 64);
            if (live && own == p) {
                const u64 o = base + mbcnt(m);
                if (o < out_cap) {
                    keys[o] = s.key;
                    counts[o] = s.count;
                }
            }
            todo &= ~m;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// deterministic synthetic reads: one lane per 16 output bytes, one 16-byte store each
// ---------------------------------------------------------------------------------------------
KH_GLOBAL __launch_bounds__(BLOCK) void synth_reads_kernel(u64 seed, u64 genome_len, uint32_t read_len,
                                                            u64 first_read, u64 n_reads, uint8_t *bases,
                                                            uint8_t *qual) {
    const u64 kg = kh_stream_key(seed, 0), ks = kh_stream_key(seed, 1), kd = kh_stream_key(seed, 2),
              ke = kh_stream_key(seed, 3);
    const u64 span = genome_len - read_len + 1;
    const u64 rstride = (u64)read_len + 1;
    const u64 total = n_reads * rstride;
    const u64 nchunks = (total + 15) / 16;
    const u64 gstride = (u64)gridDim.x * BLOCK;
    for (u64 c = (u64)blockIdx.x * BLOCK + threadIdx.x; c < nchunks; c += gstride) {
        u64 pos = c * 16;
        u64 i = pos / rstride;            // read index inside this buffer
        uint32_t j = (uint32_t)(pos - i * rstride);
        u64 r = first_read + i;
        u64 start = kh_draw(ks, r) % span;
        uint32_t strand = (uint32_t)(kh_draw(kd, r) & 1);
        uint32_t ob[4] = {0, 0, 0, 0}, oq[4] = {0, 0, 0, 0};
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            uint32_t bb = '\n', qq = '\n';
            if (pos + b < total && j < read_len) {
                u64 cc = strand ? 3 - (kh_draw(kg, start + (read_len - 1 - j)) & 3) : (kh_draw(kg, start + j) & 3);
                const u64 u = kh_draw(ke, r * read_len + j);
                const bool subst = (u & 0xFF) == 0;
                if (subst) cc = (cc + 1 + ((u >> 8) % 3)) & 3;
                const bool isn = ((u >> 16) & 0x3FF) == 0;
                bb = isn ? 'N' : (uint32_t)("ACGT"[cc]);
                const uint32_t v = (uint32_t)((u >> 32) & 0xFF);
                qq = v < 230 ? 'I' : (v < 250 ? '5' : '#');
                if (subst && ((u >> 40) & 1)) qq = '#';
            }
            ob[b >> 2] |= bb << (8 * (b & 3));
            oq[b >> 2] |= qq << (8 * (b & 3));
            if (++j == rstride) {  // crossed into the next read
                j = 0;
                ++r;
                start = kh_draw(ks, r) % span;
                strand = (uint32_t)(kh_draw(kd, r) & 1);
            }
        }
        if (pos + 16 <= total) {
            *reinterpret_cast<uint4 *>(bases + pos) = make_uint4(ob[0], ob[1], ob[2], ob[3]);
            if (qual) *reinterpret_cast<uint4 *>(qual + pos) = make_uint4(oq[0], oq[1], oq[2], oq[3]);
        } else {
            for (int b = 0; pos + b < total; ++b) {
                bases[pos + b] = (uint8_t)(ob[b >> 2] >> (8 * (b & 3)));
                if (qual) qual[pos + b] = (uint8_t)(oq[b >> 2] >> (8 * (b & 3)));
            }
        }
    }
}

#endif  // KH_HELPERS_ONLY

}  // namespace kh
