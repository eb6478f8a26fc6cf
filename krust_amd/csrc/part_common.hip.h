// part_common.hip.h -- geometry, payload formats and the chunk pool of the partitioned path: what level 1 (level1.hip.h,
// compiled in its own translation units) and the later stages (partition.hip.h) share.  See partition.hip.h for the design.
#pragma once
#include "kernels.hip.h"

#ifndef KH_ABL3
#define KH_ABL3 0  // the same for part2_scatter_lines_kernel
#endif
#ifndef KH_ABL2
#define KH_ABL2 0  // the same for part2_scatter_kernel
#endif
#ifndef KH_ABL_ARENA
#define KH_ABL_ARENA 0  // timing experiments on part2_arena_kernel (bit 0: no barrier B2, bit 1: no barrier B1); 0 in any product build
#endif
#ifndef KH_ABLR
#define KH_ABLR 0  // the same for region_count_kernel32
#endif
#ifndef KH_ABL
#define KH_ABL 0  // ablation bits for timing experiments on part1_scatter_chunked_kernel (tools/p1_ablation.sh); 0 in any product build
#endif

namespace kh {

constexpr int PART_NT = 1024;                    // lanes per workgroup in the partition kernels
constexpr int PART_TILE = PART_NT * CHUNK;       // 16384 positions / keys per batch
constexpr uint32_t MAX_P1 = 1024;
constexpr uint32_t MAX_P1_BITS = 10;
constexpr uint32_t MAX_P2_BITS = 10;             // <= 1024 regions per level-1 partition
constexpr uint32_t MAX_B2 = 1u << MAX_P2_BITS;
constexpr u64 PART2_CHUNK = 16ull * PART_TILE;   // payloads per level-2 workgroup (262144)
#ifndef KH_REGION_NT
#define KH_REGION_NT 1024  // (512: three workgroups per CU instead of two -- A/B builds)
#endif
constexpr int REGION_NT = KH_REGION_NT;          // lanes per workgroup in region_count_kernel
#ifndef KH_ARENA_LANES
#define KH_ARENA_LANES 4  // lanes that flush a bucket of the arena level 2 together (1, 2, 8: A/B builds)
#endif
#ifndef KH_REGION_RK
#define KH_REGION_RK 8
#endif
constexpr int REGION_RK = KH_REGION_RK;                     // keys prefetched per lane per round

struct PartGeom {
    uint32_t p1_bits;      // level-1 partitions = 1 << p1_bits
    uint32_t b2;           // buckets per level-1 partition (ANY number <= 1024 since round 4; regions = b2 << p1_bits).  For a
                           // table with a power-of-two region count any split describes the same layout; otherwise
                           // (p1_bits, b2) is the table's own geometry (kernels.hip.h TableGeom, kh_geom_of_regions)
    u64 b2_magic;          // ceil(2^40 / b2): r / b2 == (r * b2_magic) >> 40 for every region index r < 2^22 (part_div_b2)
    uint32_t defer;        // 1: the pool's 32-bit payloads lack the last Feistel round (hash_p1_pay32 below): level 2 finishes them
    uint32_t p2_bits;      // log2(b2) where b2 is a power of two (the hot kernels then take digit and start by shifts: their POW2
                           // instances are rounds 1-3's code, instruction for instruction), 0xFFFFFFFF otherwise
    uint32_t k;
    uint32_t shard_shift;  // as TableGeom: placement hash = kh_table_hash << shard_shift
    uint32_t shard_index;
};
__host__ __device__ inline u64 part_regions(const PartGeom &g) { return (u64)g.b2 << g.p1_bits; }
__host__ inline u64 part_magic_of(uint32_t b2) { return ((1ull << 40) + b2 - 1) / b2; }
__host__ inline uint32_t part_p2_bits_of(uint32_t b2) {
    if (b2 & (b2 - 1)) return 0xFFFFFFFFu;
    uint32_t b = 0;
    while ((1u << b) < b2) ++b;
    return b;
}
// bucket of a 32-bit payload: POW2 -- the top p2_bits bits; else the fast-range (kernels.hip.h)
template <bool POW2>
__device__ __forceinline__ uint32_t part_bucket32(uint32_t pay, const PartGeom &g) {
    if constexpr (POW2) return g.p2_bits ? (pay >> (32 - g.p2_bits)) : 0u;
    else return kh_bucket_of_x(pay, g.b2);
}
// region index -> (level-1 partition, bucket)
__device__ __forceinline__ uint32_t part_div_b2(const PartGeom &g, u64 r) { return (uint32_t)((r * g.b2_magic) >> 40); }
__device__ __forceinline__ RegionGeom rgeom(const PartGeom &g) { return RegionGeom{g.p1_bits, g.b2}; }

template <int MODE = KH_MUL_AUTO>
__device__ __forceinline__ u64 part_hash(const PartGeom &g, u64 key) { return kh_table_hash<MODE>(key, g.k) << g.shard_shift; }
// key of a placement hash of this (possibly sharded) table
__device__ __forceinline__ u64 part_unhash(const PartGeom &g, u64 Hs) {
    const u64 H = g.shard_shift ? ((Hs >> g.shard_shift) | ((u64)g.shard_index << (64 - g.shard_shift))) : Hs;
    return kh_table_unhash(H, g.k);
}

// ---- payload traits ---------------------------------------------------------------------------
template <typename PT>
struct Pay;

// Round 6: the 8-byte payload is the HASH below the level-1 digit, left-aligned (H << p1_bits) -- what the 4-byte payload has
// always been, 32 bits wider -- not the key.  Rounds 1-5 carried the canonical key through the partition buffers and hashed it
// three times: level 1 (the digit), level 2 (the bucket: Pay<u64>::p2), the region pass (the in-region start), each a four-round
// Feistel per OCCURRENCE -- configs[2]'s region pass spent a quarter of its vector instructions there.  The hash is a
// bijection: level 2's bucket is a fast-range of the payload's top word, the region pass probes on the payload and inverts the
// hash once per NEW key at the write-back.  Costs level 1 three instructions per window (window.hip.h win_hash64: the
// payload is put together from the two halves instead of falling out of the canonical choice).
// KH_EMPTY_KEY (all ones) still pads 8-byte segments: a payload's low bit is zero -- make_geom gives 8-byte payloads at least
// one level-1 bit.
template <>
struct Pay<u64> {
    __device__ static __forceinline__ u64 make(u64 key, u64 H, const PartGeom &g) { return H << g.p1_bits; }
    __device__ static __forceinline__ uint32_t p2(u64 pay, const PartGeom &g) { return kh_bucket_of_x((uint32_t)(pay >> 32), g.b2); }
    __device__ static __forceinline__ u64 hash(u64 pay, uint32_t p1, const PartGeom &g) {
        return g.p1_bits ? (((u64)p1 << (64 - g.p1_bits)) | (pay >> g.p1_bits)) : pay;
    }
    __device__ static __forceinline__ u64 key(u64 pay, uint32_t p1, const PartGeom &g) { return part_unhash(g, hash(pay, p1, g)); }
};

template <>
struct Pay<uint32_t> {  // x: bits [p1_bits, p1_bits+32) of H
    __device__ static __forceinline__ uint32_t make(u64 key, u64 H, const PartGeom &g) { return kh_x_of(H, g.p1_bits); }
    __device__ static __forceinline__ uint32_t p2(uint32_t pay, const PartGeom &g) { return kh_bucket_of_x(pay, g.b2); }
    __device__ static __forceinline__ u64 hash(uint32_t pay, uint32_t p1, const PartGeom &g) {
        const u64 top = g.p1_bits ? ((u64)p1 << (64 - g.p1_bits)) : 0ull;
        return top | ((u64)pay << (32 - g.p1_bits));
    }
    __device__ static __forceinline__ u64 key(uint32_t pay, uint32_t p1, const PartGeom &g) {
        return part_unhash(g, hash(pay, p1, g));
    }
};

__device__ __forceinline__ uint32_t p1_of_hash(u64 H, const PartGeom &g) {
    return g.p1_bits ? (uint32_t)(H >> (64 - g.p1_bits)) : 0u;
}

// Level-1 digit and 32-bit payload straight from the two k-bit halves of the hash, h = L << k | R:
// p1 = the top p1_bits of h, payload = the remaining 2k - p1_bits (<= 32) bits, left-aligned.  All
// 32-bit shifts (the generic form above costs five 64-bit shifts per window).  Valid iff the table is
// not a shard, p1_bits <= k and 1 <= 2k - p1_bits <= 32 (p1_fast_ok); same values as the generic form.
__host__ __device__ inline bool p1_fast_ok(const PartGeom &g) {
    return g.shard_shift == 0 && g.p1_bits <= g.k && 2 * g.k - g.p1_bits >= 1 && 2 * g.k - g.p1_bits <= 32 && g.k < 32;
}
// Round 4: THE LAST FEISTEL ROUND IS LEVEL 2'S.  Written as a ^= F(b), b ^= G(a), a ^= F(b), b ^= G(a) (kh_feistel_f / _g) on (a, b) = (L, R) of
// the key, the hash is a << k | b -- and a, which holds the level-1 digit and everything level 1's addresses are made of, is
// final after the THIRD round; the fourth only changes b.  Level 1 is bound by its instruction stream (21 of its 26.5 ms are
// VALU issue), level 2 by the memory system with the VALU idle half the time: so a level-1 kernel on the FAST path stores
// y = (a_low ++ b') -- the payload with b one round short -- and whoever reads the pool (the level-2 kernels: l2_finish)
// applies x = y ^ G(a) << ..., a = digit ++ a_low, which is the payload defined above, bit for bit.  PartGeom::defer says so.
// MEASURED (profiles/README.md r04, A/B builds, S100M): level 1 26.43 -> 25.42 ms at k = 21 (25.8 at k = 17, 26.9 at k = 13: one
// round of twelve instructions fewer per window, as predicted) -- and level 2 21.8 -> 23.8 ms: the arena kernel is at its 128
// registers with 14 spilled, and five more vector instructions per payload cost it more than level 1 gained.  Net +1 ms per
// step, so the DEFAULT IS OFF (0); the code stays as the measurement's record and for a level 2 with registers to spare.
#ifndef KH_L1_DEFER_ROUND
#define KH_L1_DEFER_ROUND 0
#endif
template <int MODE>
__device__ __forceinline__ void hash_p1_pay32(const uint32_t k, const uint32_t p1_bits, u64 key, uint32_t &p1, uint32_t &pay) {
    const uint32_t mask = (1u << k) - 1u;
    uint32_t a = (uint32_t)(key >> k) & mask, b = (uint32_t)key & mask;
    a = (a ^ kh_feistel_f<MODE>(b, KH_FC0, k)) & mask;
    b = (b ^ kh_feistel_g(a, KH_FC1, k)) & mask;
    a = (a ^ kh_feistel_f<MODE>(b, KH_FC2, k)) & mask;
#if !KH_L1_DEFER_ROUND
    b = (b ^ kh_feistel_g(a, KH_FC3, k)) & mask;
#endif
    p1 = a >> (k - p1_bits);
    pay = ((a << k) | b) << (32u - (2u * k - p1_bits));  // a's top p1_bits fall off the 32-bit word
}
// what level 2 does to a deferred payload y of level-1 partition p1 (valid under p1_fast_ok)
__device__ __forceinline__ uint32_t pay32_finish(uint32_t y, uint32_t p1, uint32_t k, uint32_t p1_bits) {
    const uint32_t nb = 2u * k - p1_bits;  // significant bits of the payload, left-aligned in 32
    const uint32_t alow = k > p1_bits ? y >> (32u - (k - p1_bits)) : 0u;
    const uint32_t a = (p1 << (k - p1_bits)) | alow;
    return y ^ (kh_feistel_g(a, KH_FC3, k) << (32u - nb));
}
// ---- level-2 work unit and the two level-1 output layouts it can read -----------------------------
struct Part2Block {
    u64 lo, hi;        // dense source: payload range in the level-1 output;
                       // chunked source: range of the partition's chunk list (plist indices)
    u64 mbase;         // H2/O2 index of (p2 = 0, this chunk)
    uint32_t mstride;  // chunks in this partition: H2 index of p2 is mbase + p2 * mstride
    uint32_t p1;
};

// ---- level-1 output as a pool of fixed-size chunks ------------------------------------------------
// Level 1 can then run in ONE pass: no counting pass is needed to know where a partition's data
// goes, a workgroup just takes the next free chunk when a partition's current chunk fills up.
// Chunks are handed out in per-workgroup ranges (one global atomic per POOL_GRAB chunks; a lone
// pool counter hit once per chunk would serialise at ~6-17 ns per same-address atomic).
constexpr uint32_t CHUNK_PAY = 256;     // payloads per pool chunk (1 KiB)
constexpr uint32_t POOL_GRAB = 4096;    // chunks per workgroup grab (4 MiB of payloads)
constexpr uint32_t POOL_LOW = 72;       // refill the private range below this many free chunks
constexpr uint32_t CPB = 1024;          // chunks per level-2 workgroup (262144 payloads)
constexpr uint16_t PART_NONE = 0xFFFFu; // chunk_part[] of a chunk nobody owns

struct ChunkSrc {                 // how level 2 reads a chunked level-1 output
    const void *pay;              // pool (PT payloads)
    const uint32_t *plist;        // chunk ids ordered by partition
    const uint8_t *fill8;         // payloads in the chunk minus one
};

// A level-2 workgroup first copies its slice of the chunk list (ids and fill levels) into LDS, so
// that fetching element e is ONE global load again (plist -> fill8 -> payload would be a chain of
// three dependent loads per element).
template <bool CHUNKED>
__device__ __forceinline__ void p2_stage_chunks(const ChunkSrc &cs, const Part2Block &pb, uint32_t *s_chk, uint16_t *s_cfill,
                                                int tid, int nthreads) {
    if (!CHUNKED) return;
    const uint32_t nc = (uint32_t)(pb.hi - pb.lo);
    for (uint32_t i = tid; i < nc; i += nthreads) {
        const uint32_t chunk = cs.plist[pb.lo + i];
        s_chk[i] = chunk;
        s_cfill[i] = (uint16_t)((uint32_t)cs.fill8[chunk] + 1u);
    }
}

// element e of a level-2 workgroup's input; returns false past the data
template <bool CHUNKED, typename PT>
__device__ __forceinline__ bool p2_load(const PT *__restrict__ dense, const ChunkSrc &cs, const Part2Block &pb,
                                        const uint32_t *s_chk, const uint16_t *s_cfill, uint32_t e, uint32_t n, PT &out) {
    if (!CHUNKED) {
        out = dense[pb.lo + (e < n ? e : n - 1)];
        return e < n;
    }
    const uint32_t ec = e < n ? e : n - 1;
    const uint32_t chunk = s_chk[ec >> 8];
    const uint32_t off = ec & (CHUNK_PAY - 1);
    const uint32_t have = s_cfill[ec >> 8];
    out = reinterpret_cast<const PT *>(cs.pay)[(u64)chunk * CHUNK_PAY + (off < have ? off : 0)];
    return e < n && off < have;  // (the caller finishes the payload: l2_finish)
}
// a payload as read from the level-1 pool -> the payload every later stage means (see hash_p1_pay32)
template <typename PT>
__device__ __forceinline__ PT l2_finish(PT v, uint32_t p1, const PartGeom &g) {
    if constexpr (sizeof(PT) == 4) return (KH_L1_DEFER_ROUND && g.defer) ? (PT)pay32_finish((uint32_t)v, p1, g.k, g.p1_bits) : v;
    else return v;
}
template <bool CHUNKED>
__device__ __forceinline__ uint32_t p2_count_of(const Part2Block &pb) {
    return CHUNKED ? (uint32_t)(pb.hi - pb.lo) * CHUNK_PAY : (uint32_t)(pb.hi - pb.lo);
}

// Exclusive scan of s_cnt[0..N) into s_lofs[0..N) (N = 512 or 1024) by a workgroup of >= N / 4 lanes.
// s_wsum: 4 words of scratch.  Ends with a barrier.
template <int N, typename LT>
__device__ __forceinline__ void block_exclusive_scan_n(const uint32_t *s_cnt, LT *s_lofs, uint32_t *s_wsum, int tid) {
    constexpr int LANES = N / 4;
    uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0, incl = 0;
    if (tid < LANES) {
        v0 = s_cnt[4 * tid];
        v1 = s_cnt[4 * tid + 1];
        v2 = s_cnt[4 * tid + 2];
        v3 = s_cnt[4 * tid + 3];
        incl = v0 + v1 + v2 + v3;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t n = __shfl_up(incl, off, 64);
            if ((tid & 63) >= off) incl += n;
        }
        if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
    }
    __syncthreads();
    if (tid < LANES) {
        uint32_t base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += s_wsum[w];
        const uint32_t excl = base + incl - (v0 + v1 + v2 + v3);
        s_lofs[4 * tid] = (LT)excl;
        s_lofs[4 * tid + 1] = (LT)(excl + v0);
        s_lofs[4 * tid + 2] = (LT)(excl + v0 + v1);
        s_lofs[4 * tid + 3] = (LT)(excl + v0 + v1 + v2);
    }
    __syncthreads();
}
template <typename LT>
__device__ __forceinline__ void block_exclusive_scan_1024(const uint32_t *s_cnt, LT *s_lofs, uint32_t *s_wsum, int tid) {
    block_exclusive_scan_n<1024, LT>(s_cnt, s_lofs, s_wsum, tid);
}

}  // namespace kh
