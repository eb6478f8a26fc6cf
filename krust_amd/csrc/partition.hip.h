// partition.hip.h -- the locality stage of the counting path (gfx950, wave64).
//
// The direct path (count_direct_kernel) issues one memory-side atomic per k-mer and saturates the
// chip's atomic request rate (~18.5 G/s measured, profiles/r01a).  This path issues NO global
// atomics per k-mer: canonical keys are radix-partitioned by the high bits of their table hash in
// two levels (level 1: P1 <= 1024 partitions straight from the bases; level 2: P2 <= 1024 buckets
// inside each level-1 partition), so that every bucket holds exactly the keys of ONE table region
// (kernels.hip.h: 4096 slots = 64 KiB).  One workgroup then rebuilds each region in LDS with LDS
// atomics and writes it back with coalesced 16-byte stores.
//
// Both partition levels are "count, scan, scatter" with deterministic offsets (no global cursor
// atomics): workgroup b owns a fixed contiguous range of its input, the count pass writes its
// per-partition histogram as a column of a [partition][workgroup] matrix, an exclusive scan of the
// flattened matrix yields every (partition, workgroup) output offset, and the scatter pass
// re-reads the same range, counting-sorts each batch in LDS and writes per-partition runs.
//
// Reference counterpart: none (the reference is src/run.rs:526-571 + DashMap); results are the
// same multiset of (key,count) as the direct path.
#pragma once
#include "kernels.hip.h"

namespace kh {

constexpr int PART_NT = 1024;                    // lanes per workgroup in the partition kernels
constexpr int PART_TILE = PART_NT * CHUNK;       // 16384 positions / keys per batch
constexpr uint32_t MAX_P1 = 1024;
constexpr uint32_t MAX_P2_BITS = 10;             // P2 <= 1024 regions per level-1 partition
constexpr u64 PART2_CHUNK = 16ull * PART_TILE;   // keys per level-2 workgroup (262144)
constexpr int REGION_NT = 1024;                  // lanes per workgroup in region_count_kernel
constexpr int REGION_RK = 8;                     // keys prefetched per lane per round

struct PartGeom {
    u64 nregions;      // R
    uint32_t p2_bits;  // P2 = 1 << p2_bits
    uint32_t P1;       // ceil(R / P2)
};

__device__ __forceinline__ uint32_t p1_of_key(u64 key, const PartGeom &g) {
    return (uint32_t)(region_of_hash(kh_mix64(key), g.nregions) >> g.p2_bits);
}
__device__ __forceinline__ uint32_t p2_of_key(u64 key, const PartGeom &g) {
    return (uint32_t)region_of_hash(kh_mix64(key), g.nregions) & ((1u << g.p2_bits) - 1u);
}

// Exclusive scan of s_cnt[0..1024) into s_lofs[0..1024) by a workgroup of >= 256 lanes.
// s_wsum: 4 words of scratch.  Ends with a barrier.
__device__ __forceinline__ void block_exclusive_scan_1024(const uint32_t *s_cnt, uint32_t *s_lofs, uint32_t *s_wsum,
                                                          int tid) {
    uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0, incl = 0;
    if (tid < 256) {
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
    if (tid < 256) {
        uint32_t base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += s_wsum[w];
        const uint32_t excl = base + incl - (v0 + v1 + v2 + v3);
        s_lofs[4 * tid] = excl;
        s_lofs[4 * tid + 1] = excl + v0;
        s_lofs[4 * tid + 2] = excl + v0 + v1;
        s_lofs[4 * tid + 3] = excl + v0 + v1 + v2;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// level 1, pass A: per-workgroup histogram of level-1 partition ids, straight from the bases
// ---------------------------------------------------------------------------------------------
// H1 layout: [p1][workgroup]  (nblocks = gridDim.x columns)
template <bool QUAL>
__global__ __launch_bounds__(PART_NT) void part1_count_kernel(
    const uint8_t *__restrict__ abase, const uint8_t *__restrict__ qbase, int qaligned, u64 vbeg, u64 vend, u64 wlo,
    u64 tile0, u64 ntiles, uint32_t tiles_per_block, uint32_t k, uint32_t thr, PartGeom g, uint32_t *__restrict__ H1) {
    __shared__ uint32_t s_code[2][PART_NT + 2];
    __shared__ uint16_t s_val[2][PART_NT + 2];
    __shared__ uint32_t s_hist[MAX_P1];
    const int tid = threadIdx.x;
    s_hist[tid] = 0;  // PART_NT == MAX_P1
    const u64 tb = tile0 + (u64)blockIdx.x * tiles_per_block;
    u64 te = tb + tiles_per_block;
    if (te > tile0 + ntiles) te = tile0 + ntiles;
    const u64 kmask = kh_kmask(k), vmask = valid_mask_of(k);
    int buf = 0;
    __syncthreads();
    for (u64 t = tb; t < te; ++t, buf ^= 1) {
        const WinCtx w = stage_tile<QUAL, PART_NT>(s_code, s_val, buf, t == tb, tid, abase, qbase, qaligned, t, vbeg, vend, thr);
#pragma unroll
        for (int j = 0; j < CHUNK; ++j) {
            u64 key;
            if (window_key(w, j, kmask, vmask, k, wlo, key)) atomicAdd(&s_hist[p1_of_key(key, g)], 1u);
        }
    }
    __syncthreads();
    if ((uint32_t)tid < g.P1) H1[(u64)tid * gridDim.x + blockIdx.x] = s_hist[tid];
}

// ---------------------------------------------------------------------------------------------
// level 1, pass B: scatter keys into level-1 partitions.  O1 = exclusive scan of H1 (flattened).
// ---------------------------------------------------------------------------------------------
template <bool QUAL>
__global__ __launch_bounds__(PART_NT) void part1_scatter_kernel(
    const uint8_t *__restrict__ abase, const uint8_t *__restrict__ qbase, int qaligned, u64 vbeg, u64 vend, u64 wlo,
    u64 tile0, u64 ntiles, uint32_t tiles_per_block, uint32_t k, uint32_t thr, PartGeom g,
    const u64 *__restrict__ O1, u64 *__restrict__ out) {
    __shared__ uint32_t s_code[2][PART_NT + 2];
    __shared__ uint16_t s_val[2][PART_NT + 2];
    __shared__ u64 s_stage[PART_TILE];  // 128 KiB
    __shared__ uint32_t s_cnt[MAX_P1];
    __shared__ uint32_t s_lofs[MAX_P1];
    __shared__ u64 s_gcur[MAX_P1];
    __shared__ uint32_t s_wsum[4];
    const int tid = threadIdx.x;
    s_cnt[tid] = 0;
    s_gcur[tid] = ((uint32_t)tid < g.P1) ? O1[(u64)tid * gridDim.x + blockIdx.x] : 0;
    const u64 tb = tile0 + (u64)blockIdx.x * tiles_per_block;
    u64 te = tb + tiles_per_block;
    if (te > tile0 + ntiles) te = tile0 + ntiles;
    const u64 kmask = kh_kmask(k), vmask = valid_mask_of(k);
    int buf = 0;
    __syncthreads();
    RawChunk raw = load_raw<QUAL>(abase, qbase, qaligned, chunk_pos<PART_NT>(tb, tid), vbeg, tb < te ? vend : 0);
    for (u64 t = tb; t < te; ++t, buf ^= 1) {
        const WinCtx w = stage_tile_raw<QUAL, PART_NT>(s_code, s_val, buf, t == tb, tid, raw, abase, qbase, qaligned, t, vbeg, vend, thr);
        u64 key[CHUNK];
        uint32_t tag[CHUNK];  // (p1 << 16) | rank-in-partition, 0xFFFFFFFF = no key
#pragma unroll
        for (int j = 0; j < CHUNK; ++j) {
            tag[j] = 0xFFFFFFFFu;
            if (window_key(w, j, kmask, vmask, k, wlo, key[j])) {
                const uint32_t p = p1_of_key(key[j], g);
                tag[j] = (p << 16) | atomicAdd(&s_cnt[p], 1u);  // rank < 16384 fits 16 bits
            }
        }
        __syncthreads();
        block_exclusive_scan_1024(s_cnt, s_lofs, s_wsum, tid);
#pragma unroll
        for (int j = 0; j < CHUNK; ++j)
            if (tag[j] != 0xFFFFFFFFu) s_stage[s_lofs[tag[j] >> 16] + (tag[j] & 0xFFFFu)] = key[j];
        __syncthreads();
        // next tile's bases are fetched while this tile's runs are written out
        raw = load_raw<QUAL>(abase, qbase, qaligned, chunk_pos<PART_NT>(t + 1, tid), vbeg, t + 1 < te ? vend : 0);
        const uint32_t total = s_lofs[MAX_P1 - 1] + s_cnt[MAX_P1 - 1];
        for (uint32_t i = tid; i < total; i += PART_NT) {
            const u64 kk = s_stage[i];
            const uint32_t p = p1_of_key(kk, g);
            out[s_gcur[p] + (i - s_lofs[p])] = kk;  // consecutive lanes -> consecutive addresses inside a run
        }
        __syncthreads();
        s_gcur[tid] += s_cnt[tid];
        s_cnt[tid] = 0;
        // the next tile's stage_tile() barrier orders these updates before its atomics
    }
}

// ---------------------------------------------------------------------------------------------
// level 2 work list: one workgroup per PART2_CHUNK keys of a level-1 partition
// ---------------------------------------------------------------------------------------------
struct Part2Block {
    u64 lo, hi;        // key range in the level-1 output
    u64 mbase;         // H2/O2 index of (p2 = 0, this chunk)
    uint32_t mstride;  // chunks in this partition: H2 index of p2 is mbase + p2 * mstride
    uint32_t p1;
};


// ---------------------------------------------------------------------------------------------
// generic exclusive scan u32 -> u64 (three small kernels)
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_NT = 256;
constexpr int SCAN_PER = 16;
constexpr int SCAN_CHUNK = SCAN_NT * SCAN_PER;  // 4096 entries per workgroup

__global__ __launch_bounds__(SCAN_NT) void scan_partials_kernel(const uint32_t *__restrict__ in, u64 n, u64 *__restrict__ partial) {
    __shared__ u64 s_w[SCAN_NT / 64];
    const u64 base = (u64)blockIdx.x * SCAN_CHUNK;
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_PER; ++i) {
        const u64 idx = base + (u64)i * SCAN_NT + threadIdx.x;
        if (idx < n) s += in[idx];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// single workgroup: in-place exclusive scan of partial[0..nb), total -> partial[nb]
__global__ __launch_bounds__(1024) void scan_spine_kernel(u64 *__restrict__ partial, u64 nb) {
    __shared__ u64 s_w[16];
    __shared__ u64 s_carry;
    const int tid = threadIdx.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (u64 base = 0; base < nb; base += 1024) {
        const u64 idx = base + tid;
        const u64 v = idx < nb ? partial[idx] : 0;
        u64 incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u64 n = __shfl_up(incl, off, 64);
            if ((tid & 63) >= off) incl += n;
        }
        if ((tid & 63) == 63) s_w[tid >> 6] = incl;
        __syncthreads();
        u64 wbase = s_carry;
        for (int w = 0; w < (tid >> 6); ++w) wbase += s_w[w];
        if (idx < nb) partial[idx] = wbase + incl - v;
        __syncthreads();
        if (tid == 1023) s_carry = wbase + incl;
        __syncthreads();
    }
    if (tid == 0) partial[nb] = s_carry;
}

__global__ __launch_bounds__(SCAN_NT) void scan_apply_kernel(const uint32_t *__restrict__ in, u64 n,
                                                             const u64 *__restrict__ partial, u64 *__restrict__ out) {
    __shared__ u64 s_w[SCAN_NT / 64];
    const int tid = threadIdx.x;
    // lane owns SCAN_PER consecutive entries
    const u64 first = (u64)blockIdx.x * SCAN_CHUNK + (u64)tid * SCAN_PER;
    uint32_t v[SCAN_PER];
    u64 sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_PER; ++i) {
        v[i] = (first + i < n) ? in[first + i] : 0u;
        sum += v[i];
    }
    u64 incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u64 nn = __shfl_up(incl, off, 64);
        if ((tid & 63) >= off) incl += nn;
    }
    if ((tid & 63) == 63) s_w[tid >> 6] = incl;
    __syncthreads();
    u64 run = partial[blockIdx.x] + incl - sum;
    for (int w = 0; w < (tid >> 6); ++w) run += s_w[w];
#pragma unroll
    for (int i = 0; i < SCAN_PER; ++i) {
        if (first + i < n) out[first + i] = run;
        run += v[i];
    }
    if (blockIdx.x == gridDim.x - 1 && tid == SCAN_NT - 1) out[n] = run;  // grand total
}

// ---------------------------------------------------------------------------------------------
// level 2 plan.  Single workgroup.  Level-1 partition p1 starts at O1[p1 * o1_stride]; the grand
// total is O1[o1_total_index].  Writes the block list, info[0] = number of blocks, info[1] = H2
// entries used, info[2] = total keys, moff[p1] = H2 index of (p1, p2 = 0, chunk 0), nch[p1] =
// chunks of p1.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void part2_plan_kernel(const u64 *__restrict__ O1, uint32_t o1_stride, PartGeom g,
                                                          u64 o1_total_index, Part2Block *__restrict__ blocks,
                                                          u64 max_blocks, u64 *__restrict__ moff,
                                                          uint32_t *__restrict__ nch, u64 *__restrict__ info) {
    __shared__ u64 s_seg[MAX_P1 + 1];
    __shared__ u64 s_bbase[MAX_P1 + 1];
    const int tid = threadIdx.x;
    if ((uint32_t)tid < g.P1) s_seg[tid] = O1[(u64)tid * o1_stride];
    if (tid == 0) s_seg[g.P1] = O1[o1_total_index];
    __syncthreads();
    if (tid == 0) {  // P1 <= 1024: a serial prefix is cheap
        u64 b = 0;
        for (uint32_t p = 0; p < g.P1; ++p) {
            s_bbase[p] = b;
            b += (s_seg[p + 1] - s_seg[p] + PART2_CHUNK - 1) / PART2_CHUNK;
        }
        s_bbase[g.P1] = b;
        info[0] = b;
        info[1] = b << g.p2_bits;
        info[2] = s_seg[g.P1];
    }
    __syncthreads();
    if ((uint32_t)tid < g.P1) {
        const u64 lo = s_seg[tid], hi = s_seg[tid + 1];
        const u64 b0 = s_bbase[tid];
        const uint32_t n = (uint32_t)(s_bbase[tid + 1] - b0);
        moff[tid] = b0 << g.p2_bits;
        nch[tid] = n;
        for (uint32_t c = 0; c < n; ++c) {
            if (b0 + c >= max_blocks) break;
            Part2Block pb;
            pb.lo = lo + (u64)c * PART2_CHUNK;
            pb.hi = pb.lo + PART2_CHUNK < hi ? pb.lo + PART2_CHUNK : hi;
            pb.mbase = (b0 << g.p2_bits) + c;
            pb.mstride = n;
            pb.p1 = (uint32_t)tid;
            blocks[b0 + c] = pb;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// level 2, pass A: histogram of bucket ids (p2) per workgroup.  H2 must be zero-filled.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PART_NT) void part2_count_kernel(const u64 *__restrict__ keys, const Part2Block *__restrict__ blocks,
                                                              const u64 *__restrict__ info, PartGeom g,
                                                              uint32_t *__restrict__ H2) {
    __shared__ uint32_t s_hist[1u << MAX_P2_BITS];
    if ((u64)blockIdx.x >= info[0]) return;
    const Part2Block pb = blocks[blockIdx.x];
    const int tid = threadIdx.x;
    s_hist[tid] = 0;  // PART_NT == 1 << MAX_P2_BITS
    __syncthreads();
    for (u64 i = pb.lo + tid; i < pb.hi; i += PART_NT) atomicAdd(&s_hist[p2_of_key(keys[i], g)], 1u);
    __syncthreads();
    if ((uint32_t)tid < (1u << g.p2_bits)) H2[pb.mbase + (u64)tid * pb.mstride] = s_hist[tid];
}

// ---------------------------------------------------------------------------------------------
// level 2, pass B: scatter into buckets (one bucket == one table region)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PART_NT) void part2_scatter_kernel(const u64 *__restrict__ keys, const Part2Block *__restrict__ blocks,
                                                                const u64 *__restrict__ info, PartGeom g,
                                                                const u64 *__restrict__ O2, u64 *__restrict__ out) {
    __shared__ u64 s_stage[PART_TILE];  // 128 KiB
    __shared__ uint32_t s_cnt[MAX_P1];  // only the first P2 entries are used; sized for the shared scan
    __shared__ uint32_t s_lofs[MAX_P1];
    __shared__ u64 s_gcur[1u << MAX_P2_BITS];
    __shared__ uint32_t s_wsum[4];
    if ((u64)blockIdx.x >= info[0]) return;
    const Part2Block pb = blocks[blockIdx.x];
    const int tid = threadIdx.x;
    const uint32_t P2 = 1u << g.p2_bits;
    s_cnt[tid] = 0;
    if ((uint32_t)tid < P2) s_gcur[tid] = O2[pb.mbase + (u64)tid * pb.mstride];
    __syncthreads();
    u64 key[CHUNK];
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) {  // lane-contiguous: coalesced 8-byte loads
        const u64 i = pb.lo + (u64)j * PART_NT + tid;
        key[j] = i < pb.hi ? keys[i] : KH_EMPTY_KEY;
    }
    for (u64 base = pb.lo; base < pb.hi; base += PART_TILE) {
        uint32_t tag[CHUNK];
#pragma unroll
        for (int j = 0; j < CHUNK; ++j) {
            tag[j] = 0xFFFFFFFFu;
            if (key[j] != KH_EMPTY_KEY) {
                const uint32_t p = p2_of_key(key[j], g);
                tag[j] = (p << 16) | atomicAdd(&s_cnt[p], 1u);
            }
        }
        __syncthreads();
        block_exclusive_scan_1024(s_cnt, s_lofs, s_wsum, tid);
#pragma unroll
        for (int j = 0; j < CHUNK; ++j)
            if (tag[j] != 0xFFFFFFFFu) s_stage[s_lofs[tag[j] >> 16] + (tag[j] & 0xFFFFu)] = key[j];
        __syncthreads();
        // next batch's keys are fetched while this batch's runs are written out
#pragma unroll
        for (int j = 0; j < CHUNK; ++j) {
            const u64 i = base + PART_TILE + (u64)j * PART_NT + tid;
            key[j] = i < pb.hi ? keys[i] : KH_EMPTY_KEY;
        }
        const uint32_t total = s_lofs[MAX_P1 - 1] + s_cnt[MAX_P1 - 1];
        for (uint32_t i = tid; i < total; i += PART_NT) {
            const u64 kk = s_stage[i];
            const uint32_t p = p2_of_key(kk, g);
            out[s_gcur[p] + (i - s_lofs[p])] = kk;
        }
        __syncthreads();
        if ((uint32_t)tid < P2) s_gcur[tid] += s_cnt[tid];
        s_cnt[tid] = 0;
        __syncthreads();
    }
}

// bstart[r] = first key of region r's bucket in the level-2 output, r in [0, R]; bstart[R] = total
__global__ __launch_bounds__(256) void bucket_bounds_kernel(const u64 *__restrict__ O2, const u64 *__restrict__ moff,
                                                            const uint32_t *__restrict__ nch, const u64 *__restrict__ info,
                                                            PartGeom g, u64 *__restrict__ bstart) {
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r > g.nregions) return;
    if (r == g.nregions) {
        bstart[r] = info[2];
        return;
    }
    const uint32_t p1 = (uint32_t)(r >> g.p2_bits), p2 = (uint32_t)r & ((1u << g.p2_bits) - 1u);
    // an empty level-1 partition has no chunks: its buckets all start where the partition starts,
    // which is the O2 value at the next partition's first entry (or the grand total)
    bstart[r] = nch[p1] ? O2[moff[p1] + (u64)p2 * nch[p1]] : O2[moff[p1]];
}

// ---------------------------------------------------------------------------------------------
// region rebuild: one workgroup per table region, table image in LDS, no global atomics
// ---------------------------------------------------------------------------------------------
// FRESH: the table is known to be empty (skip the 128 KiB read).  A region that overflows is left
// untouched in HBM and flagged; the host re-inserts its bucket after growing the table.
template <bool FRESH>
__global__ __launch_bounds__(REGION_NT) void region_count_kernel(Slot *__restrict__ table, const u64 *__restrict__ keys,
                                                                 const u64 *__restrict__ bstart, uint8_t *__restrict__ rfail,
                                                                 uint32_t *__restrict__ rnew) {
    // LDS image is structure-of-arrays: 8-byte key probes then spread over all 64 banks instead of
    // the 16 bank pairs a 16-byte-slot layout would hit.
    __shared__ u64 s_key[REGION_SLOTS];  // 32 KiB
    __shared__ u64 s_cnt[REGION_SLOTS];  // 32 KiB
    __shared__ uint32_t s_fail;
    __shared__ uint32_t s_new;
    const int tid = threadIdx.x;
    const u64 r = blockIdx.x;
    const u64 lo = bstart[r], hi = bstart[r + 1];
    if (lo == hi) {  // nothing new for this region
        if (tid == 0) rnew[r] = 0;
        return;
    }
    Slot *reg = table + r * REGION_SLOTS;
    u64 kbuf[REGION_RK];  // first round of keys: in flight while the region image is loaded
#pragma unroll
    for (int j = 0; j < REGION_RK; ++j) {
        const u64 i = lo + (u64)j * REGION_NT + tid;
        kbuf[j] = i < hi ? keys[i] : KH_EMPTY_KEY;
    }
    const uint4 *g4 = reinterpret_cast<const uint4 *>(reg);
#pragma unroll
    for (uint32_t i = tid; i < REGION_SLOTS; i += REGION_NT) {
        if (FRESH) {
            s_key[i] = KH_EMPTY_KEY;
            s_cnt[i] = 0;
        } else {
            const uint4 v = g4[i];
            s_key[i] = ((u64)v.y << 32) | v.x;
            s_cnt[i] = ((u64)v.w << 32) | v.z;
        }
    }
    if (tid == 0) {
        s_fail = 0;
        s_new = 0;
    }
    __syncthreads();
    uint32_t nd = 0;
    for (u64 base = lo; base < hi; base += (u64)REGION_RK * REGION_NT) {
        u64 nbuf[REGION_RK];
#pragma unroll
        for (int j = 0; j < REGION_RK; ++j) {  // next round's keys in flight while this round is inserted
            const u64 i = base + (u64)(REGION_RK + j) * REGION_NT + tid;
            nbuf[j] = i < hi ? keys[i] : KH_EMPTY_KEY;
        }
#pragma unroll
        for (int j = 0; j < REGION_RK; ++j) {
            const u64 key = kbuf[j];
            if (key == KH_EMPTY_KEY) continue;
            uint32_t off = start_of_hash(kh_mix64(key));
            uint32_t probes = 0;
            for (; probes < REGION_SLOTS; ++probes) {
                u64 cur = s_key[off];
                if (cur == KH_EMPTY_KEY) {
                    cur = atomicCAS(&s_key[off], (u64)KH_EMPTY_KEY, key);  // ds_cmpst_rtn_b64
                    if (cur == KH_EMPTY_KEY) {
                        ++nd;
                        cur = key;
                    }
                }
                if (cur == key) {
                    atomicAdd(&s_cnt[off], 1ull);  // ds_add_u64
                    break;
                }
                off = (off + 1) & REGION_MASK;
            }
            if (probes == REGION_SLOTS) s_fail = 1;
        }
#pragma unroll
        for (int j = 0; j < REGION_RK; ++j) kbuf[j] = nbuf[j];
    }
    // No global atomics here: a million workgroups adding to one counter word serialise at the
    // memory side (measured: ~6 ns per same-address atomic, i.e. >100 ms per pass).  Per-region
    // results go to rnew[]/rfail[] and region_reduce_kernel folds them afterwards.
    const uint32_t dw = (uint32_t)wave_sum((u64)nd);
    if ((tid & 63) == 0 && dw) atomicAdd(&s_new, dw);  // LDS
    __syncthreads();
    if (s_fail) {
        if (tid == 0) {
            rfail[r] = 1;
            rnew[r] = 0;
        }
        return;
    }
    uint4 *o4 = reinterpret_cast<uint4 *>(reg);
#pragma unroll
    for (uint32_t i = tid; i < REGION_SLOTS; i += REGION_NT) {
        const u64 kk = s_key[i], cc = s_cnt[i];
        o4[i] = make_uint4((uint32_t)kk, (uint32_t)(kk >> 32), (uint32_t)cc, (uint32_t)(cc >> 32));
    }
    if (tid == 0) rnew[r] = s_new;
}

// Folds the per-region results of one region_count pass into the context counters.
__global__ __launch_bounds__(BLOCK) void region_reduce_kernel(const u64 *__restrict__ bstart, const uint8_t *__restrict__ rfail,
                                                              const uint32_t *__restrict__ rnew, u64 nregions, Counters *ctr) {
    const u64 stride = (u64)gridDim.x * BLOCK;
    u64 d = 0, km = 0, nf = 0;
    for (u64 r = (u64)blockIdx.x * BLOCK + threadIdx.x; r < nregions; r += stride) {
        if (rfail[r]) {
            ++nf;
        } else {
            d += rnew[r];
            km += bstart[r + 1] - bstart[r];
        }
    }
    d = wave_sum(d);
    km = wave_sum(km);
    nf = wave_sum(nf);
    if (lane_id() == 0) {
        if (d) atomicAdd(&ctr->distinct, d);
        if (km) atomicAdd(&ctr->kmers, km);
        if (nf) atomicAdd(&ctr->part_failed, nf);
    }
}

// Direct (atomic) insertion of the buckets whose region overflowed, after the table was grown.
// One workgroup per ORIGINAL region index; rfail/bstart refer to the geometry the buckets were
// built with, `table`/`nregions` to the grown table.
__global__ __launch_bounds__(BLOCK) void failed_buckets_insert_kernel(Slot *table, u64 nregions, const u64 *__restrict__ keys,
                                                                      const u64 *__restrict__ bstart,
                                                                      const uint8_t *__restrict__ rfail, Counters *ctr) {
    const u64 r = blockIdx.x;
    if (!rfail[r]) return;
    const u64 lo = bstart[r], hi = bstart[r + 1];
    uint32_t nd = 0, nf = 0;
    for (u64 i = lo + threadIdx.x; i < hi; i += BLOCK) upsert(table, nregions, keys[i], 1ull, nd, nf);
    const u64 d = wave_sum((u64)nd), f = wave_sum((u64)nf);
    if (lane_id() == 0) {  // rare path (only regions that overflowed): plain counter atomics are fine
        if (d) atomicAdd(&ctr->distinct, d);
        if (f) atomicAdd(&ctr->failed, f);
    }
    if (threadIdx.x == 0) atomicAdd(&ctr->kmers, hi - lo);
}

}  // namespace kh
