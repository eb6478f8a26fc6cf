// partition.hip.h -- the locality stage of the counting path (gfx950, wave64).
//
// The direct path (count_direct_kernel) issues one memory-side atomic per k-mer and saturates the
// chip's atomic request rate (~18.5 G/s measured, profiles/r01a).  This path issues NO global
// atomics per k-mer: canonical keys are radix-partitioned by the top bits of their table hash H in
// two levels (level 1: 2^p1_bits <= 1024 partitions straight from the bases; level 2: 2^p2_bits
// <= 1024 buckets inside each level-1 partition), so that every bucket holds exactly the keys of
// ONE table region (kernels.hip.h: 4096 slots = 64 KiB).  One workgroup then rebuilds each region
// in LDS with LDS atomics and writes it back with coalesced 16-byte stores.
//
// Two payload formats travel through the partition buffers (template parameter PT):
//   u64       the canonical key itself; partition digits are recomputed from its hash.  Any k.
//   uint32_t  bits [p1_bits, p1_bits+32) of H.  Because the table hash is a bijection on 2k bits
//             and level 1 has already consumed the top p1_bits, these 32 bits identify the key
//             whenever 2k - p1_bits <= 32 (k <= 21 with 1024 level-1 partitions).  Halves the
//             HBM traffic of every stage after the extraction.
//
// Level 1 is ONE pass: a workgroup counting-sorts the payloads of a tile in LDS and appends every
// partition's run to that partition's current chunk of a chunk pool (see below), so no counting pass
// is needed to know where data goes.  Level 2 is "count, scan, scatter" with deterministic offsets (no
// global cursor atomics): workgroup b owns a fixed slice of one level-1 partition's chunk list, the
// count pass writes its per-bucket histogram as a column of a [bucket][workgroup] matrix, an exclusive
// scan of the flattened matrix yields every (bucket, workgroup) output offset, and the scatter pass
// re-reads the same slice, counting-sorts each batch in LDS and writes per-bucket runs.
//
// Reference counterpart: none (the reference is src/run.rs:526-571 + DashMap); results are the
// same multiset of (key,count) as the direct path.
#pragma once
#include <type_traits>
#include "part_common.hip.h"

namespace kh {

// ---- chunk list: chunk ids ordered by partition ---------------------------------------------------
// pcount[p] += chunks owned by partition p among ids [0, nchunks)
// ptotal (optional): ptotal[p] += payloads in those chunks
KH_GLOBAL __launch_bounds__(1024) void chunk_hist_kernel(const uint16_t *__restrict__ chunk_part, const u64 *__restrict__ pool_next,
                                                          u64 pool_chunks, uint32_t *__restrict__ pcount,
                                                          const uint8_t *__restrict__ fill8, u64 *__restrict__ ptotal) {
    __shared__ uint32_t s_h[MAX_P1];
    __shared__ uint32_t s_f[MAX_P1];
    const int tid = threadIdx.x;
    s_h[tid] = 0;
    s_f[tid] = 0;
    __syncthreads();
    u64 n = *pool_next;
    if (n > pool_chunks) n = pool_chunks;
    const u64 stride = (u64)gridDim.x * 1024;
    for (u64 i = (u64)blockIdx.x * 1024 + tid; i < n; i += stride) {  // (<= 2^32 / 256 chunks per workgroup here: the sums fit 32 bits)
        const uint16_t p = chunk_part[i];
        if (p != PART_NONE) {
            atomicAdd(&s_h[p], 1u);
            if (ptotal) atomicAdd(&s_f[p], (uint32_t)fill8[i] + 1u);
        }
    }
    __syncthreads();
    if (s_h[tid]) atomicAdd(&pcount[tid], s_h[tid]);
    if (ptotal && s_f[tid]) atomicAdd(&ptotal[tid], (u64)s_f[tid]);
}

// plist[pstart[p] + rank] = chunk id.  cursors[] starts as a copy of pstart[] (low 32 bits suffice:
// a pool holds far fewer than 2^32 chunks).  One workgroup per 16384 chunk ids.
KH_GLOBAL __launch_bounds__(1024) void chunk_list_kernel(const uint16_t *__restrict__ chunk_part, const u64 *__restrict__ pool_next,
                                                          u64 pool_chunks, uint32_t *__restrict__ cursors,
                                                          uint32_t *__restrict__ plist) {
    __shared__ uint32_t s_h[MAX_P1];
    __shared__ uint32_t s_base[MAX_P1];
    const int tid = threadIdx.x;
    u64 n = *pool_next;
    if (n > pool_chunks) n = pool_chunks;
    const u64 lo = (u64)blockIdx.x * 16384;
    if (lo >= n) return;
    s_h[tid] = 0;
    __syncthreads();
    uint32_t tag[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const u64 i = lo + (u64)j * 1024 + tid;
        tag[j] = 0xFFFFFFFFu;
        if (i < n) {
            const uint16_t p = chunk_part[i];
            if (p != PART_NONE) tag[j] = ((uint32_t)p << 16) | atomicAdd(&s_h[p], 1u);
        }
    }
    __syncthreads();
    s_base[tid] = s_h[tid] ? atomicAdd(&cursors[tid], s_h[tid]) : 0u;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (tag[j] != 0xFFFFFFFFu) plist[s_base[tag[j] >> 16] + (tag[j] & 0xFFFFu)] = (uint32_t)(lo + (u64)j * 1024 + tid);
}

// ---- how many DISTINCT keys does the batch hold?  (round 4) --------------------------------------------------------------
// Level 1 splits the batch by the top 10 bits of a bijective hash: a partition holds every occurrence of an exact 1/1024
// of the key space, whatever the table will look like.  So the distinct payloads of a few partitions, counted exactly, times
// 1024 / (partitions sampled) is the batch's distinct k-mer count to ~0.1 % -- known BEFORE the table's geometry is needed
// (level 2 is the first kernel that depends on it).  The host sizes the table from it (kmerhip.hip, partition_batch): no
// capacity hint, no worst-case table, and the load the region pass sees is the one it was tuned for.
// An open-addressing SET in global memory (u64 slots, ~0 = free; value = payload, with the partition above a 32-bit one):
// one plain load settles the ~11 of 12 payloads that are copies, a compare-and-swap the rest.  sub_bits: of the sampled
// partitions' keys only those whose next sub_bits hash bits are zero are counted -- still every occurrence of an exact share
// of the key space, so that a large batch's sample stays ~1 M payloads and its set in the L2.  out[0] += new values,
// out[1] += payloads counted, out[2] += values that found no room within 128 probes (the estimate is then void).
template <typename PT>
__global__ __launch_bounds__(BLOCK) void distinct_sample_kernel(ChunkSrc cs, const u64 *__restrict__ pstart, uint32_t p_first, uint32_t np,
                                                                uint32_t sub_bits, u64 *__restrict__ set, u64 set_mask, u64 *__restrict__ out) {
    const u64 c0 = pstart[p_first], c1 = pstart[p_first + np];
    uint32_t nd = 0, nf = 0;
    u64 seen = 0;
    uint32_t p = p_first;
    for (u64 ci = c0 + blockIdx.x; ci < c1; ci += gridDim.x) {  // one chunk (<= 256 payloads) per iteration
        while (pstart[p + 1] <= ci) ++p;                         // (uniform: the chunk's partition)
        const uint32_t chunk = cs.plist[ci];
        const uint32_t fill = (uint32_t)cs.fill8[chunk] + 1u;
        if (threadIdx.x >= fill) continue;
        const PT pay = reinterpret_cast<const PT *>(cs.pay)[(u64)chunk * CHUNK_PAY + threadIdx.x];
        if (sizeof(PT) == 8 && (u64)pay == KH_EMPTY_KEY) continue;
        if (sub_bits) {  // (a 32-bit payload IS hash bits; for an 8-byte one any fixed function of it will do)
            const uint32_t sel = sizeof(PT) == 4 ? (uint32_t)pay >> (32 - sub_bits) : (uint32_t)(kh_mix64((u64)pay) >> 40) >> (24 - sub_bits);
            if (sel) continue;
        }
        ++seen;
        // (a payload names a key inside its partition: the sampled partition's number rides in the 8-byte payload's low bits, which are
        //  zero -- ten of them: the sample is taken with 1024 level-1 partitions -- and beside the 4-byte one)
        const u64 v = sizeof(PT) == 4 ? ((u64)(p - p_first) << 32) | (u64)pay : ((u64)pay | (u64)(p - p_first));
        u64 h = kh_mix64(v) & set_mask;
        uint32_t probes = 0;
        for (; probes < 128; ++probes, h = (h + 1) & set_mask) {
            u64 cur = set[h];
            if (cur == ~0ull) {
                cur = atomicCAS((unsigned long long *)&set[h], ~0ull, (unsigned long long)v);
                if (cur == ~0ull) {
                    ++nd;
                    break;
                }
            }
            if (cur == v) break;
        }
        nf += probes == 128;
    }
    const u64 d = wave_sum((u64)nd), f = wave_sum((u64)nf);
    seen = wave_sum(seen);
    if (lane_id() == 0) {
        if (d) atomicAdd(&out[0], d);
        if (seen) atomicAdd(&out[1], seen);
        if (f) atomicAdd(&out[2], f);
    }
}

// Level-2 plan over chunk lists.  pstart = exclusive scan of pcount (P1 + 1 entries).  Same outputs as
// part2_plan_kernel; info[2] is left to the level-2 scan (the grand total is not known yet).
// only (optional): plan blocks for the partitions with only[p] != 0 alone -- the heavy partitions of a batch whose other
// partitions went through the arena kernel; the others then have no blocks (nch = 0).  cursors is left alone then.
// exclusive prefix sum over the 1024 lanes of a workgroup (s_scan: 1024 words of LDS); returns the lane's prefix, *total = the sum
// (the plan kernels' serial loops over 1024 partitions -- one lane, a dependent global load per step -- were 0.2 ms each)
__device__ __forceinline__ u64 block_scan_1024(u64 v, u64 *s_scan, int tid, u64 *total) {
    s_scan[tid] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const u64 add = tid >= o ? s_scan[tid - o] : 0ull;
        __syncthreads();
        s_scan[tid] += add;
        __syncthreads();
    }
    const u64 incl = s_scan[tid];
    if (total) *total = s_scan[1023];
    __syncthreads();
    return incl - v;
}

KH_GLOBAL __launch_bounds__(1024) void part2_plan_chunked_kernel(const u64 *__restrict__ pstart, PartGeom g,
                                                                  Part2Block *__restrict__ blocks, u64 max_blocks,
                                                                  u64 *__restrict__ moff, uint32_t *__restrict__ nch,
                                                                  u64 *__restrict__ info, uint32_t *__restrict__ cursors,
                                                                  uint32_t force_wide, const uint8_t *__restrict__ only) {
    __shared__ u64 s_bbase[MAX_P1 + 1];
    __shared__ u64 s_scan[1024];
    const int tid = threadIdx.x;
    const int P1 = 1 << g.p1_bits;
    if (tid < P1 && !only) cursors[tid] = (uint32_t)pstart[tid];
    auto nchunks_of = [&](int p) -> u64 { return (only && !only[p]) ? 0ull : pstart[p + 1] - pstart[p]; };
    const u64 my_chunks = tid < P1 ? nchunks_of(tid) : 0ull;
    const u64 my_blocks = (my_chunks + CPB - 1) / CPB;
    u64 b = 0;
    const u64 my_base = block_scan_1024(my_blocks, s_scan, tid, &b);
    if (tid < P1) s_bbase[tid] = my_base;
    // info[3]: some partition's level-2 output (payloads + unit padding) does not fit 32-bit offsets -- only
    // when > 4 G k-mers of one batch share a level-1 digit.  The unit-writing level-2 kernel then stands
    // down for the whole batch and the unaligned one runs (both are launched, each checks this word).
    // (KMERHIP_P2_FORCE_WIDE=1: tests exercise the stand-down without 4 G k-mers)
    const int wide = __syncthreads_or((force_wide || my_chunks * CHUNK_PAY + my_blocks * 1024ull * 64ull >= (1ull << 32)) ? 1 : 0);
    if (tid == 0) {
        s_bbase[P1] = b;
        info[0] = b;
        info[1] = b * g.b2;
        info[2] = 0;
        info[3] = wide ? 1 : 0;
    }
    __syncthreads();
    if (tid < P1) {
        const u64 lo = pstart[tid], hi = lo + nchunks_of(tid);
        const u64 b0 = s_bbase[tid];
        const uint32_t n = (uint32_t)(s_bbase[tid + 1] - b0);
        moff[tid] = b0 * g.b2;
        nch[tid] = n;
        for (uint32_t c = 0; c < n; ++c) {
            if (b0 + c >= max_blocks) break;
            Part2Block pb;
            pb.lo = lo + (u64)c * CPB;
            pb.hi = pb.lo + CPB < hi ? pb.lo + CPB : hi;
            pb.mbase = b0 * g.b2 + c;
            pb.mstride = n;
            pb.p1 = (uint32_t)tid;
            blocks[b0 + c] = pb;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// level 2 work list: one workgroup per PART2_CHUNK keys of a level-1 partition
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// generic exclusive scan u32 -> u64 (three small kernels)
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_NT = 256;
constexpr int SCAN_PER = 16;
constexpr int SCAN_CHUNK = SCAN_NT * SCAN_PER;  // 4096 entries per workgroup

KH_GLOBAL __launch_bounds__(SCAN_NT) void scan_partials_kernel(const uint32_t *__restrict__ in, u64 n, u64 *__restrict__ partial) {
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
KH_GLOBAL __launch_bounds__(1024) void scan_spine_kernel(u64 *__restrict__ partial, u64 nb) {
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

KH_GLOBAL __launch_bounds__(SCAN_NT) void scan_apply_kernel(const uint32_t *__restrict__ in, u64 n,
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
// level 2, pass A: histogram of bucket ids (p2) per workgroup.  H2 must be zero-filled.
// ---------------------------------------------------------------------------------------------
// pad: every (bucket, workgroup) count is rounded up to a multiple of `pad` payloads (1 = exact), so that the
// exclusive scan puts every segment of the level-2 output on a line boundary (part2_scatter_lines_kernel).
template <typename PT, bool CHUNKED>
__global__ __launch_bounds__(PART_NT) void part2_count_kernel(const PT *__restrict__ pays, ChunkSrc cs,
                                                              const Part2Block *__restrict__ blocks,
                                                              const u64 *__restrict__ info, PartGeom g,
                                                              uint32_t *__restrict__ H2, uint32_t pad) {
    __shared__ uint32_t s_hist[1u << MAX_P2_BITS];
    __shared__ uint32_t s_chk[CHUNKED ? CPB : 1];
    __shared__ uint16_t s_cfill[CHUNKED ? CPB : 1];
    if ((u64)blockIdx.x >= info[0]) return;
    const Part2Block pb = blocks[blockIdx.x];
    const int tid = threadIdx.x;
    s_hist[tid] = 0;  // PART_NT == 1 << MAX_P2_BITS
    p2_stage_chunks<CHUNKED>(cs, pb, s_chk, s_cfill, tid, PART_NT);
    __syncthreads();
    // eight independent loads in flight per lane (a one-load-per-iteration loop is latency bound)
    const uint32_t n = p2_count_of<CHUNKED>(pb);
    for (uint32_t base = 0; base < n; base += 8 * PART_NT) {
        PT v[8];
        uint32_t ok = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            ok |= (uint32_t)p2_load<CHUNKED, PT>(pays, cs, pb, s_chk, s_cfill, base + (uint32_t)j * PART_NT + tid, n, v[j]) << j;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (ok & (1u << j)) atomicAdd(&s_hist[Pay<PT>::p2(l2_finish(v[j], pb.p1, g), g)], 1u);
    }
    __syncthreads();
    if (pad > 1 && info[3]) pad = 1;  // (the unit-writing kernel stands down for this batch: no padding)
    if ((uint32_t)tid < g.b2) H2[pb.mbase + (u64)tid * pb.mstride] = (s_hist[tid] + pad - 1) / pad * pad;
}

// ---------------------------------------------------------------------------------------------
// level 2, pass B: scatter into buckets (one bucket == one table region).
// 512 lanes x 32 payloads = 16384 per batch.  The batch size sets the length of the per-bucket runs
// (batch / 512 buckets = 32 payloads = one 128-byte line): measured on S100M, 8192 -> 44.1 ms (64-byte
// runs, writes 1.48x the algorithmic bytes), 16384 -> 36.6 ms, 24576 / 32768 -> 38 ms (one workgroup
// per CU, nothing left to overlap with).  1024-lane workgroups are no faster at any batch size; 512
// lanes also lift the 128-VGPR ceiling of 1024-lane workgroups.
// ---------------------------------------------------------------------------------------------
#ifndef KH_PART2_NT
#define KH_PART2_NT 512
#endif
#ifndef KH_PART2_TILE
#define KH_PART2_TILE 16384
#endif
constexpr int PART2_NT = KH_PART2_NT;
constexpr int PART2_TILE = KH_PART2_TILE;
constexpr int P2_PER = PART2_TILE / PART2_NT;   // payloads per lane per batch

// NBK = size of the per-bucket LDS arrays: 1024 (any p2_bits <= 10), or 512 when the geometry has at most 512
// buckets per partition (the headline table: p2_bits = 9).  With 512 the workgroup's LDS drops from 86 to 79 KB,
// so that TWO workgroups share a CU (4 waves per SIMD instead of 2: the scattered line-sized writes and the LDS
// round trips of one overlap the other's); registers are capped at 128 for that.
template <typename PT, bool CHUNKED, int NBK>
__global__ __launch_bounds__(PART2_NT, (NBK == 512 && sizeof(PT) == 4) ? 4 : 2) void part2_scatter_kernel(
    const PT *__restrict__ pays, ChunkSrc cs, const Part2Block *__restrict__ blocks, const u64 *__restrict__ info, PartGeom g,
    const u64 *__restrict__ O2, PT *__restrict__ out, uint32_t only_if_wide) {
    constexpr int OWN = NBK / PART2_NT;         // buckets whose output cursor a lane keeps in registers
    __shared__ PT s_stage[PART2_TILE + 1];      // 64 KiB (u32) / 128 KiB (u64), + a trash slot
    __shared__ uint32_t s_cnt[NBK];
    __shared__ uint16_t s_lofs[NBK];            // batch-local run starts (< PART2_TILE <= 32768)
    __shared__ u64 s_dst[NBK];                  // global position of run p minus its batch-local start
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_chk[CHUNKED ? CPB : 1];
    __shared__ uint16_t s_cfill[CHUNKED ? CPB : 1];
    if ((u64)blockIdx.x >= info[0]) return;
    if (only_if_wide && !info[3]) return;  // launched behind part2_scatter_lines_kernel: runs only where that one stood down
    const Part2Block pb = blocks[blockIdx.x];
    const int tid = threadIdx.x;
    const int P2 = (int)g.b2;
    p2_stage_chunks<CHUNKED>(cs, pb, s_chk, s_cfill, tid, PART2_NT);
    // lane tid owns buckets tid + q * PART2_NT: their running output cursors live in registers
    u64 gcur[OWN];
#pragma unroll
    for (int q = 0; q < OWN; ++q) {
        const int b = tid + q * PART2_NT;
        s_cnt[b] = 0;
        gcur[q] = b < P2 ? O2[pb.mbase + (u64)b * pb.mstride] : 0;
    }
    __syncthreads();
    // Branch-free loads: indices are block-relative 32-bit, clamped to the last valid payload (the
    // block is never empty), validity is a bit mask.  (Conditional loads made the compiler carry
    // sixteen 64-bit addresses and their phi copies through the loop: 219 VGPRs.)
    const uint32_t n = p2_count_of<CHUNKED>(pb);
    PT pay[P2_PER];
    uint32_t have = 0;  // bit j: pay[j] holds a payload
#pragma unroll
    for (int j = 0; j < P2_PER; ++j)  // lane-contiguous: coalesced loads
        have |= (uint32_t)p2_load<CHUNKED, PT>(pays, cs, pb, s_chk, s_cfill, (uint32_t)j * PART2_NT + tid, n, pay[j]) << j;
    for (uint32_t base = 0; base < n; base += PART2_TILE) {
        uint32_t tag[P2_PER];
#pragma unroll
        for (int j = 0; j < P2_PER; ++j) pay[j] = l2_finish(pay[j], pb.p1, g);
#pragma unroll
        for (int j = 0; j < P2_PER; ++j) tag[j] = (have & (1u << j)) ? (Pay<PT>::p2(pay[j], g) << 16) : 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < P2_PER; ++j)  // all LDS rank atomics in flight before the first is consumed
            if (tag[j] != 0xFFFFFFFFu) tag[j] |= atomicAdd(&s_cnt[tag[j] >> 16], 1u);
        __syncthreads();
        block_exclusive_scan_n<NBK>(s_cnt, s_lofs, s_wsum, tid);
        {  // branch-free staging (see part1_scatter_chunked_kernel): all run starts first, then the stores
            uint32_t rs[P2_PER];
#pragma unroll
            for (int j = 0; j < P2_PER; ++j) rs[j] = s_lofs[(tag[j] >> 16) & (NBK - 1)];
#pragma unroll
            for (int j = 0; j < P2_PER; ++j)
                s_stage[tag[j] != 0xFFFFFFFFu ? rs[j] + (tag[j] & 0xFFFFu) : (uint32_t)PART2_TILE] = pay[j];
        }
#pragma unroll
        for (int q = 0; q < OWN; ++q) {  // publish run destinations, advance the cursors
            const int b = tid + q * PART2_NT;
            s_dst[b] = gcur[q] - s_lofs[b];
            gcur[q] += s_cnt[b];
        }
        const uint32_t total = (uint32_t)s_lofs[NBK - 1] + s_cnt[NBK - 1];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < OWN; ++q) s_cnt[tid + q * PART2_NT] = 0;
        // next batch's payloads are fetched while this batch's runs are written out
        have = 0;
#pragma unroll
        for (int j = 0; j < P2_PER; ++j)
            have |= (uint32_t)p2_load<CHUNKED, PT>(pays, cs, pb, s_chk, s_cfill, base + PART2_TILE + (uint32_t)j * PART2_NT + tid, n, pay[j]) << j;
#if KH_ABL2 & 28  /* timing experiment: the same scatter pattern and byte volume, but every write one whole aligned 128 / 64 / 32-byte unit */
        for (uint32_t i = tid; i < (uint32_t)PART2_TILE; i += PART2_NT) {
            constexpr uint32_t LP = ((KH_ABL2 & 4) ? 128 : (KH_ABL2 & 8) ? 64 : 32) / sizeof(PT);
            const uint32_t b = (i / LP) % (uint32_t)P2;
            out[((s_dst[b] + s_lofs[b]) & ~(u64)(LP - 1)) + (i % LP)] = s_stage[i];
        }
#else
#pragma unroll 2
        for (uint32_t i = tid; i < total; i += PART2_NT) {
            const PT v = s_stage[i];
#if KH_ABL2 & 1   /* timing experiment: LDS side only, no global stores */
            if (v == (PT)0x12345678u && s_dst[Pay<PT>::p2(v, g)] == 5) out[0] = v;
#elif KH_ABL2 & 2 /* timing experiment: the same bytes written densely (batch after batch) instead of scattered */
            out[(u64)blockIdx.x * (CPB * CHUNK_PAY) + base + i + (s_dst[Pay<PT>::p2(v, g)] & 0)] = v;
#else
            out[s_dst[Pay<PT>::p2(v, g)] + i] = v;  // consecutive lanes -> consecutive addresses inside a run
#endif
        }
#endif
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// level 2, pass B writing whole aligned 64-byte units only (32-bit payloads, <= 512 buckets per partition)
// ---------------------------------------------------------------------------------------------
// Timing experiments on the kernel above (KH_ABL2, S100M: 36.4 ms): loads + LDS sort 18.2 ms; the same bytes
// stored densely +2.7; stored to the same scattered places but as whole aligned 128-byte lines +3.8, as whole
// aligned 64-byte units +5.2, 32-byte units +12; stored as today -- one 128-byte run per bucket and batch,
// starting wherever the previous run ended, i.e. partly written sectors at both ends -- +18.  The memory
// system has to merge or read-modify-write every partly written sector; whole ones it just takes.  So here
// nothing but whole aligned 64-byte units (UNIT = 16 payloads) ever leaves the workgroup:
//   * every (bucket, workgroup) segment of the output starts on a unit boundary: the count pass pads the
//     counts to multiples of UNIT, and the pad is filled with SENTINELS at the end of the segment (payloads
//     whose level-2 digit is not the bucket's: the region pass skips them);
//   * per bucket the <= UNIT - 1 payloads that do not fill a unit stay in LDS until the bucket's next payloads
//     complete it.  They never get copied: the counting sort knows, before it stores anything, how many of a
//     bucket's new payloads complete units (those go to the sorted stage) and which are the new tail (those go
//     straight to the bucket's slot of the residue array of the NEXT batch -- two residue arrays, by parity);
//   * the units that became complete are enumerated (s_unit: output offset, where in the stage the unit
//     starts, how many of its first payloads come from the carried residue) and written by 16 lanes each.
// 1024 lanes x 16 payloads per batch: with the stores cheap the kernel is bound by its own instruction stream
// (one workgroup per CU: 152 KB of LDS), and sixteen waves hide the LDS round trips better than eight.
// 64-bit payloads (k >= 22): the same kernel with 8 payloads per unit and batches of 8192 (same LDS bytes); the
// sentinel there is KH_EMPTY_KEY, which the 64-bit region pass skips anyway.
constexpr int P2L_NBK = 512;   // buckets per partition this kernel handles
constexpr int P2L_NT = 1024;   // lanes per workgroup
template <typename PT>
struct P2L {
    static constexpr int UNIT = 64 / (int)sizeof(PT);                                  // payloads per unit (64 bytes)
    static constexpr int TILE = sizeof(PT) == 4 ? PART2_TILE : PART2_TILE / 2;         // payloads per batch
    static constexpr int PER = TILE / P2L_NT;
};

// sentinel of bucket `digit`: 32-bit payloads -- a payload of ANOTHER bucket (needs b2 >= 2);
// 64-bit payloads (keys) -- the empty key
template <typename PT>
__device__ __forceinline__ PT p2_sentinel(uint32_t digit, uint32_t b2) {
    if (sizeof(PT) == 8) return (PT)KH_EMPTY_KEY;
    return (PT)kh_xlo(digit ? digit - 1u : 1u, b2);  // the first payload of a neighbouring bucket
}

template <typename PT, bool CHUNKED>
__global__ __launch_bounds__(P2L_NT) void part2_scatter_lines_kernel(const PT *__restrict__ pays, ChunkSrc cs,
                                                                     const Part2Block *__restrict__ blocks,
                                                                     const u64 *__restrict__ info, PartGeom g,
                                                                     const u64 *__restrict__ O2, PT *__restrict__ out) {
    constexpr int NBK = P2L_NBK, UNIT = P2L<PT>::UNIT, NT = P2L_NT, PER = P2L<PT>::PER, TILE = P2L<PT>::TILE;
    constexpr int MAXU = TILE / UNIT + NBK;             // units one batch can complete
    constexpr uint32_t RES0 = TILE + 1;                 // s_buf: [sorted stage | trash | residues (even) | residues (odd)]
    constexpr uint32_t RES_SZ = NBK * UNIT;
    __shared__ PT s_buf[TILE + 1 + 2 * NBK * UNIT];
    __shared__ uint32_t s_cnt[NBK];
    __shared__ uint2 s_ofs[NBK];        // x: stage start of the bucket's run | payloads that go to the stage << 16
                                        // y: s_buf index of the new tail's payload of rank 0 (biased by those payloads)
    __shared__ uint2 s_unit[MAXU];      // x: output offset of the unit (payloads, from the block's base position)
                                        // y: (stage index of unit position 0) + UNIT | carried payloads at its front << 16 | bucket << 20
    __shared__ uint32_t s_wsum[NBK / 64];
    __shared__ uint32_t s_nu;
    __shared__ uint32_t s_chk[CHUNKED ? CPB : 1];
    __shared__ uint16_t s_cfill[CHUNKED ? CPB : 1];
    if ((u64)blockIdx.x >= info[0] || info[3]) return;  // info[3]: a partition too large for 32-bit offsets (see the plan kernel)
    const Part2Block pb = blocks[blockIdx.x];
    const int tid = threadIdx.x;
    const int P2 = (int)g.b2;
    p2_stage_chunks<CHUNKED>(cs, pb, s_chk, s_cfill, tid, NT);
    // lane b < 512 owns bucket b: its output cursor (always on a unit boundary, relative to the block's base
    // position = bucket 0's segment: the segments of the higher buckets lie above it) and its carried count
    const u64 gbase = O2[pb.mbase];
    uint32_t gdone = (tid < P2) ? (uint32_t)(O2[pb.mbase + (u64)tid * pb.mstride] - gbase) : 0u;
    uint32_t res = 0;
    uint32_t par = 0;  // residues of this batch are read from array `par`, new tails go to array `par ^ 1`
    if (tid < NBK) s_cnt[tid] = 0;
    __syncthreads();
    const uint32_t n = p2_count_of<CHUNKED>(pb);
    // Element e = first + j * NT + tid of the block's input lies in chunk (e >> 8) of its chunk list, and a wave's
    // 64 lanes always share that chunk (NT and the wave's first lane are multiples of 64, chunks hold 256): chunk
    // id and fill level are wave-uniform -- read once, kept in scalar registers, the address is base + lane offset.
    const uint32_t nchk = CHUNKED ? (uint32_t)(pb.hi - pb.lo) : 0u;
    const uint32_t woff = (uint32_t)tid & (CHUNK_PAY - 1);  // offset inside the chunk
    auto load_batch = [&](uint32_t first, PT (&pay)[PER], uint32_t &have) {  // have: bit j = pay[j] holds a payload
        have = 0;
        // lane (l mod PER) of the wave fetches the metadata of the wave's j-th chunk: ONE LDS read per wave and array
        // instead of PER; v_readlane hands every lane the j-th pair
        uint32_t mychunk = 0, myfill = 0;
        if (CHUNKED) {
            const uint32_t myci = (first >> 8) + ((uint32_t)tid & (PER - 1)) * (NT / CHUNK_PAY) + ((uint32_t)tid >> 8);
            const uint32_t mycc = myci < nchk ? myci : nchk - 1;  // (clamped: the loads below are unconditional)
            mychunk = s_chk[mycc];
            myfill = myci < nchk ? (uint32_t)s_cfill[mycc] : 0u;  // (no such chunk: no payload is valid)
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (CHUNKED) {
                const uint32_t chunk = __builtin_amdgcn_readlane(mychunk, j);
                const uint32_t fillc = __builtin_amdgcn_readlane(myfill, j);
                const bool ok = woff < fillc;
                pay[j] = reinterpret_cast<const PT *>(cs.pay)[(u64)chunk * CHUNK_PAY + (ok ? woff : 0u)];
                have |= (uint32_t)ok << j;
            } else {
                have |= (uint32_t)p2_load<CHUNKED, PT>(pays, cs, pb, s_chk, s_cfill, first + (uint32_t)j * NT + tid, n, pay[j]) << j;
            }
        }
    };
    // One batch.  `pay` holds its payloads; the NEXT batch's are requested into `nxt` after the first barrier and
    // awaited right BEFORE this batch's units are stored: vmcnt counts loads and stores alike, so a wait for loaded
    // payloads placed after the stores (at the top of the next batch, as it used to be) is a wait for the
    // acknowledgement of every store just issued -- a memory round trip per batch.  Hence two register sets, used
    // alternately.
    auto batch = [&](uint32_t base, PT (&pay)[PER], uint32_t have, PT (&nxt)[PER], uint32_t &have_nxt) {
        const uint32_t res_old = RES0 + par * RES_SZ, res_new = RES0 + (par ^ 1u) * RES_SZ;
        uint32_t tag[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) pay[j] = l2_finish(pay[j], pb.p1, g);
#pragma unroll
        for (int j = 0; j < PER; ++j) tag[j] = (have & (1u << j)) ? (Pay<PT>::p2(pay[j], g) << 16) : 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < PER; ++j)  // all LDS rank atomics in flight before the first is consumed
            if (tag[j] != 0xFFFFFFFFu) tag[j] |= atomicAdd(&s_cnt[tag[j] >> 16], 1u);
        __syncthreads();
        load_batch(base + TILE, nxt, have_nxt);
        // owner lanes: how the bucket's new payloads split into completed units and the new tail
        uint32_t c = 0, nu = 0, thr = 0, packed = 0, incl = 0;
        if (tid < NBK) {
            c = s_cnt[tid];
            nu = (res + c) / UNIT;                  // units completing in this batch
            thr = nu ? nu * UNIT - res : 0u;        // new payloads that go to the stage (<= c)
            packed = thr | (nu << 16);              // one scan for both: sums <= 16384 payloads, <= 1536 units
            incl = packed;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t nb = __shfl_up(incl, off, 64);
                if ((tid & 63) >= off) incl += nb;
            }
            if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
        }
        __syncthreads();
        if (tid < NBK) {
            uint32_t excl = incl - packed;
            for (int q = 0; q < (tid >> 6); ++q) excl += s_wsum[q];
            const uint32_t lofs = excl & 0xFFFFu, uofs = excl >> 16;
            const uint32_t dst0 = nu ? 0u : res;    // where the new tail starts in the bucket's residue slot
            s_ofs[tid] = make_uint2(lofs | (thr << 16), res_new + (uint32_t)tid * UNIT + dst0 - thr);
            for (uint32_t l = 0; l < nu; ++l)
                s_unit[uofs + l] = make_uint2(gdone + l * UNIT,
                                              (lofs + l * UNIT - res + UNIT) | ((l == 0 ? res : 0u) << 16) | ((uint32_t)tid << 20));
            if (!nu && res)  // no unit completes: the carried payloads move on to the next batch's array
                for (uint32_t i = 0; i < res; ++i) s_buf[res_new + tid * UNIT + i] = s_buf[res_old + tid * UNIT + i];
            if (tid == NBK - 1) s_nu = uofs + nu;
            gdone += nu * UNIT;
            res = (res + c) % UNIT;
        }
        __syncthreads();
        {  // branch-free staging: all offsets first, then the stores (stage for unit payloads, residue array for tails)
            uint2 o[PER];
#pragma unroll
            for (int j = 0; j < PER; ++j) o[j] = s_ofs[(tag[j] >> 16) & (NBK - 1)];
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t rank = tag[j] & 0xFFFFu;
                uint32_t addr = rank < (o[j].x >> 16) ? (o[j].x & 0xFFFFu) + rank : o[j].y + rank;
                if (tag[j] == 0xFFFFFFFFu) addr = (uint32_t)TILE;  // trash slot
                s_buf[addr] = pay[j];
            }
        }
        __syncthreads();
        if (tid < NBK) s_cnt[tid] = 0;
        // (the wait for the next batch's payloads goes HERE, before the stores)
#pragma unroll
        for (int j = 0; j < PER; ++j) asm volatile("" : "+v"(nxt[j]));
        // the completed units: 16 lanes per unit, carried payloads first, then the sorted run
#if KH_ABL3 & 2  /* timing experiment: no write-out */
        const uint32_t nslots = 0;
#else
        const uint32_t nslots = s_nu * UNIT;
#endif
        PT *__restrict__ obase = out + gbase;
#pragma unroll 4
        for (uint32_t x = tid; x < nslots; x += NT) {
            const uint2 u = s_unit[x / UNIT];
            const uint32_t i = x % UNIT;
            const uint32_t nres = (u.y >> 16) & 0xFu, b = u.y >> 20;
            const uint32_t idx = i < nres ? res_old + b * UNIT + i : (u.y & 0xFFFFu) - UNIT + i;
#if KH_ABL3 & 1  /* timing experiment: LDS side of the write-out only */
            if (s_buf[idx] == (PT)0x12345678u && u.x == 77u) out[0] = 1;
#else
            obase[u.x + i] = s_buf[idx];
#endif
        }
        par ^= 1u;
        // (the next batch's barriers order everything: its owner lanes rewrite s_ofs / s_unit after two of them,
        //  its tails go to the array this batch has just finished reading)
    };
    PT payA[PER], payB[PER];
    uint32_t haveA = 0, haveB = 0;
    load_batch(0, payA, haveA);
#pragma unroll
    for (int j = 0; j < PER; ++j) asm volatile("" : "+v"(payA[j]));  // (waited for here, so that the loop's top does not wait: it would wait for stores)
    for (uint32_t base = 0; base < n; base += 2 * TILE) {
        batch(base, payA, haveA, payB, haveB);
        if (base + TILE < n) batch(base + TILE, payB, haveB, payA, haveA);
    }
    // the last, incomplete unit of every bucket: padded with sentinels (the count pass reserved the room)
    __syncthreads();
    if (tid < NBK) s_ofs[tid] = make_uint2(res, gdone);
    __syncthreads();
    const uint32_t res_fin = RES0 + par * RES_SZ;
    for (uint32_t x = tid; x < (uint32_t)NBK * UNIT; x += NT) {
        const uint32_t b = x / UNIT, i = x % UNIT;
        const uint2 m = s_ofs[b];
        if (m.x && b < (uint32_t)P2) out[gbase + m.y + i] = i < m.x ? s_buf[res_fin + b * UNIT + i] : p2_sentinel<PT>(b, g.b2);
    }
}

// ---------------------------------------------------------------------------------------------
// level 2 WITHOUT a counting pass: per-bucket arenas sized from the level-1 partition totals
// ---------------------------------------------------------------------------------------------
// The count pass exists to give every (bucket, workgroup) segment of the level-2 output its exact offset; it reads all
// 51 GB of level-1 output for that (9.6 of the step's 89 ms on S100M).  Level 1 already knows how many payloads each
// of its partitions holds (chunk fill levels), and a partition's buckets are equally likely: bucket (p, b) gets an ARENA
// of n_p / P2 payloads plus a quarter plus 1024 (the counts of coverage-deep data spread ~3.6 x wider than a
// multinomial's, sigma ~ 570 around 24.8 K on S100M), ONE workgroup handles a whole partition -- so a bucket's write
// position lives in its owner lanes' registers, no atomics -- and whatever does not fit its arena (a heavy hitter's
// copies, essentially) goes to an overflow list of (region, payload) pairs that is inserted through the direct path
// after the region pass.  The region pass reads [bstart[r], bend[r]); the gaps are address space, not traffic.
// The workgroup sorts through per-bucket bins in LDS as level 1 does (128 KiB shared out among the partition's P2 <= 512
// buckets: 256 bytes each at 512; rank atomic -> bin, flush of whole units after half a batch); a payload whose rank
// does not fit its bin goes to the overflow list too.
// If the list itself would overflow (ovf[1] set), the host runs the exact count -> scan -> scatter path for the batch.

// bstart[r] for r = (p, b): arenas of cap_p = align32(ceil(n_p / P2) * 5 / 4 + 1024) payloads, partition after partition
// ovf[1] = 2 if some partition holds more than skew_x times the mean (one workgroup handles a whole partition: a
// partition that heavy -- a homopolymer's, say -- would be the whole pass; the exact path splits partitions into blocks)
// Round 3: a HEAVY partition (more than skew_x times the mean: a homopolymer's, a satellite's) no longer sends the whole
// batch to the exact path.  It gets no arenas and no workgroup here (heavy[p] = 1, capacity 0); the host then runs the exact
// count -> scan -> scatter kernels over the heavy partitions' chunk lists alone (they split a partition into blocks of 1024
// chunks, any number of workgroups) and their buckets follow the arenas in the same buffer.  ovf[2] = heavy partitions,
// ovf[3] = payloads in them; ovf[1] = 2 only if those exceed heavy_room payloads (the room the host has reserved behind
// the arenas): then the batch does take the exact path as a whole.
KH_GLOBAL __launch_bounds__(1024) void arena_plan_kernel(const u64 *__restrict__ ptotal, PartGeom g, u64 *__restrict__ bstart,
                                                          uint32_t *__restrict__ pcap, u64 *__restrict__ ovf, uint32_t skew_x,
                                                          uint8_t *__restrict__ heavy, u64 heavy_room) {
    __shared__ u64 s_base[MAX_P1 + 1];
    __shared__ uint32_t s_cap[MAX_P1];
    __shared__ u64 s_scan[1024];
    const int tid = threadIdx.x;
    const int P1 = 1 << g.p1_bits;
    const uint32_t P2 = g.b2;
    const u64 mine = tid < P1 ? ptotal[tid] : 0ull;
    u64 tot = 0;
    (void)block_scan_1024(mine, s_scan, tid, &tot);
    const u64 limit = skew_x ? (u64)skew_x * (tot / P1) + (1u << 20) : ~0ull;
    const bool hv = tid < P1 && mine > limit;
    uint32_t cap = 0;
    const bool first = blockIdx.x == 0;  // (every workgroup makes the plan -- four scans -- for its share of bstart[]; one publishes it)
    if (tid < P1) {
        const u64 m = (mine + P2 - 1) / P2;
        cap = hv ? 0u : (uint32_t)((m + (m >> 2) + 1024 + 31) & ~31ull);
        s_cap[tid] = cap;
        if (first) {
            heavy[tid] = hv ? 1 : 0;
            pcap[tid] = cap;
        }
    }
    u64 end = 0, nh = 0, ht = 0;
    const u64 base = block_scan_1024((u64)cap * P2, s_scan, tid, &end);
    (void)block_scan_1024(hv ? 1ull : 0ull, s_scan, tid, &nh);
    (void)block_scan_1024(hv ? mine : 0ull, s_scan, tid, &ht);
    if (tid < P1) s_base[tid] = base;
    if (tid == 0) {
        s_base[P1] = end;
        if (first) {
            ovf[0] = 0;
            ovf[1] = ht > heavy_room ? 2 : 0;
            ovf[2] = nh;
            ovf[3] = ht;
        }
    }
    __syncthreads();
    // bstart[] of every bucket: this workgroup's share (the launch has one workgroup per 1024 buckets, or fewer: they stride)
    const u64 nb = (u64)P1 * P2;
    for (u64 r = (u64)blockIdx.x * 1024 + tid; r < nb; r += (u64)gridDim.x * 1024) {
        const uint32_t p = part_div_b2(g, r), bk = (uint32_t)r - p * P2;
        bstart[r] = s_base[p] + (u64)bk * s_cap[p];
    }
    if (blockIdx.x == 0 && tid == 0) bstart[nb] = s_base[P1];
}

struct OvfEntry {
    uint32_t region;
    uint32_t pad;
    u64 pay;
};

// UNITB: bytes a flush writes at a time -- 128 (whole lines: 3.6 instead of 2.8 TB/s for such appends,
// tools/ubench/scatter_runs.hip) for 4-byte payloads, 64 for 8-byte ones (a bin is 256 bytes either way, and what a
// flush keeps back has to leave room for a half batch's arrivals).
// NBK: 512 (2^5 .. 2^9 buckets per partition) or 1024 (2^10: tables of 2^20 regions -- 2^32 slots, what an hg38-sized or an
// unhinted 2-billion-key input gets; round 2 sent those to the exact path with the unaligned scatter).  The 128 KiB of
// bins are shared out among the partition's buckets either way: 32 4-byte (16 8-byte) payloads per bin at 1024, units of
// 64 bytes there, and the ranks that do not fit such a small bin (Poisson tail of ~8 arrivals per flush) take the overflow list.
// IT (round 6): the type of the payloads the pool holds, where that is not PT -- u64 with PT = uint32_t: LEVEL 2 NARROWS.  An 8-byte
// payload is the hash below the level-1 digit (part_common.hip.h Pay<u64>); once level 2 has put it into its bucket, the region is
// known, and in a table of 2^R regions only 2k - R hash bits are left -- 31 at k = 25 in the headline's 2^19 regions.  Where they
// fit 32 bits (and b2 is a power of two: the bits below the region index are then a bit field), the bins, the arenas and
// everything behind them hold THAT word: half the bytes written here and read by the region pass, which is then the 32-bit
// kernel -- over the virtual geometry (p1_bits = R, b2 = 1), for which such a word is exactly the payload it expects (batch.hip).
template <typename PT, int UNITB, int NBK, bool POW2, typename IT = PT>
__global__ __launch_bounds__(P2L_NT) void part2_arena_kernel(ChunkSrc cs, const u64 *__restrict__ pstart, PartGeom g,
                                                             const u64 *__restrict__ bstart, const uint32_t *__restrict__ pcap,
                                                             PT *__restrict__ out, u64 *__restrict__ bend,
                                                             OvfEntry *__restrict__ ovf_list, u64 *__restrict__ ovf, u64 ovf_cap,
                                                             const uint8_t *__restrict__ heavy) {
    constexpr bool NARROWS = !std::is_same<IT, PT>::value;
#ifndef KH_ARENA_ONE_FLUSH
#define KH_ARENA_ONE_FLUSH 1  // (0: A/B builds -- two flushes per batch whatever the payload)
#endif
    constexpr bool ONE_FLUSH = KH_ARENA_ONE_FLUSH != 0 && sizeof(IT) == 8 && (NARROWS || NBK != 1024);
    static_assert(!NARROWS || (sizeof(IT) == 8 && sizeof(PT) == 4 && POW2), "level 2 narrows 8-byte payloads to 4 bytes, power-of-two geometries");
    constexpr int UNIT = UNITB / (int)sizeof(PT), NT = P2L_NT, PER = P2L<IT>::PER, TILE = P2L<IT>::TILE;
    // what a payload is kept as in the bins / arenas / overflow list: itself, or (NARROWS) the 32 bits behind its bucket digit
    auto kept = [&](IT w) -> PT {
        if constexpr (NARROWS) return (PT)(((u64)w << g.p2_bits) >> 32);
        else return (PT)w;
    };
    constexpr int HALF = PER / 2;
    constexpr uint32_t CAP = 256 / sizeof(PT);                // payloads per bin at 512 buckets (256 bytes)
    // payloads all bins hold together: 128 KiB -- 144 KiB in the 768-bucket instance (round 5), which is what the CU's 160 KiB
    // leave beside the counters, the chunk list and the positions: at 640 buckets a bin then holds 56 payloads instead of 48, and
    // that is the room a whole-LINE unit (32 payloads) needs beside a half batch's arrivals (12.8 +- 3.6: with 48 the bins
    // overflowed into the list 1.5 % of the time and ate what the 128-byte units had won -- profiles/README.md r04a, 2)
    constexpr uint32_t TOTAL = NBK == 768 ? (144u << 10) / (uint32_t)sizeof(PT) : P2L_NBK * CAP;
    constexpr uint32_t UW = UNITB / 16;                       // 16-byte words per unit
    static_assert(NBK == 512 || NBK == 768 || (NBK == 1024 && UNITB == 64), "1024 buckets: 128-byte bins, 64-byte units");
    __shared__ __attribute__((aligned(16))) PT s_bin[TOTAL + UNIT];  // 128 KiB (+ a trash unit)
    __shared__ uint32_t s_cnt[NBK];
    __shared__ uint32_t s_chk[CPB];
    __shared__ uint16_t s_cfill[CPB];
    __shared__ u64 s_ovf_next, s_ovf_end;  // the workgroup's private segment of the overflow list (none to begin with)
    __shared__ uint32_t s_ovf_want;
    constexpr u64 OVF_SEG = 8192, OVF_LOW = 2048;  // (a request that finds the segment short is served from the global cursor)
    const uint32_t p = blockIdx.x;
    const int tid = threadIdx.x;
    // The batch is not for this path (the plan found a partition too heavy), or is lost to it already (another workgroup
    // found the overflow list full): nothing to do, the host takes the exact path.  Decided by ONE lane for the whole
    // workgroup -- the flag can change while the lanes are reading it, and a workgroup must not split at a barrier.
    __shared__ uint32_t s_skip;
    if (tid == 0) s_skip = ovf[1] != 0 || heavy[p] != 0;  // (a heavy partition: the exact kernels take it, see arena_plan_kernel)
    __syncthreads();
    if (s_skip) return;
    const uint32_t P2 = g.b2;
    // payloads per bin: the 128 KiB are shared out among the partition's P2 buckets (whole 16-byte words)
    const uint32_t capr = POW2 ? TOTAL >> g.p2_bits : (TOTAL / P2) & ~(16u / (uint32_t)sizeof(PT) - 1u);
    if (tid == 0) {
        s_ovf_next = 0;
        s_ovf_end = 0;
        s_ovf_want = 0;
    }
    // LP consecutive lanes own a bucket together: all keep its arena and how much of it is written (a multiple of UNIT
    // until the end); of every unit lane i stores the 16-byte words i, i + LP, ... -- so that ONE store instruction covers
    // LP x 16 contiguous bytes (measured, profiles/README.md r02g: the more of a unit one instruction stores, the fewer
    // write requests the memory side sees and the faster the kernel; one lane per unit: 152 GB written and 54 ms, two: 120 GB
    // and 27.6 ms, four: 65 GB and 21.7 ms).  1024 lanes / LP < 512 buckets: a lane group owns NB buckets, NT / LP apart.
    constexpr uint32_t LP = KH_ARENA_LANES < UW ? KH_ARENA_LANES : UW;
    constexpr uint32_t NB = (NBK * LP + NT - 1) / NT;  // buckets per lane group
    const uint32_t og = (uint32_t)tid / LP, oi = (uint32_t)tid % LP;
    // A lane group's buckets' write positions: in registers (512 buckets: two per group).  With 1024 buckets a group owns
    // four; four more base / position pairs beside the two payload sets pushed the kernel over the 128 registers a
    // 1024-lane workgroup has -- 1.1 KB of scratch per lane and 525 ms for a 125 M-read batch (measured, round 3) -- so
    // there the positions live in LDS (s_apos) and the bases are computed: a partition's arenas are equally large and
    // consecutive (arena_plan_kernel), bucket b's starts at the partition's first + b x capacity.
#ifndef KH_ARENA_POS_LDS
#define KH_ARENA_POS_LDS 0  // 1: positions in LDS for 512 buckets too (A/B builds)
#endif
    // Round 4 (any number of buckets up to 1024, kernels.hip.h TableGeom): NBK = 768 for 513 .. 768 buckets -- three buckets per
    // lane group instead of four, 40-48 payloads per bin instead of 32.
    constexpr bool POS_LDS = NBK > 512 || KH_ARENA_POS_LDS != 0;  // (in registers they spill there: 222 / 518 VGPRs at 768 / 1024 buckets, -Rpass-analysis)
    constexpr bool BASE_CALC = POS_LDS;
    __shared__ uint32_t s_apos[POS_LDS ? NBK : 1];
    const uint32_t acap = pcap[p];
    const u64 pbase = bstart[(u64)p * P2];
    u64 abase_r[BASE_CALC ? 1 : NB];
    uint32_t apos_r[POS_LDS ? 1 : NB];
    if constexpr (POS_LDS) {
        if (tid < NBK) s_apos[tid] = 0;
    } else {
#pragma unroll
        for (uint32_t it = 0; it < NB; ++it) {
            const uint32_t ob = og + it * (NT / LP);
            if constexpr (!BASE_CALC) abase_r[it] = ob < P2 ? bstart[(u64)p * P2 + ob] : 0;
            apos_r[it] = 0;
        }
    }
    auto abase_of = [&](uint32_t it, uint32_t ob) -> u64 {
        if constexpr (BASE_CALC) return pbase + (u64)ob * acap;
        else return abase_r[it];
    };
    if (tid < NBK) s_cnt[tid] = 0;
    const uint32_t woff = (uint32_t)tid & (CHUNK_PAY - 1);  // offset inside the chunk
    // Appends k entries of this lane to the overflow list; false if the list is full (the host then redoes the batch).
    // On skewed input MANY lanes of MANY workgroups do this: the list is handed out in private segments (one global
    // atomic per OVF_SEG entries and workgroup, LDS atomics inside), a request that finds the segment short takes its
    // entries straight from the global cursor, the unused tail of a segment is marked invalid (region = ~0).
    auto ovf_take = [&](uint32_t k, u64 &at) -> bool {
        at = atomicAdd(&s_ovf_next, (u64)k);  // LDS
        if (at + k > s_ovf_end) {
            // (the request that crosses the segment's end leaves [at, end) unused: nobody else will write or mark those
            //  entries -- the cursor is past them -- and they may hold an earlier batch's entries)
            for (u64 i = at; i < s_ovf_end && i < ovf_cap; ++i) ovf_list[i].region = 0xFFFFFFFFu;
            s_ovf_want = 1u;
            at = atomicAdd(&ovf[0], (u64)k);
        }
        if (at + k > ovf_cap) {
            ovf[1] = 1;
            return false;
        }
        return true;
    };
    // (between barriers, nobody appending) a fresh segment when one was missed or the current one runs low
    auto ovf_refill = [&]() {
        if (!s_ovf_want && s_ovf_next + OVF_LOW <= s_ovf_end) return;  // uniform: read after a barrier
        const u64 tail0 = s_ovf_next, tail1 = s_ovf_end;
        for (u64 i = tail0 + tid; i < tail1 && i < ovf_cap; i += NT) ovf_list[i].region = 0xFFFFFFFFu;
        __syncthreads();
        if (tid == 0) {
            s_ovf_next = atomicAdd(&ovf[0], (u64)OVF_SEG);
            s_ovf_end = s_ovf_next + OVF_SEG;
            s_ovf_want = 0u;
        }
        __syncthreads();
    };
    for (u64 blo = pstart[p]; blo < pstart[p + 1]; blo += CPB) {
        Part2Block pb;
        pb.lo = blo;
        pb.hi = min(blo + (u64)CPB, pstart[p + 1]);
        __syncthreads();  // (the previous block's batches are done with s_chk / s_cfill)
        p2_stage_chunks<true>(cs, pb, s_chk, s_cfill, tid, NT);
        __syncthreads();
        const uint32_t nchk = (uint32_t)(pb.hi - pb.lo);
        const uint32_t n = nchk * CHUNK_PAY;
        auto load_batch = [&](uint32_t first, IT (&pay)[PER], uint32_t &have) {  // (as in part2_scatter_lines_kernel)
            have = 0;
            const uint32_t myci = (first >> 8) + ((uint32_t)tid & (PER - 1)) * (NT / CHUNK_PAY) + ((uint32_t)tid >> 8);
            const uint32_t mycc = myci < nchk ? myci : nchk - 1;
            const uint32_t mychunk = s_chk[mycc];
            const uint32_t myfill = myci < nchk ? (uint32_t)s_cfill[mycc] : 0u;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t chunk = __builtin_amdgcn_readlane(mychunk, j);
                const uint32_t fillc = __builtin_amdgcn_readlane(myfill, j);
                const bool ok = woff < fillc;
                pay[j] = reinterpret_cast<const IT *>(cs.pay)[(u64)chunk * CHUNK_PAY + (ok ? woff : 0u)];
                have |= (uint32_t)ok << j;
            }
        };
        // the lane group writes the whole units of its buckets' bins to their arenas (or, an arena full, its first lane to the overflow list)
        // (A/B, round 5, hg-shaped input through the 1024-bucket instance: interleaving two buckets' flushes, or skipping the
        //  bins that hold less than a unit, changed nothing -- 10.2-10.3 ms either way; the instance is bound by the chain of its
        //  phases between barriers with ONE workgroup per CU, like the 512-bucket one, at half the payloads per flush)
        auto flush = [&]() {
#pragma unroll POS_LDS ? 1 : NB
            for (uint32_t it = 0; it < NB; ++it) {
                const uint32_t ob = og + it * (NT / LP);
                if (ob >= P2) continue;
                const u64 abase = abase_of(it, ob);
                uint32_t apos;
                if constexpr (POS_LDS) apos = s_apos[ob];
                else apos = apos_r[it];
                uint4 *const bin4 = reinterpret_cast<uint4 *>(s_bin + ob * capr);
                const uint32_t c = min(s_cnt[ob], capr);  // (ranks beyond the bin went to the overflow list)
                const uint32_t nun = c / UNIT, r = c % UNIT;
                for (uint32_t u = 0; u < nun; ++u) {
                    if (apos + UNIT <= acap) {
                        uint4 *d = reinterpret_cast<uint4 *>(out + abase + apos);
                        // (named values, not a local array: the compiler kept `uint4 x[2]` on the stack and stored its
                        //  second word to scratch with every unit -- 16 dead bytes per lane and unit, found in the ISA)
                        if constexpr (UW / LP == 2) {
                            const uint4 x0 = bin4[UW * u + oi], x1 = bin4[UW * u + oi + LP];
                            d[oi] = x0;
                            d[oi + LP] = x1;
                        } else if constexpr (UW / LP == 1) {
                            d[oi] = bin4[UW * u + oi];
                        } else {
                            uint4 x[UW / LP];
#pragma unroll
                            for (uint32_t q = 0; q < UW / LP; ++q) x[q] = bin4[UW * u + oi + LP * q];
#pragma unroll
                            for (uint32_t q = 0; q < UW / LP; ++q) d[oi + LP * q] = x[q];
                        }
                        apos += UNIT;
                    } else {
                        // The arena is full (a heavy hitter's bucket: every later unit of it comes this way): the group's first
                        // lane takes the room, ALL its lanes write entries -- consecutive lanes consecutive entries.  (One lane
                        // writing the unit's 16 or 32 entries one after the other was a chain of as many stores on the way to
                        // the barrier, in some wave at nearly every flush of an hg-shaped batch: 4 % of its payloads come here.)
                        u64 at = 0;
                        bool room = false;
                        if (oi == 0) room = ovf_take(UNIT, at);
                        const int src = (int)((uint32_t)tid & 63u & ~(LP - 1u));
                        at = (u64)__shfl((unsigned long long)at, src, 64);
                        room = __shfl((int)room, src, 64) != 0;
                        if (room) {
                            const PT *e = s_bin + ob * capr + u * UNIT;
                            for (uint32_t q = oi; q < (uint32_t)UNIT; q += LP) {
                                OvfEntry oe;
                                oe.region = p * P2 + ob;
                                oe.pad = 0;
                                oe.pay = (u64)e[q];
                                ovf_list[at + q] = oe;
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (nun) {
                    const uint32_t nw = (r * (uint32_t)sizeof(PT) + 15u) / 16u;
                    uint4 m[UW / LP];
#pragma unroll
                    for (uint32_t q = 0; q < UW / LP; ++q) m[q] = oi + LP * q < nw ? bin4[UW * nun + oi + LP * q] : make_uint4(0, 0, 0, 0);
#pragma unroll
                    for (uint32_t q = 0; q < UW / LP; ++q)
                        if (oi + LP * q < nw) bin4[oi + LP * q] = m[q];
                }
                if (oi == 0) s_cnt[ob] = r;
                if constexpr (POS_LDS) {
                    if (oi == 0) s_apos[ob] = apos;  // (the group's lanes read it together at the top: same wave, next flush is behind barriers)
                } else {
                    apos_r[it] = apos;
                }
            }
        };
        // FULL: every lane of the wave holds all its payloads of the batch -- nineteen chunks in twenty are full -- and the
        // ranks are taken without the per-payload test (and-compare-saveexec-restore around every LDS atomic: from the ISA)
        auto batch_as = [&](auto full_tag, uint32_t base, IT (&pay)[PER], uint32_t have, IT (&nxt)[PER], uint32_t &have_nxt) {
            constexpr bool FULL = decltype(full_tag)::value;
            load_batch(base + TILE, nxt, have_nxt);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t rk[HALF], dg[HALF];
                // (a partition of the arena path has >= 32 buckets: the POW2 digit is a plain shift, no "no digit at all" case)
                auto digit = [&](IT w) -> uint32_t {
                    if constexpr (sizeof(IT) == 4 && POW2) return (uint32_t)w >> (32u - g.p2_bits);
                    else if constexpr (sizeof(IT) == 4) return part_bucket32<POW2>((uint32_t)w, g);
                    else return Pay<IT>::p2(w, g);
                };
#pragma unroll
                for (int j = 0; j < HALF; ++j) {
                    pay[h * HALF + j] = l2_finish(pay[h * HALF + j], p, g);  // (level 1 left the last Feistel round to us)
                    dg[j] = digit(pay[h * HALF + j]);
                    if constexpr (FULL) rk[j] = atomicAdd(&s_cnt[dg[j]], 1u);
                    else rk[j] = ((have >> (h * HALF + j)) & 1u) ? atomicAdd(&s_cnt[dg[j]], 1u) : 0xFFFFFFFFu;
                }
                uint32_t omask = 0;
#pragma unroll
                for (int j = 0; j < HALF; ++j) {
                    const uint32_t r = rk[j];
                    s_bin[r < capr ? dg[j] * capr + r : TOTAL] = kept(pay[h * HALF + j]);
                    omask |= (r != 0xFFFFFFFFu && r >= capr) ? (1u << j) : 0u;
                }
                if (kh_any(omask != 0)) {  // ranks that did not fit their bins (a heavy bucket): straight to the overflow list
                    const uint32_t k = (uint32_t)__builtin_popcount(omask);
                    // ONE request per wave (round 5): the lanes' counts are scanned in the wave and its last lane asks for the
                    // sum -- with repeats in the input (4 % of an hg-shaped batch's payloads come here) some twenty lanes of
                    // every wave would otherwise queue up at the same LDS word in every half batch.
                    uint32_t incl = k;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
                        if ((int)lane_id() >= o) incl += up;
                    }
                    const uint32_t wtotal = (uint32_t)__shfl((int)incl, 63, 64);
                    u64 wat = 0;
                    bool wok = false;
                    if (lane_id() == 63u) wok = ovf_take(wtotal, wat);
                    wat = (u64)__shfl((unsigned long long)wat, 63, 64);
                    wok = __shfl((int)wok, 63, 64) != 0;
                    u64 at = wat + (incl - k);
                    if (k && wok) {
                        uint32_t q = 0;
#pragma unroll
                        for (int j = 0; j < HALF; ++j)
                            if ((omask >> j) & 1u) {
                                OvfEntry oe;
                                oe.region = p * P2 + dg[j];
                                oe.pad = 0;
                                oe.pay = (u64)kept(pay[h * HALF + j]);
                                ovf_list[at + q++] = oe;
                            }
                    }
                }
                // ONE FLUSH PER BATCH for 8-byte input (round 6): a batch of those is half as many payloads as a 4-byte one (the same
                // registers), so a half batch brings a bin 4-8 arrivals where it has room for 17-33 beside what a flush keeps back --
                // and a flush is a barrier pair and the whole flush code in every wave whatever it finds (level1_64.hip: the same
                // lesson).  Both halves are ranked, then the bins are flushed once.  (Not the 1024-bucket instance of 8-byte OUTPUT:
                // its bins hold 16.)
                if (ONE_FLUSH && h == 0) continue;
#if !(KH_ABL_ARENA & 2)  /* timing experiment otherwise (wrong results): no barrier between the ranks and the flush */
                __syncthreads();  // B1
#endif
                if (ONE_FLUSH || h == 0) {  // the wait for the next batch's payloads goes HERE, before the first store (vmcnt: see part1_bins_kernel)
#pragma unroll
                    for (int j = 0; j < PER; ++j) asm volatile("" : "+v"(nxt[j]));
                }
                flush();
#if !(KH_ABL_ARENA & 1)  /* timing experiment otherwise (wrong results): no barrier between the flush and the next half batch's ranks */
                __syncthreads();  // B2
#endif
                if (s_ovf_want || s_ovf_end) ovf_refill();  // (uniform; skipped entirely while nothing has overflowed)
            }
        };
        auto batch = [&](uint32_t base, IT (&pay)[PER], uint32_t have, IT (&nxt)[PER], uint32_t &have_nxt) {
            // (decided per wave: both forms pass the same barriers)
            if (!kh_any(have != (1u << PER) - 1u)) batch_as(std::true_type{}, base, pay, have, nxt, have_nxt);
            else batch_as(std::false_type{}, base, pay, have, nxt, have_nxt);
        };
        IT payA[PER], payB[PER];
        uint32_t haveA = 0, haveB = 0;
        load_batch(0, payA, haveA);
#pragma unroll
        for (int j = 0; j < PER; ++j) asm volatile("" : "+v"(payA[j]));
        for (uint32_t base = 0; base < n; base += 2 * TILE) {
            batch(base, payA, haveA, payB, haveB);
            if (base + TILE < n) batch(base + TILE, payB, haveB, payA, haveA);
        }
    }
    // what is left in the bins (< UNIT payloads per bucket), one by one; then the bucket's end
    __syncthreads();  // (a partition without chunks comes here straight from the initialisation of the counters)
#pragma unroll POS_LDS ? 1 : NB
    for (uint32_t it = 0; it < NB; ++it) {
        const uint32_t ob = og + it * (NT / LP);
        if (oi != 0 || ob >= P2) continue;
        const uint32_t r = s_cnt[ob];
        const PT *bin = s_bin + ob * capr;
        const u64 abase = abase_of(it, ob);
        uint32_t apos;
        if constexpr (POS_LDS) apos = s_apos[ob];
        else apos = apos_r[it];
        if (r) {
            if (apos + r <= acap) {
                for (uint32_t i = 0; i < r; ++i) out[abase + apos + i] = bin[i];
                apos += r;
            } else {
                u64 at;
                if (ovf_take(r, at))
                    for (uint32_t i = 0; i < r; ++i) {
                        OvfEntry oe;
                        oe.region = p * P2 + ob;
                        oe.pad = 0;
                        oe.pay = (u64)bin[i];
                        ovf_list[at + i] = oe;
                    }
            }
        }
        bend[(u64)p * P2 + ob] = abase + apos;
    }
    __syncthreads();
    for (u64 i = s_ovf_next + tid; i < s_ovf_end && i < ovf_cap; i += NT) ovf_list[i].region = 0xFFFFFFFFu;
}

// touched[region] = 1 for every (valid) entry of the overflow list: the regions whose exchange-head counts the list's insert
// changes (batch.hip counts those regions again instead of dropping the region pass's counts for the whole table)
KH_GLOBAL __launch_bounds__(BLOCK) void ovf_touch_kernel(const OvfEntry *__restrict__ list, const u64 *__restrict__ ovf, u64 ovf_cap,
                                                          uint8_t *__restrict__ touched) {
    const u64 n = ovf[0] < ovf_cap ? ovf[0] : ovf_cap;
    const u64 stride = (u64)gridDim.x * BLOCK;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const uint32_t r = list[i].region;
        if (r != 0xFFFFFFFFu) touched[r] = 1;
    }
}

// ---- the 8-byte table image (round 3): slot = count << 32 | 32-bit payload, 0 = free --------------------------------
// Written by region_count_kernel32<.., NARROW = true>; region r = slots [4096 r, 4096 (r + 1)), in-region start as in the
// 16-byte table (the same hash bits), linear probing inside the region.  The key of slot i:
__device__ __forceinline__ u64 narrow_key(const PartGeom &g, u64 slot_index, uint32_t pay) {
    const uint32_t r = (uint32_t)(slot_index >> REGION_BITS);
    return Pay<uint32_t>::key(pay, part_div_b2(g, r), g);
}
__device__ __forceinline__ uint32_t narrow_start(const PartGeom &g, uint32_t pay) {
    return kh_start_of_x(pay, g.b2);
}
// count[pay] += addend in the narrow image of one region; false = the count might leave 32 bits (nothing changed).
// `guard` = an upper bound of everything the running kernel may still add to one key (the length of its list): a count
// below 2^32 - guard takes a plain fire-and-forget atomic add -- a compare-and-swap loop on a heavy hitter's slot is a
// retry storm (measured, round 3: an hg-shaped batch spent 2.6 s in it, against 8 ms with the add).
__device__ __forceinline__ bool narrow_upsert(u64 *nreg, const PartGeom &g, uint32_t pay, u64 addend, u64 guard, uint32_t &ndistinct,
                                              uint32_t &nfailed) {
    uint32_t off = narrow_start(g, pay);
    for (uint32_t probes = 0; probes < REGION_SLOTS; ++probes, off = (off + 1) & REGION_MASK) {
        u64 cur = __hip_atomic_load(&nreg[off], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((cur >> 32) == 0) {
            const u64 old = atomicCAS(&nreg[off], 0ull, (addend << 32) | pay);
            if (old == 0) {
                ++ndistinct;
                return true;
            }
            cur = old;  // someone else claimed it first; may be our payload
        }
        if ((uint32_t)cur == pay) {
            if ((cur >> 32) + guard >= 0xFFFFFFFFull) return false;
            (void)__hip_atomic_fetch_add(&nreg[off], addend << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the count half only
            return true;
        }
    }
    ++nfailed;  // region full: the host keeps the load factor far below this
    return true;
}

// The overflow list of a batch with repeats is mostly copies (a tandem repeat sends the same few payloads into one bin in a
// burst, a repeat family's copies overflow their bucket's arena; an hg-shaped 3 Gbp input hands out ~10^8 entries): every
// workgroup sums slices of 8192 entries in an LDS table keyed by (region, payload) and applies the sums -- an atomic per key and
// table flush instead of one per entry.  4-byte payloads only (region and payload are one 64-bit key).  NARROW: only where no
// count can leave 32 bits in the 8-byte image (the host checks the table's k-mer total), so that nothing has to stay in the list.
// 512 lanes per workgroup (round 5): the 48 KiB table allows three workgroups per CU either way, and with 256 lanes that was
// twelve waves per CU for a kernel made of LDS round trips: 2.4 -> 1.5 ms for the hg-shaped batch's 1.3e8 entries (1024 lanes,
// two workgroups per CU: 2.2).
constexpr int OVF_AGG_NT = 512;
template <bool NARROW>
__global__ __launch_bounds__(OVF_AGG_NT) void ovf_agg_insert_kernel(TableGeom tg, PartGeom g, const OvfEntry *__restrict__ list,
                                                               const u64 *__restrict__ ovf, u64 ovf_cap, Counters *ctr, u64 *__restrict__ ntab) {
    constexpr u64 SLICE = 8192;
    constexpr int TAB = 4096, NT = OVF_AGG_NT, PER = 1024 / NT;
    constexpr u64 FREE = ~0ull;  // (a valid entry's region is never 0xFFFFFFFF: that marks the unused entries of a segment)
    __shared__ u64 s_key[TAB];
    __shared__ uint32_t s_cnt[TAB];
    __shared__ uint32_t s_fill;
    const int tid = threadIdx.x;
    const u64 n = ovf[0] < ovf_cap ? ovf[0] : ovf_cap;  // (the cursor moves in whole segments: see ovf_insert_kernel)
    for (int i = tid; i < TAB; i += NT) {
        s_key[i] = FREE;
        s_cnt[i] = 0;
    }
    if (tid == 0) s_fill = 0;
    __syncthreads();
    uint32_t nd = 0, nf = 0;
    u64 km = 0;
    auto apply = [&]() {  // the LDS table -> the table in HBM; leaves it empty (all threads)
        __syncthreads();
        for (int i = tid; i < TAB; i += NT) {
            const uint32_t cnt = s_cnt[i];
            if (cnt) {
                const u64 key = s_key[i];
                const uint32_t region = (uint32_t)(key >> 32), pay = (uint32_t)key;
                if constexpr (NARROW) (void)narrow_upsert(ntab + (u64)region * REGION_SLOTS, g, pay, (u64)cnt, 0ull, nd, nf);
                else upsert(tg, Pay<uint32_t>::key(pay, part_div_b2(g, region), g), (u64)cnt, nd, nf);
                s_key[i] = FREE;
                s_cnt[i] = 0;
            }
        }
        if (tid == 0) s_fill = 0;
        __syncthreads();
    };
    for (u64 s0 = (u64)blockIdx.x * SLICE; s0 < n; s0 += (u64)gridDim.x * SLICE) {  // (uniform per workgroup: barriers inside)
        const u64 s1 = s0 + SLICE < n ? s0 + SLICE : n;
        for (u64 c0 = s0; c0 < s1; c0 += PER * NT) {  // 1024 entries between two looks at the fill
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const u64 i = c0 + (u64)j * NT + tid;
                if (i >= s1) continue;
                const OvfEntry e = list[i];
                if (e.region == 0xFFFFFFFFu) continue;
                ++km;
                const u64 key = ((u64)e.region << 32) | (uint32_t)e.pay;
                uint32_t h = ((uint32_t)(key ^ (key >> 29)) * 2654435761u) >> 20;  // 12 bits
                for (;;) {
                    u64 cur = s_key[h];
                    if (cur == FREE) {
                        cur = atomicCAS(&s_key[h], FREE, key);
                        if (cur == FREE) {
                            atomicAdd(&s_fill, 1u);
                            cur = key;
                        }
                    }
                    if (cur == key) {
                        atomicAdd(&s_cnt[h], 1u);
                        break;
                    }
                    h = (h + 1) & (TAB - 1);
                }
            }
            __syncthreads();
            const bool full = s_fill > TAB / 2;  // (uniform: read between two barriers; at most 1024 more before the next look)
            __syncthreads();
            if (full) apply();
        }
    }
    apply();
    const u64 d = wave_sum((u64)nd), f = wave_sum((u64)nf);
    km = wave_sum(km);
    if (lane_id() == 0) {
        if (d) atomicAdd(&ctr->distinct, d);
        if (f) atomicAdd(&ctr->failed, f);
        if (km) atomicAdd(&ctr->kmers, km);
    }
}

// ntab -> the 16-byte table (every slot of it is written)
KH_GLOBAL __launch_bounds__(BLOCK) void ntable_widen_kernel(const u64 *__restrict__ ntab, u64 cap, PartGeom g, Slot *__restrict__ table) {
    const u64 stride = (u64)gridDim.x * BLOCK;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < cap; i += stride) {
        const u64 sl = ntab[i];
        const u64 cnt = sl >> 32;
        const u64 key = cnt ? narrow_key(g, i, (uint32_t)sl) : (u64)KH_EMPTY_KEY;
        *reinterpret_cast<uint4 *>(&table[i]) = make_uint4((uint32_t)key, (uint32_t)(key >> 32), (uint32_t)cnt, 0u);
    }
}
// the narrow twins of table_count_kernel / table_compact_kernel / table_hist_kernel / table_lookup_kernel (kernels.hip.h)
KH_GLOBAL __launch_bounds__(BLOCK) void ntable_count_kernel(const u64 *__restrict__ ntab, u64 cap, u64 min_count, Counters *ctr) {
    const u64 stride = (u64)gridDim.x * BLOCK;
    u64 n = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < cap; i += stride) {
        const u64 cnt = ntab[i] >> 32;
        n += (cnt != 0 && cnt >= min_count) ? 1 : 0;
    }
    n = wave_sum(n);
    if (lane_id() == 0 && n) atomicAdd(&ctr->cursor, n);
}
KH_GLOBAL __launch_bounds__(BLOCK) void ntable_compact_kernel(const u64 *__restrict__ ntab, u64 cap, PartGeom g, u64 min_count, u64 *keys,
                                                               u64 *counts, u64 out_cap, Counters *ctr) {
    compact_tiles(cap, keys, counts, out_cap, ctr, [&](u64 i, u64 &key, u64 &count) {
        const u64 sl = ntab[i];
        count = sl >> 32;
        const bool live = count != 0 && count >= min_count;
        key = live ? narrow_key(g, i, (uint32_t)sl) : 0ull;  // (the inverse hash only for what goes out)
        return live;
    });
}
KH_GLOBAL __launch_bounds__(BLOCK) void ntable_hist_kernel(const u64 *__restrict__ ntab, u64 cap, u64 min_count, u64 *dense, u64 *big,
                                                            u64 big_cap, Counters *ctr) {
    __shared__ uint32_t s_bins[HIST_LDS];
    for (uint32_t i = threadIdx.x; i < HIST_LDS; i += BLOCK) s_bins[i] = 0;
    __syncthreads();
    // Two slots per lane and load; and the counts nearly every key of a real table has -- 1, 2, 3: error k-mers, the unique
    // k-mers of a genome -- are tallied by ballot and popcount in the wave's own registers: sixty-four lanes adding to ONE LDS word
    // is sixty-four serialised atomics (round 5: 10.3 ms for the 34 GB image of an hg-shaped table, 3.3 TB/s).
    const u64 stride = (u64)gridDim.x * BLOCK;
    const uint4 *nt2 = reinterpret_cast<const uint4 *>(ntab);
    uint32_t w1 = 0, w2 = 0, w3 = 0;
    auto tally = [&](u64 cnt) {
        const bool live = cnt != 0 && cnt >= min_count;
        w1 += (uint32_t)__builtin_popcountll(kh_ballot(live && cnt == 1));
        w2 += (uint32_t)__builtin_popcountll(kh_ballot(live && cnt == 2));
        w3 += (uint32_t)__builtin_popcountll(kh_ballot(live && cnt == 3));
        if (!live || cnt <= 3) return;
        if (cnt < HIST_LDS) {
            atomicAdd(&s_bins[(uint32_t)cnt], 1u);
        } else if (cnt < HIST_DENSE) {
            atomicAdd(&dense[cnt], 1ull);
        } else {
            const u64 o = atomicAdd(&ctr->big, 1ull);
            if (o < big_cap) big[o] = cnt;
        }
    };
    const u64 pairs = cap >> 1;  // (a table is whole regions: an even number of slots)
    const u64 rounds = (pairs + stride - 1) / stride;
    for (u64 r = 0; r < rounds; ++r) {  // (uniform trip count: every lane takes part in the ballots)
        const u64 i = r * stride + (u64)blockIdx.x * BLOCK + threadIdx.x;
        const uint4 v = i < pairs ? nt2[i] : make_uint4(0u, 0u, 0u, 0u);
        tally((u64)v.y);
        tally((u64)v.w);
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
KH_GLOBAL __launch_bounds__(BLOCK) void ntable_lookup_kernel(const u64 *__restrict__ ntab, PartGeom g, const u64 *keys, u64 n, u64 *out) {
    const u64 stride = (u64)gridDim.x * BLOCK;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const u64 key = keys[i];
        if (key & ~kh_kmask(g.k)) {  // not a k-mer of this k (the hash only looks at 2k bits: it must not alias one)
            out[i] = 0;
            continue;
        }
        // (a shard's image identifies a key by the hash bits BELOW the owner's: a key of another shard must not alias one of this)
        if (g.shard_shift && (kh_table_hash(key, g.k) >> (64 - g.shard_shift)) != g.shard_index) {
            out[i] = 0;
            continue;
        }
        const u64 H = part_hash(g, key);
        const uint32_t pay = Pay<uint32_t>::make(key, H, g);
        const u64 region = (u64)kh_p1_of(H, g.p1_bits) * g.b2 + kh_bucket_of_x(pay, g.b2);
        const u64 *reg = ntab + region * REGION_SLOTS;
        uint32_t off = narrow_start(g, pay);
        u64 res = 0;
        for (uint32_t probes = 0; probes < REGION_SLOTS; ++probes) {
            const u64 sl = reg[off];
            if ((sl >> 32) == 0) break;
            if ((uint32_t)sl == pay) { res = sl >> 32; break; }
            off = (off + 1) & REGION_MASK;
        }
        out[i] = res;
    }
}

// the overflow list through the direct path (after the region pass: the table then holds the batch's other k-mers).
// NARROW: into the 8-byte image; an entry whose count would not fit stays in the list (ctr->narrow_ovf counts them), every
// other entry is marked consumed (region = ~0), so that the host can widen the table and run the wide form over what is left.
template <typename PT, bool NARROW>
__global__ __launch_bounds__(BLOCK) void ovf_insert_kernel(TableGeom tg, PartGeom g, OvfEntry *__restrict__ list,
                                                           const u64 *__restrict__ ovf, u64 ovf_cap, Counters *ctr, u64 *__restrict__ ntab) {
    // ovf[0] is a CURSOR, advanced by whole segments (part2_arena_kernel, ovf_refill): a workgroup's last segment may
    // straddle or lie beyond the list's end without the "list full" flag ever being raised (nothing was appended there).
    // Entries are only ever written, and unused ones only ever marked invalid, below ovf_cap: never read past it.
    const u64 n = ovf[0] < ovf_cap ? ovf[0] : ovf_cap;
    uint32_t nd = 0, nf = 0;
    u64 km = 0, kept = 0;
    const u64 nround = (n + BLOCK - 1) / BLOCK * BLOCK;  // (whole waves take part in the ballots)
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < nround; i += (u64)gridDim.x * BLOCK) {
        OvfEntry e;
        e.region = 0xFFFFFFFFu;
        e.pay = 0;
        if (i < n) e = list[i];
        const bool valid = e.region != 0xFFFFFFFFu;
        bool mine = valid;
        const u64 key = NARROW ? (((u64)e.region << 32) | (uint32_t)e.pay)  // (identity inside the image: region + payload)
                               : (mine ? Pay<PT>::key((PT)e.pay, part_div_b2(g, e.region), g) : 0ull);
        // skew guard as in count_direct_kernel: an overflow list is mostly copies of a few heavy keys
        u64 weight = 1;
        int first = -1;
        bool absorbed = false;
        const u64 vmask = kh_ballot(mine);
        if (vmask) {
            first = __builtin_ctzll(vmask);
            const u64 lead = __shfl(key, first, 64);
            const bool same = mine && key == lead;
            const u64 smask = kh_ballot(same);
            if (__builtin_popcountll(smask) > 1) {
                if ((int)lane_id() == first) weight = (u64)__builtin_popcountll(smask);
                else if (same) {
                    mine = false;
                    absorbed = true;
                }
            }
        }
        bool ok = true;
        if (mine) {
            if (NARROW) ok = narrow_upsert(ntab + (u64)e.region * REGION_SLOTS, g, (uint32_t)e.pay, weight, n, nd, nf);
            else upsert(tg, key, weight, nd, nf);
        }
        if (NARROW) {
            const bool lead_ok = first >= 0 ? (bool)__shfl((int)ok, first, 64) : true;
            if (absorbed) ok = lead_ok;
            if (valid && ok) list[i].region = 0xFFFFFFFFu;  // consumed
            if (valid && !ok) ++kept;
            km += (valid && ok) ? 1 : 0;
        } else {
            km += valid;
        }
    }
    const u64 d = wave_sum((u64)nd), f = wave_sum((u64)nf);
    km = wave_sum(km);
    kept = wave_sum(kept);
    if (lane_id() == 0) {
        if (d) atomicAdd(&ctr->distinct, d);
        if (f) atomicAdd(&ctr->failed, f);
        if (km) atomicAdd(&ctr->kmers, km);
        if (kept) atomicAdd(&ctr->narrow_ovf, kept);
    }
}

// bstart[r] = first payload of region r's bucket in the level-2 output, r in [0, R]; bstart[R] = total
KH_GLOBAL __launch_bounds__(256) void bucket_bounds_kernel(const u64 *__restrict__ O2, u64 o2_total_index,
                                                            const u64 *__restrict__ moff, const uint32_t *__restrict__ nch,
                                                            PartGeom g, u64 *__restrict__ bstart) {
    const u64 nregions = part_regions(g);
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r > nregions) return;
    if (r == nregions) {
        bstart[r] = O2[o2_total_index];  // grand total of the level-2 scan
        return;
    }
    const uint32_t p1 = part_div_b2(g, r), p2 = (uint32_t)r - p1 * g.b2;
    // an empty level-1 partition has no chunks: its buckets all start where the partition starts,
    // which is the O2 value at the next partition's first entry (or the grand total)
    bstart[r] = nch[p1] ? O2[moff[p1] + (u64)p2 * nch[p1]] : O2[moff[p1]];
}

// The same for the HEAVY partitions of an arena batch: their buckets were written by the exact kernels into the buffer
// behind the arenas (at payload offset `base`); start and end of every bucket of a heavy partition, nothing for the others.
KH_GLOBAL __launch_bounds__(256) void bucket_bounds_heavy_kernel(const u64 *__restrict__ O2, const u64 *__restrict__ moff,
                                                                  const uint32_t *__restrict__ nch, PartGeom g,
                                                                  const uint8_t *__restrict__ heavy, u64 base,
                                                                  u64 *__restrict__ bstart, u64 *__restrict__ bend) {
    const u64 nregions = part_regions(g);
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r >= nregions) return;
    const uint32_t p1 = part_div_b2(g, r), p2 = (uint32_t)r - p1 * g.b2;
    if (!heavy[p1]) return;
    const u64 n = nch[p1];  // (> 0: a heavy partition has chunks)
    // the partition's [bucket][block] matrix is contiguous in the scan: bucket p2 + 1 starts where bucket p2 ends, and
    // the entry behind the last bucket is the next heavy partition's first (or the scan's total)
    bstart[r] = base + O2[moff[p1] + (u64)p2 * n];
    bend[r] = base + O2[moff[p1] + (u64)(p2 + 1) * n];
}

// ---------------------------------------------------------------------------------------------
// region rebuild: one workgroup per table region, table image in LDS, no global atomics
// ---------------------------------------------------------------------------------------------
// FRESH: the table is known to be empty (skip the 64 KiB read).  A region that overflows is left
// untouched in HBM and flagged; the host re-inserts its bucket after growing the table.
// No global atomics here: a million workgroups adding to one counter word serialise at the memory
// side (measured: ~6 ns per same-address atomic, >100 ms per pass).  Per-region results go to
// rnew[]/rfail[] and region_reduce_kernel folds them afterwards.
//
// u64 payloads: LDS image is structure-of-arrays {payload[], count[]}, 64 KiB, so that 8-byte
// probes spread over all 64 banks.  Round 6: the payload is the hash below the level-1 digit (part_common.hip.h Pay<u64>), the
// image holds payloads, a slot's place comes from the payload's top word (no hash per occurrence: rounds 1-5 hashed every key
// a third time here), and the table's 16-byte slots get their KEYS at the write-back -- one inverse hash per slot the batch
// claimed; a pass over a filled table turns the old keys into payloads when it loads them.
// `dirty`: kh_reset no longer clears the table (5.5 ms for 34 GB); the first FRESH pass after it
// overwrites every region instead, so the regions it would otherwise skip (empty bucket, overflow)
// must be written as empty images.
__device__ __forceinline__ void write_empty_region(Slot *reg, int tid, int nt = REGION_NT) {
    uint4 *o4 = reinterpret_cast<uint4 *>(reg);
    for (uint32_t i = tid; i < REGION_SLOTS; i += nt) o4[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
}

// NT: as for region_count_kernel32 -- a FRESH pass keeps 32-bit count deltas (48 KiB of LDS with the keys: three 512-lane
// workgroups per CU); a pass over a filled table has the old 64-bit counts in LDS too (64 KiB, two workgroups per CU).
template <bool FRESH, int NT = REGION_NT>
__global__ __launch_bounds__(NT, NT == 512 ? 6 : 8) void region_count_kernel64(TableGeom tg, PartGeom g, const u64 *__restrict__ keys, const u64 *__restrict__ bend,
                                                                   const u64 *__restrict__ bstart, uint8_t *__restrict__ rfail,
                                                                   uint32_t *__restrict__ rnew, u64 hot_threshold, uint32_t dirty,
                                                                   u64 *__restrict__ rreal, u64 skip_threshold) {
    __shared__ __attribute__((aligned(16))) u64 s_key[REGION_SLOTS];
    __shared__ u64 s_cnt[FRESH ? 1 : REGION_SLOTS];       // counts (old + new) of a pass over a filled table
    __shared__ uint32_t s_add[FRESH ? REGION_SLOTS : 1];  // count deltas of a fresh pass
    __shared__ uint32_t s_fail;
    __shared__ uint32_t s_new;
    __shared__ u64 s_real;
    const int tid = threadIdx.x;
    const u64 r = blockIdx.x;
    const u64 lo = bstart[r], hi = bend[r];
    Slot *reg = tg.table + r * REGION_SLOTS;
    if (lo == hi || hi - lo > skip_threshold) {  // nothing new for this region, or a bucket left to hot_buckets_kernel
        if (FRESH && dirty) write_empty_region(reg, tid, NT);
        if (tid == 0) {
            rnew[r] = 0;
            rreal[r] = 0;
        }
        return;
    }
    if (FRESH && hi - lo >= 0xFFFFFFFFull) {  // a 32-bit delta could wrap (only with hot buckets switched off): the direct path takes it
        if (dirty) write_empty_region(reg, tid, NT);
        if (tid == 0) {
            rfail[r] = 1;
            rnew[r] = 0;
        }
        return;
    }
    // branch-free loads: bucket-relative index clamped to the last valid key, validity folded into the
    // EMPTY marker (buckets of >= 2^32 keys take the 64-bit index path below through `n` saturation)
    const u64 *__restrict__ src = keys + lo;
    const u64 n = hi - lo;
    const bool hot = n > hot_threshold;  // far above the mean bucket: skewed keys likely (see region32_probe_round)
    u64 kbuf[REGION_RK];  // first round of keys: in flight while the region image is loaded
    u64 nreal = 0;  // keys of the bucket that are k-mers (KH_EMPTY_KEY pads its segments to whole units)
#pragma unroll
    for (int j = 0; j < REGION_RK; ++j) {
        const u64 i = (u64)j * NT + tid;
        const u64 v = src[i < n ? i : n - 1];
        kbuf[j] = i < n ? v : KH_EMPTY_KEY;
        nreal += kbuf[j] != KH_EMPTY_KEY;
    }
    const uint4 *g4 = reinterpret_cast<const uint4 *>(reg);
    const uint32_t p1 = part_div_b2(g, r);
#pragma unroll
    for (uint32_t i = tid; i < REGION_SLOTS; i += NT) {
        if (FRESH) {
            s_key[i] = KH_EMPTY_KEY;
            s_add[i] = 0;
        } else {
            const uint4 v = g4[i];
            const u64 ok_ = ((u64)v.y << 32) | v.x;
            s_key[i] = ok_ == KH_EMPTY_KEY ? (u64)KH_EMPTY_KEY : Pay<u64>::make(ok_, table_hash(tg, ok_), g);  // an old key, as the payload it is probed for
            s_cnt[i] = ((u64)v.w << 32) | v.z;
        }
    }
    if (tid == 0) {
        s_fail = 0;
        s_new = 0;
        s_real = 0;
    }
    __syncthreads();
    uint32_t nd = 0;
    for (u64 base = 0; base < n; base += (u64)REGION_RK * NT) {
        u64 nbuf[REGION_RK];
#pragma unroll
        for (int j = 0; j < REGION_RK; ++j) {  // next round's keys in flight while this round is inserted
            const u64 i = base + (u64)(REGION_RK + j) * NT + tid;
            const u64 v = src[i < n ? i : n - 1];
            nbuf[j] = i < n ? v : KH_EMPTY_KEY;
            nreal += nbuf[j] != KH_EMPTY_KEY;
        }
#pragma unroll
        for (int j = 0; j < REGION_RK; ++j) {
            u64 key = kbuf[j];
            // skew guard for hot buckets (all lanes): lanes holding the first valid lane's key hand it their increment
            u64 weight = 1;
            const u64 vmask = hot ? kh_ballot(key != KH_EMPTY_KEY) : 0ull;
            if (vmask) {
                const int first = __builtin_ctzll(vmask);
                const u64 lead = __shfl(key, first, 64);
                const bool same = key == lead;  // lead is a real key, so EMPTY lanes never match
                const u64 smask = kh_ballot(same);
                if (__builtin_popcountll(smask) > 1) {
                    if ((int)lane_id() == first) weight = (u64)__builtin_popcountll(smask);
                    else if (same) key = KH_EMPTY_KEY;
                }
            }
            if (key == KH_EMPTY_KEY) continue;
            // a probe sequence starts at a multiple of REGION_GROUP slots (start_of): the group's keys come with one LDS read;
            // a value that is not EMPTY is final, a stale EMPTY is corrected by what the compare-and-swap returns
            uint32_t off = kh_start_of_x((uint32_t)(key >> 32), g.b2);  // (`key` is the payload: its top word is x)
            uint32_t probes = 0;
            bool placed = false;
            for (; probes < REGION_SLOTS && !placed; probes += REGION_GROUP) {
                u64 grp[REGION_GROUP];
                if constexpr (REGION_GROUP == 2) {
                    const uint4 x = *reinterpret_cast<const uint4 *>(&s_key[off]);  // ds_read_b128
                    grp[0] = ((u64)x.y << 32) | x.x;
                    grp[1] = ((u64)x.w << 32) | x.z;
                } else {
#pragma unroll
                    for (uint32_t i = 0; i < REGION_GROUP; ++i) grp[i] = s_key[off + i];
                }
#pragma unroll
                for (uint32_t i = 0; i < REGION_GROUP; ++i) {
                    if (placed) continue;
                    u64 cur = grp[i];
                    if (cur == KH_EMPTY_KEY) {
                        cur = atomicCAS(&s_key[off + i], (u64)KH_EMPTY_KEY, key);  // ds_cmpst_rtn_b64
                        if (cur == KH_EMPTY_KEY) {
                            ++nd;
                            cur = key;
                        }
                    }
                    if (cur == key) {
                        if (FRESH) atomicAdd(&s_add[off + i], (uint32_t)weight);  // ds_add_u32
                        else atomicAdd(&s_cnt[off + i], weight);                   // ds_add_u64
                        placed = true;
                    }
                }
                off = (off + REGION_GROUP) & REGION_MASK;
            }
            if (!placed) s_fail = 1;
        }
#pragma unroll
        for (int j = 0; j < REGION_RK; ++j) kbuf[j] = nbuf[j];
    }
    const uint32_t dw = (uint32_t)wave_sum((u64)nd);
    if ((tid & 63) == 0 && dw) atomicAdd(&s_new, dw);  // LDS
    const u64 rw = wave_sum(nreal);
    if ((tid & 63) == 0 && rw) atomicAdd(&s_real, rw);  // LDS, 64-bit: a hot bucket may hold >= 2^32 keys
    __syncthreads();
    if (s_fail) {
        if (FRESH && dirty) write_empty_region(reg, tid, NT);
        if (tid == 0) {
            rfail[r] = 1;
            rnew[r] = 0;
        }
        return;
    }
    uint4 *o4 = reinterpret_cast<uint4 *>(reg);
#pragma unroll
    for (uint32_t i = tid; i < REGION_SLOTS; i += NT) {
        const u64 pv = s_key[i], cc = FRESH ? (u64)s_add[i] : s_cnt[i];
        const u64 kk = pv == KH_EMPTY_KEY ? (u64)KH_EMPTY_KEY : Pay<u64>::key(pv, p1, g);  // the slot's key back from its payload
        o4[i] = make_uint4((uint32_t)kk, (uint32_t)(kk >> 32), (uint32_t)cc, (uint32_t)(cc >> 32));
    }
    if (tid == 0) {
        rnew[r] = s_new;
        rreal[r] = s_real;
    }
}

constexpr uint32_t R32_FREE = 0xFFFFFFFFu;  // free marker of the 32-bit LDS image

// Where a payload of region (p1, digit) starts probing, and whether a payload of the bucket's data IS one of the region's
// (sentinels -- payloads of another bucket -- pad the exact level 2's segments).  POW2: b2 = 2^p2_bits, digit and start are
// bit fields of the payload (rounds 1-3's instructions); else the two words of payload * b2 (kernels.hip.h TableGeom).
template <bool POW2>
struct R32Geo {
    uint32_t b2, digit, sshift, dshift, dmask;
    __device__ __forceinline__ uint32_t start(uint32_t pay) const {
        if constexpr (POW2) return (pay >> sshift) & REGION_START_MASK;
        else return kh_start_of_x(pay, b2);
    }
    // the same as a BYTE offset into a 4-byte-per-slot LDS array (the straight-line first probe addresses LDS by bytes: the
    // shift folds into the field extraction)
    __device__ __forceinline__ uint32_t start_b(uint32_t pay) const {
        if constexpr (POW2) return (pay >> (sshift - 2u)) & (REGION_START_MASK << 2);  // (sshift >= 10: p2_bits <= 10)
        else return kh_start_of_x(pay, b2) << 2;
    }
    __device__ __forceinline__ bool mine(uint32_t pay) const {
        if constexpr (POW2) return (((pay >> dshift) ^ digit) & dmask) == 0;
        else return kh_bucket_of_x(pay, b2) == digit;
    }
};
template <bool POW2>
__device__ __forceinline__ R32Geo<POW2> r32_geo(const PartGeom &g, uint32_t digit) {
    R32Geo<POW2> q;
    q.b2 = g.b2;
    q.digit = digit;
    q.sshift = POW2 ? 32 - g.p2_bits - REGION_BITS : 0;
    q.dshift = (POW2 && g.p2_bits) ? 32 - g.p2_bits : 0;
    q.dmask = (POW2 && g.p2_bits) ? 0xFFFFFFFFu : 0u;  // (no level-2 digit: nothing to check, no sentinels exist)
    return q;
}

// One round of lane-decoupled probing over the lanes' private payload queues (see
// region_count_kernel32).  GUARD = skew guard for buckets far above the mean size: a bucket
// dominated by one key (poly-A, satellites) would send every lane's increment to one LDS word, so
// lanes whose current payload equals the first active lane's hand their weight to that lane and
// move on.  It costs a shuffle and two ballots per iteration, hence only for hot buckets.
// The lanes' payload queues in LDS: one block of (REGION_RK + 1) rows x 64 lanes per WAVE, row-major -- lane l's private
// queue is column l of its wave's block (conflict-free), and the n-th item a wave queues together (first probe of
// region_count_kernel32) is word n of the block.  Row REGION_RK is a dummy row for predicated stores.
constexpr uint32_t R32_QBLOCK = (REGION_RK + 1) * 64;
__device__ __forceinline__ uint32_t r32_qbase(int tid) { return ((uint32_t)tid >> 6) * R32_QBLOCK + ((uint32_t)tid & 63u); }

template <bool GUARD, bool POW2>
__device__ __forceinline__ void region32_probe_round(uint32_t nk, const uint32_t *s_q, uint32_t *s_pay,
                                                     uint32_t *s_add, uint32_t *s_special, uint32_t *s_fail, int tid,
                                                     const R32Geo<POW2> &rg, uint32_t &nd) {
    uint32_t idx = 0, pay = 0, off = 0, probes = 0, weight = 1;
    const uint32_t qb = r32_qbase(tid);
    bool active = nk > 0;
    if (active) {
        pay = s_q[qb];
        off = rg.start(pay);
    }
    for (;;) {
        const u64 amask = kh_ballot(active);
        if (amask == 0) break;
        bool placed = false;
        if (GUARD) {  // all lanes take part
            const int first = __builtin_ctzll(amask);
            const uint32_t lead = (uint32_t)__shfl((int)pay, first, 64);
            const bool same = active && pay == lead;
            const u64 smask = kh_ballot(same);
            if (__builtin_popcountll(smask) > 1) {
                uint32_t wsum = same ? weight : 0u;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) wsum += (uint32_t)__shfl_xor((int)wsum, o, 64);
                if ((int)lane_id() == first) weight = wsum;
                else if (same) placed = true;  // absorbed: nothing left to insert for this item
            }
        }
        if (active && !placed) {
            if (pay == R32_FREE) {  // collides with the free marker: counted, placed at write-back
                atomicAdd(s_special, weight);
                placed = true;
            } else {
                uint32_t cur = s_pay[off];
                if (cur == R32_FREE) {
                    cur = atomicCAS(&s_pay[off], R32_FREE, pay);
                    if (cur == R32_FREE) {
                        ++nd;
                        cur = pay;
                    }
                }
                if (cur == pay) {
                    atomicAdd(&s_add[off], weight);  // no-return ds_add_u32
                    placed = true;
                } else {
                    off = (off + 1) & REGION_MASK;
                    if (++probes >= REGION_SLOTS) {  // region full
                        *s_fail = 1;
                        placed = true;
                    }
                }
            }
        }
        if (active && placed) {
            ++idx;
            active = idx < nk;
            weight = 1;
            if (active) {
                pay = s_q[qb + idx * 64];
                off = rg.start(pay);
                probes = 0;
            }
        }
    }
}

// The same round for regions in which no payload can equal the free marker (the top p2_bits of every
// payload are the region's own level-2 digit: unless that digit is all ones, 0xFFFFFFFF cannot occur)
// and that are not hot -- i.e. nearly all of them: predicated straight-line code instead of nested
// branches (the general round spends more scalar than vector instructions on exec-mask bookkeeping),
// and a GROUP of REGION_GROUP = 2 slots per step: a probe sequence starts at an even slot (REGION_START_MASK), so one
// 8-byte LDS read shows the two slots a key most likely sits in, or its first free one.
// the REGION_GROUP slots from s_pay[grp] on (grp a multiple of REGION_GROUP): one LDS read
struct R32Group {
    uint32_t v[REGION_GROUP];
};
__device__ __forceinline__ R32Group r32_group_load(const uint32_t *s_pay, uint32_t grp) {
    R32Group c;
    if constexpr (REGION_GROUP == 4) {
        const uint4 x = *reinterpret_cast<const uint4 *>(&s_pay[grp]);
        c.v[0] = x.x; c.v[1] = x.y; c.v[2] = x.z; c.v[3] = x.w;
    } else if constexpr (REGION_GROUP == 2) {
        const uint2 x = *reinterpret_cast<const uint2 *>(&s_pay[grp]);
        c.v[0] = x.x; c.v[1] = x.y;
    } else {
        c.v[0] = s_pay[grp];
    }
    return c;
}
// (byte-addressed twins for the straight-line first probe: gb = 4 x the group's first slot)
__device__ __forceinline__ R32Group r32_group_load_b(const uint32_t *s_pay, uint32_t gb) {
    R32Group c;
    const char *const b = reinterpret_cast<const char *>(s_pay) + gb;
    if constexpr (REGION_GROUP == 4) {
        const uint4 x = *reinterpret_cast<const uint4 *>(b);
        c.v[0] = x.x; c.v[1] = x.y; c.v[2] = x.z; c.v[3] = x.w;
    } else if constexpr (REGION_GROUP == 2) {
        const uint2 x = *reinterpret_cast<const uint2 *>(b);
        c.v[0] = x.x; c.v[1] = x.y;
    } else {
        c.v[0] = *reinterpret_cast<const uint32_t *>(b);
    }
    return c;
}
__device__ __forceinline__ bool r32_group_find_b(const R32Group &c, uint32_t pay, uint32_t &o4) {
    bool any = c.v[0] == pay;
    o4 = 0;
#pragma unroll
    for (uint32_t i = 1; i < REGION_GROUP; ++i) {
        const bool m = c.v[i] == pay;
        any = any || m;
        o4 = m ? 4u * i : o4;
    }
    return any;
}
// is `pay` in the group, and where (o: its slot's offset inside the group; a payload sits in at most one slot)
__device__ __forceinline__ bool r32_group_find(const R32Group &c, uint32_t pay, uint32_t &o) {
    bool any = c.v[0] == pay;
    o = 0;
#pragma unroll
    for (uint32_t i = 1; i < REGION_GROUP; ++i) {
        const bool m = c.v[i] == pay;
        any = any || m;
        o = m ? i : o;
    }
    return any;
}
// the first free slot of the group (false: none)
__device__ __forceinline__ bool r32_group_free(const R32Group &c, uint32_t &fo) {
    bool any = false;
    fo = 0;
#pragma unroll
    for (int i = (int)REGION_GROUP - 1; i >= 0; --i) {
        const bool f = c.v[i] == R32_FREE;
        any = any || f;
        fo = f ? (uint32_t)i : fo;
    }
    return any;
}

// the first free slot of the group, as a byte offset (false: none)
__device__ __forceinline__ bool r32_group_free_b(const R32Group &c, uint32_t &f4) {
    bool any = false;
    f4 = 0;
#pragma unroll
    for (int i = (int)REGION_GROUP - 1; i >= 0; --i) {
        const bool f = c.v[i] == R32_FREE;
        any = any || f;
        f4 = f ? 4u * (uint32_t)i : f4;
    }
    return any;
}

// The probing loop.  Round 4, from its ISA (27 vector + ~30 scalar instructions per iteration, 45 % of the kernel's vector
// instructions): everything is a BYTE offset (group, slot, the lane's place in its queue column: no shifts), "is the lane
// still busy" is a compare of its own every iteration (a loop-carried predicate is rebuilt as 0 / 1 and compared again for
// the ballot, and merged by four scalar instructions at every back edge), a claim is a plain divergent branch (the
// "does anybody claim" ballot in front of it cost two vector instructions per iteration to skip a block the exec mask
// skips by itself), and "the region is full" is the probe sequence coming back to its first group (no probe counter).
template <bool POW2>
__device__ __forceinline__ void region32_probe_lean(uint32_t nk, const uint32_t *s_q, uint32_t *s_pay,
                                                    uint32_t *s_add, uint32_t *s_fail, int tid, const R32Geo<POW2> &rg, uint32_t &nd) {
    constexpr uint32_t GB = 4u * REGION_GROUP, WRAP = REGION_MASK << 2;
    const char *qp = reinterpret_cast<const char *>(s_q + r32_qbase(tid));  // the lane's column: row i is 256 i bytes on (rows 0 .. REGION_RK exist)
    char *const payb = reinterpret_cast<char *>(s_pay);
    char *const addb = reinterpret_cast<char *>(s_add);
    uint32_t idx = 0;
    uint32_t pay = *reinterpret_cast<const uint32_t *>(qp);  // (every queue slot holds a loaded payload, real or clamped)
    uint32_t gb = rg.start_b(pay), gb0 = gb;                  // the group being looked at; the item's first one
    for (;;) {
        const bool active = idx < nk;
        if (kh_ballot(active) == 0) break;
        const R32Group c = r32_group_load_b(s_pay, gb);
        uint32_t o4, f4;
        bool hit = r32_group_find_b(c, pay, o4) && active;
        const bool claim = r32_group_free_b(c, f4) && active && !hit;
        bool again = false;
        if (claim) {  // (rare once the region's keys are in)
            const uint32_t old = atomicCAS(reinterpret_cast<uint32_t *>(payb + (gb | f4)), R32_FREE, pay);
            if (old == R32_FREE) ++nd;
            hit = old == R32_FREE || old == pay;
            o4 = f4;
            again = !hit;  // another key took that slot meanwhile: look at the group again
        }
        if (hit) atomicAdd(reinterpret_cast<uint32_t *>(addb + (gb | o4)), 1u);  // no-return ds_add_u32
        const bool miss = active && !hit && !again;  // the group's slots hold other keys
        const uint32_t nb = (gb + GB) & WRAP;
        bool done = hit;
        if (miss && nb == gb0) {  // every group seen: region full
            *s_fail = 1;
            done = true;
        }
        gb = miss ? nb : gb;
        if (done) {
            ++idx;
            qp += 256;
            pay = *reinterpret_cast<const uint32_t *>(qp);
            gb = gb0 = rg.start_b(pay);
        }
    }
}

// The same loop over a queue shared by the WAVE (round 6): `total` items, item n at word n of the wave's block of s_q.  A lane
// that has placed its item PULLS the next unclaimed one (ballot + mbcnt over the lanes that finished in this iteration, a
// wave-uniform cursor) instead of walking a share of its own: the loop ran until the wave's unluckiest lane -- the one whose
// items had the longest probe sequences -- was done; now it runs for the wave's total number of probes / 64, rounded up.
// (next4 <= 256 + 4 total <= 4 R32_QBLOCK: a lane that pulls beyond `total` reads the dummy row and stays idle.)
// Measured (round 6, same box, twice): for the rounds that go through the loop whole -- the first round of a region that has no
// full round, i.e. the hg-shaped input's million regions of 2.9 K payloads, nine in ten of them new keys -- 44.6 / 44.4 -> 44.0 /
// 44.0 ms per step; behind the straight-line first probe, where a lane has two or three items left, the pulling costs what the
// balance gains (headline 16.2 -> 16.4 ms, configs[3]'s share 19.9 -> 20.0): static shares stay there.
#ifndef KH_REGION_PULL
#define KH_REGION_PULL 1  // (0: A/B builds -- the lanes' own queues and region32_probe_lean for whole rounds)
#endif
#ifndef KH_REGION_PULL_FP
#define KH_REGION_PULL_FP 0  // (1: A/B builds -- the pulling loop behind the straight-line first probe as well)
#endif
#ifndef KH_REGION_CLAIM_PARTIAL
#define KH_REGION_CLAIM_PARTIAL 0  // (1: A/B builds -- the straight-line claims for a first round that is not full: two in flight, empty rows skipped)
#endif
template <bool POW2>
__device__ __forceinline__ void region32_probe_pull(uint32_t total, const uint32_t *s_q, uint32_t *s_pay, uint32_t *s_add,
                                                    uint32_t *s_fail, int tid, const R32Geo<POW2> &rg, uint32_t &nd) {
    constexpr uint32_t GB = 4u * REGION_GROUP, WRAP = REGION_MASK << 2;
    const char *const qw = reinterpret_cast<const char *>(s_q + ((uint32_t)tid >> 6) * R32_QBLOCK);  // the wave's block
    char *const payb = reinterpret_cast<char *>(s_pay);
    char *const addb = reinterpret_cast<char *>(s_add);
    const uint32_t total4 = 4u * total;
    uint32_t i4 = 4u * ((uint32_t)tid & 63u);  // the lane's item (byte offset into the block)
    uint32_t next4 = 256u;                      // the first unclaimed item (wave-uniform)
    uint32_t pay = *reinterpret_cast<const uint32_t *>(qw + i4);
    uint32_t gb = rg.start_b(pay), gb0 = gb;
    for (;;) {
        const bool active = i4 < total4;
        if (kh_ballot(active) == 0) break;
        const R32Group c = r32_group_load_b(s_pay, gb);
        uint32_t o4, f4;
        bool hit = r32_group_find_b(c, pay, o4) && active;
        const bool claim = r32_group_free_b(c, f4) && active && !hit;
        bool again = false;
        if (claim) {
            const uint32_t old = atomicCAS(reinterpret_cast<uint32_t *>(payb + (gb | f4)), R32_FREE, pay);
            if (old == R32_FREE) ++nd;
            hit = old == R32_FREE || old == pay;
            o4 = f4;
            again = !hit;
        }
        if (hit) atomicAdd(reinterpret_cast<uint32_t *>(addb + (gb | o4)), 1u);
        const bool miss = active && !hit && !again;
        const uint32_t nb = (gb + GB) & WRAP;
        bool done = hit;
        if (miss && nb == gb0) {  // every group seen: region full
            *s_fail = 1;
            done = true;
        }
        gb = miss ? nb : gb;
        const u64 dm = kh_ballot(done);
        if (done) {
            i4 = next4 + 4u * __builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u));
            pay = *reinterpret_cast<const uint32_t *>(qw + i4);
            gb = gb0 = rg.start_b(pay);
        }
        next4 += 4u * (uint32_t)__builtin_popcountll(dm);
    }
}

// uint32_t payloads: the LDS image is two 32-bit arrays, s_pay[] (0xFFFFFFFF = free) and s_add[] (count
// added by this batch), 32 KiB per region, plus the lanes' payload queues -- nine words per lane: 36 KiB at 1024 lanes,
// 18 at 512 (see the probing loop); the count update is a no-return ds_add_u32.  Slots that were
// already occupied keep their 64-bit key and count in the registers of the lane that owns them; at
// write-back the batch's delta is added, and new slots get their key back through the inverse hash.
//   * a bucket of >= 2^32 payloads could wrap a 32-bit delta: such a region is failed up front;
//   * the one payload that collides with the free marker (0xFFFFFFFF) is counted in s_special and
//     inserted by a single lane at write-back.
// A failed region is left untouched in HBM; its bucket then goes through the direct path.

// NARROW (round 3): the table image this pass reads and writes is the 8-byte form -- one u64 per slot, count << 32 | the
// slot's 32-bit payload (the very word the partition buffers carry), 0 = free: `ntab`, laid out like the table (region r
// = slots [4096 r, 4096 (r + 1))).  Half the bytes of the 16-byte {key, count} slots on the way out (and, on a pass over
// a filled table, on the way in), no inverse hash at the write-back, half the registers for the old slots.  The key of a
// narrow slot is Pay<uint32_t>::key(payload, region >> p2_bits, g): ntable_widen_kernel turns the image into the 16-byte
// table when something needs that (kmerhip.hip ensure_wide); counting, histogram, compaction and lookups read it as it is.
// A count that does not fit 32 bits fails the region (rfail = 2): the host widens the table and re-inserts the bucket.
// NT = lanes of the workgroup: 1024 (two workgroups per CU: 72 KiB of LDS each) or 512 (52 KiB: THREE per CU, a third fewer
// waves).  Measured (profiles/README.md r03b): at load 0.5 -- 24 K payloads per bucket, the probing loop a third of the kernel --
// three workgroups in different phases beat two: 21.1 -> 19.2 ms (S50M at load 0.58: 12.2 -> 11.2; S10M at 0.44: 4.1 -> 3.3);
// at load 0.61 with 30 K payloads per bucket the loop wants the waves: 36.4 -> 40.9 ms (but at 0.62 with 2.9 K per bucket:
// 27.4 -> 20.5).  kmerhip.hip (region_small_groups) chooses.
template <bool FRESH, bool NARROW, int NT = REGION_NT, bool POW2 = false>
__global__ __launch_bounds__(NT, NT == 512 ? 6 : 8) void region_count_kernel32(TableGeom tg, PartGeom g, const uint32_t *__restrict__ pays, const u64 *__restrict__ bend,
                                                                   const u64 *__restrict__ bstart, uint8_t *__restrict__ rfail,
                                                                   uint32_t *__restrict__ rnew, u64 hot_threshold, uint32_t dirty,
                                                                   uint32_t head_cb, uint32_t *__restrict__ rheads, Counters *ctr,
                                                                   u64 *__restrict__ rreal, u64 *__restrict__ ntab, u64 skip_threshold,
                                                                   const uint8_t *__restrict__ arena_heavy) {
    // arena_heavy: nullptr, or -- when the batch's level 2 was the arena kernel -- its per-partition "left to the exact
    // kernels" flags: a partition that is not flagged has buckets without sentinels (see first_probe below).
    // rreal[r] = payloads of the bucket that are k-mers (the bucket may hold SENTINELS, payloads with another
    // level-2 digit that pad its segments to whole lines: part2_scatter_lines_kernel; they are skipped here)
    // head_cb != 0 (FRESH only): also leave in rheads[r] the number of 32-bit exchange heads the region
    // will need (shard.hip.h heads_of), so that a multi-GPU export right after this pass can skip its
    // counting pass over the table.
    __shared__ __attribute__((aligned(16))) uint32_t s_pay[REGION_SLOTS];
    __shared__ uint32_t s_add[REGION_SLOTS + NT];  // (+ one dummy word per lane for predicated adds)
    __shared__ uint32_t s_fail;
    __shared__ uint32_t s_new;
    __shared__ uint32_t s_special, s_sp_off, s_sp_new, s_heads, s_real;
    __shared__ uint32_t s_q[(REGION_RK + 1) * NT];  // the lanes' payload queues, one block per wave (r32_qbase)
    constexpr int SPL = REGION_SLOTS / NT;  // slots per lane
    const int tid = threadIdx.x;
    const u64 r = blockIdx.x;
    const u64 lo = bstart[r], hi = bend[r];
    auto write_empty = [&]() {
        if (NARROW) {
            uint4 *o4 = reinterpret_cast<uint4 *>(ntab + r * REGION_SLOTS);
            for (uint32_t i = tid; i < REGION_SLOTS / 2; i += NT) o4[i] = make_uint4(0u, 0u, 0u, 0u);
        } else {
            write_empty_region(tg.table + r * REGION_SLOTS, tid, NT);
        }
    };
    if (lo == hi || hi - lo > skip_threshold) {  // (a bucket above skip_threshold is left to hot_buckets_kernel)
        if (FRESH && dirty) write_empty();
        if (tid == 0) {
            rnew[r] = 0;
            rreal[r] = 0;
            if (head_cb) rheads[r] = 0;
        }
        return;
    }
    if (hi - lo >= 0xFFFFFFFFull) {  // a 32-bit delta could wrap
        if (FRESH && dirty) write_empty();
        if (tid == 0) {
            rfail[r] = 1;
            rnew[r] = 0;
        }
        return;
    }
    const uint32_t p1 = part_div_b2(g, r);
    // H = [p1 | payload << (32 - p1_bits) ...]: bucket and in-region start are the high and the low word of
    // payload * b2 (kh_bucket_of_x / kh_start_of_x; for b2 = 2^j the top j bits of the payload and the REGION_BITS behind them)
    const uint32_t b2 = g.b2;
    Slot *reg = tg.table + r * REGION_SLOTS;
    const uint32_t *__restrict__ src = pays + lo;
    const uint32_t n = (uint32_t)(hi - lo);  // < 2^32 - 1 (checked above)
    const bool hot = (hi - lo) > hot_threshold;  // far above the mean bucket: skewed keys likely
    // can a payload of this region equal the free marker 0xFFFFFFFF?  Only in a partition's last bucket.
    const uint32_t digit = (uint32_t)r - p1 * b2;             // the region's bucket: a real payload x has (x * b2) >> 32 == digit
    const R32Geo<POW2> rg = r32_geo<POW2>(g, digit);
    // (0xFFFFFFFF falls into the LAST bucket; b2 == 1: the only one -- and only where all 32 bits of a payload are hash bits: in a
    //  batch whose level 2 narrowed its payloads, g = (log2 regions, 1), a k = 25 payload has 31 and its low bit is zero)
    const bool may_special = digit == b2 - 1u && 2 * (int)g.k - (int)g.shard_shift - (int)g.p1_bits >= 32;
    const bool no_sentinels = arena_heavy != nullptr && arena_heavy[p1] == 0;  // (uniform)
    uint32_t nreal = 0;
    uint32_t kbuf[REGION_RK];
#pragma unroll
    for (int j = 0; j < REGION_RK; ++j) {  // branch-free: clamped index
        const uint32_t i = (uint32_t)j * NT + tid;
        kbuf[j] = src[i < n ? i : n - 1];
    }
    Slot old[NARROW ? 1 : SPL];
    u64 oldn[NARROW ? SPL : 1];  // the narrow form of the lane's old slots
    const u64 *const nreg = NARROW ? ntab + r * REGION_SLOTS : nullptr;
    bool unrepresentable = false;
    const uint4 *g4 = reinterpret_cast<const uint4 *>(reg);
#pragma unroll
    for (int q = 0; q < SPL; ++q) {
        const uint32_t i = (uint32_t)q * NT + tid;
        uint32_t w = R32_FREE;
        if (NARROW) {
            oldn[q] = FRESH ? 0ull : nreg[i];
            if (oldn[q] >> 32) {  // live: the payload is stored as it is probed for
                w = (uint32_t)oldn[q];
                if (w == R32_FREE) {  // (see below: the slot must not look free)
                    if (b2 > 1) w = 0u;
                    else unrepresentable = true;
                }
            }
        } else if (FRESH) {
            old[q].key = KH_EMPTY_KEY;
            old[q].count = 0;
        } else {
            const uint4 v = g4[i];
            old[q].key = ((u64)v.y << 32) | v.x;
            old[q].count = ((u64)v.w << 32) | v.z;
            if (old[q].key != KH_EMPTY_KEY) {
                w = Pay<uint32_t>::make(old[q].key, table_hash(tg, old[q].key), g);
                if (w == R32_FREE) {
                    // An old key whose payload equals the free marker: new arrivals of that payload are counted
                    // in s_special and merged below, but its SLOT must not look free to the other payloads (a
                    // claim there would add the newcomer's count to this key and lose the newcomer).  Every
                    // payload of this region carries the region's level-2 digit -- all ones here -- in its top
                    // bits, so 0 can neither arrive nor be probed for: it marks the slot as taken.  Without a
                    // level-2 digit (tables of <= 1024 regions) there is no such value: fail the region, its
                    // bucket then goes through the direct path.
                    if (b2 > 1) w = 0u;
                    else unrepresentable = true;
                }
            }
        }
        s_pay[i] = w;
        s_add[i] = 0;
    }
    if (tid == 0) {
        s_fail = 0;
        s_new = 0;
        s_special = 0;
        s_heads = 0;
        s_real = 0;
    }
    __syncthreads();
    if (unrepresentable) s_fail = 1;
    uint32_t nd = 0;
    // Lane-decoupled probing.  A wave that walks key j of all 64 lanes together pays, for every
    // key, the LONGEST probe sequence among its lanes (~6 at load 0.5).  Here each lane keeps its own
    // cursor into a private queue of REGION_RK payloads (LDS, a column of its wave's block: r32_qbase, conflict-free) and takes
    // its next payload as soon as its current one is placed, so a round costs the largest SUM of
    // probe lengths of one lane (~1.5 per key) instead of the sum of the per-key maxima.
    for (u64 base = 0; base < n; base += (u64)REGION_RK * NT) {
        // payloads of this round -> the lane's queue; how many of them are real
        const u64 rem = n - base;  // > 0
        uint32_t nk = 0;
        if ((u64)tid < rem) nk = (uint32_t)((rem - tid + NT - 1) / NT);
        if (nk > REGION_RK) nk = REGION_RK;
        uint32_t pj[REGION_RK];
#pragma unroll
        for (int j = 0; j < REGION_RK; ++j) pj[j] = kbuf[j];
        if (rem > (u64)REGION_RK * NT) {  // (uniform; a region of one round -- most regions of a sparse table -- loads nothing here)
#pragma unroll
            for (int j = 0; j < REGION_RK; ++j) {  // next round's payloads in flight during the probing
                const u64 i64 = base + (u64)(REGION_RK + j) * NT + tid;
                kbuf[j] = src[i64 < n ? (uint32_t)i64 : n - 1];
            }
        }
#ifndef KH_REGION_R1_LOOP
#define KH_REGION_R1_LOOP 1  // (0: A/B builds -- the first round takes the straight-line first probe like the others)
#endif
        // The FIRST round of a fresh pass finds the region's image empty: a first probe cannot hit there (every payload would
        // go on to the queue and the loop anyway), so the round skips it and goes through the lanes' own queues -- which are
        // balanced there, eight items each.
        const bool first_round = KH_REGION_R1_LOOP && FRESH && base == 0;
        // ... unless the payloads are plain (no hot bucket, no free-marker payload): then the first round CLAIMS in straight-line
        // code -- a compare-and-swap on the first slot of every payload's home group, eight in flight -- and only what finds that
        // slot taken by another key goes through the loop (round 5, below)
#ifndef KH_REGION_R1_CLAIM
#define KH_REGION_R1_CLAIM 1  // (0: A/B builds -- the first round of a fresh pass goes through the lanes' queues and the loop)
#endif
        // (a FULL round only: with 2.9 K payloads in a region's one round -- the hg-shaped input -- three of a lane's eight places
        //  are empty and still swap at a dummy word, and the pass was 1.0 ms slower than through the queues: 16.5 -> 17.5 ms)
        const bool claim_round = KH_REGION_R1_CLAIM && first_round && !hot && !may_special && (KH_REGION_CLAIM_PARTIAL || rem >= (u64)REGION_RK * NT);
        const uint32_t rows = rem >= (u64)REGION_RK * NT ? (uint32_t)REGION_RK : (uint32_t)((rem + NT - 1) / NT);  // (uniform: rows of the round that hold a payload)
        if (hot || may_special || !FRESH || (first_round && !claim_round)) {
            // (the pass over a filled table keeps the old slots in registers; the straight-line first probe below
            // would push it over the 64 registers that two workgroups per CU allow)
            if (KH_REGION_PULL && FRESH && !hot && !may_special) {  // (FRESH only: a pass over a filled table keeps its old slots in registers, and the 1024-lane kernel's 64 would spill)
                // real payloads (sentinels dropped), compacted into the WAVE's queue: item n at word n of its block
                const uint32_t lane = (uint32_t)tid & 63u, wq = ((uint32_t)tid >> 6) * R32_QBLOCK;
                uint32_t wrun = 0;
#pragma unroll
                for (int j = 0; j < REGION_RK; ++j) {
                    const bool valid = (uint32_t)j < nk && rg.mine(pj[j]);
                    nreal += valid;
                    const u64 qm = kh_ballot(valid);
                    const uint32_t pos = wrun + __builtin_amdgcn_mbcnt_hi((uint32_t)(qm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)qm, 0u));
                    s_q[wq + (valid ? pos : (uint32_t)REGION_RK * 64u + lane)] = pj[j];  // (row REGION_RK is a dummy row)
                    wrun += (uint32_t)__builtin_popcountll(qm);
                }
                region32_probe_pull<POW2>(wrun, s_q, s_pay, s_add, &s_fail, tid, rg, nd);
            } else {
            uint32_t rq = 0;  // real payloads, compacted into the lane's queue (sentinels dropped)
#pragma unroll
            for (int j = 0; j < REGION_RK; ++j)
                if ((uint32_t)j < nk && rg.mine(pj[j])) s_q[r32_qbase(tid) + 64 * rq++] = pj[j];
            nreal += rq;
            if (hot) region32_probe_round<true, POW2>(rq, s_q, s_pay, s_add, &s_special, &s_fail, tid, rg, nd);
            else if (may_special) region32_probe_round<false, POW2>(rq, s_q, s_pay, s_add, &s_special, &s_fail, tid, rg, nd);
            else region32_probe_lean<POW2>(rq, s_q, s_pay, s_add, &s_fail, tid, rg, nd);
            }
        } else {
            // First probe of all eight payloads as straight-line code: the eight slot reads are in flight
            // together and a payload that finds its key right there (most of them: a key comes ~12 times, and at
            // load 0.5 two thirds of the keys sit in their home slot) costs a compare and a no-return ds_add --
            // no loop, no ballot.  A slot only ever goes FREE -> payload once, so an equal value is final; a
            // stale FREE or a different payload sends the item to the probing loop, which starts over at the
            // home slot.
            // What the first probe does not settle goes to the probing loop -- through a queue shared by the WAVE, not the
            // lane's own: the loop runs until the wave's busiest lane is done (ablation: the loop is 8.9 of the
            // kernel's 24.9 ms), and with private queues that lane has twice the average work.  The wave's items are
            // numbered by ballot + mbcnt and item n goes to row n / 64, column n % 64 of the wave's 8 x 64 corner of s_q:
            // every lane then takes rows 0 .. of its own column, one item more or less than its neighbours.
#ifndef KH_REGION_FP
#define KH_REGION_FP 4
#endif
            // group reads in flight: FP x REGION_GROUP registers (8 x 4 would not fit two workgroups per CU: 52 bytes of scratch
            // per lane, 29 ms); measured with pairs: 4 in flight 21.2 ms, 8 in flight 21.9 ms
#if KH_ABLR & 8  /* timing experiment: loads, LDS image and write-back only -- no probing at all */
            if (n) continue;
#endif
            using FPfull = std::integral_constant<int, KH_REGION_FP>;
            using FPpart = std::integral_constant<int, KH_REGION_CLAIM_PARTIAL ? 2 : KH_REGION_FP>;
            const uint32_t lane = (uint32_t)tid & 63u, wq = ((uint32_t)tid >> 6) * R32_QBLOCK;
            const uint32_t dummy_b = 4u * (REGION_SLOTS + (uint32_t)tid);
            // CHECK = false: a FULL round of a bucket that holds no sentinels (the arena level 2 writes none) -- every lane has
            // REGION_RK payloads and all of them are the region's: no validity test per payload (3 of the ~18 vector
            // instructions a payload costs here)
            // CLAIM (round 5): the first round of a fresh pass.  The image is empty, a read-only probe cannot hit: the payload is
            // compared-and-swapped into the FIRST slot of its home group instead (FREE -> payload: a new key; already this
            // payload: a copy that came a moment earlier), FP of them in flight.  What finds another key there goes to the loop,
            // which looks at the whole group and on.  Measured with probing taken out (KH_ABLR-style builds, round 5): of the
            // 16.6 ms of an hg-shaped input's pass -- a million regions of 2.9 K payloads, nine in ten of them new keys, ALL of them
            // first-round payloads -- 11 were the loop, 27 vector + 30 scalar instructions per iteration and payload.  (That input
            // keeps the loop all the same -- see claim_round; the full first rounds of denser regions gain: 16.4 -> 16.0 ms at the headline.)
            auto first_probe = [&](auto chk, auto clm, auto fpv) -> uint32_t {
                constexpr bool CHECK = decltype(chk)::value, CLAIM = decltype(clm)::value;
                constexpr int FP = decltype(fpv)::value;
                uint32_t wrun = 0;  // items queued by the wave so far (wave-uniform)
#pragma unroll
                for (int h = 0; h < REGION_RK; h += FP) {
                    if (CHECK && (uint32_t)h >= rows) break;  // (uniform)
                    uint32_t oj[FP];
                    R32Group cj[FP];
                    bool vj[FP];
#pragma unroll
                    for (int j = 0; j < FP; ++j) oj[j] = rg.start_b(pj[h + j]);  // (byte offsets)
#pragma unroll
                    for (int j = 0; j < FP; ++j) {
                        vj[j] = true;
                        if constexpr (CHECK) vj[j] = (uint32_t)(h + j) < nk && rg.mine(pj[h + j]);
                        if constexpr (CLAIM) {  // (a payload that is not the region's swaps at the lane's dummy word)
                            char *const at = vj[j] ? reinterpret_cast<char *>(s_pay) + oj[j] : reinterpret_cast<char *>(s_add) + dummy_b;
                            cj[j].v[0] = atomicCAS(reinterpret_cast<uint32_t *>(at), R32_FREE, pj[h + j]);
                        } else {
                            cj[j] = r32_group_load_b(s_pay, oj[j]);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < FP; ++j) {
                        // (predicated, not branched: the exec-mask bookkeeping of sixteen small branches per round cost
                        //  as many scalar instructions as the kernel has vector ones)
                        const uint32_t pay = pj[h + j];
                        const bool valid = vj[j];
                        if constexpr (CHECK) nreal += valid;
                        uint32_t o4;
                        bool hit;
                        if constexpr (CLAIM) {
                            const uint32_t was = cj[j].v[0];
                            nd += (valid && was == R32_FREE) ? 1u : 0u;
                            hit = valid && (was == R32_FREE || was == pay);
                            o4 = 0;
                        } else {
                            hit = r32_group_find_b(cj[j], pay, o4) && valid;
                        }
#if !(KH_ABLR & 2)  /* timing experiment otherwise: no count updates in the first probe */
                        // no-return ds_add_u32; misses add to a private dummy word
                        atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s_add) + (hit ? (oj[j] | o4) : dummy_b)), 1u);
#endif
                        const bool queue = valid && !hit;
                        const u64 qm = kh_ballot(queue);
                        const uint32_t pos = wrun + __builtin_amdgcn_mbcnt_hi((uint32_t)(qm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)qm, 0u));
#if !(KH_ABLR & 16)  /* timing experiment otherwise (with & 1): no queue store in the first probe */
                        s_q[wq + (queue ? pos : (uint32_t)REGION_RK * 64u + lane)] = pay;  // (row REGION_RK is a dummy row)
#else
                        (void)pos;
#endif
                        wrun += (uint32_t)__builtin_popcountll(qm);
                    }
                }
                if constexpr (!CHECK) nreal += REGION_RK;
                return wrun;
            };
#ifndef KH_REGION_NOCHECK
#define KH_REGION_NOCHECK 1  // (0: A/B builds -- every round tests every payload)
#endif
            const bool plain = KH_REGION_NOCHECK && no_sentinels && rem >= (u64)REGION_RK * NT;  // uniform
            uint32_t wrun;
            if (claim_round) wrun = plain ? first_probe(std::false_type{}, std::true_type{}, FPfull{}) : first_probe(std::true_type{}, std::true_type{}, FPpart{});
            else wrun = plain ? first_probe(std::false_type{}, std::false_type{}, FPfull{}) : first_probe(std::true_type{}, std::false_type{}, FPfull{});
            const uint32_t r = wrun > lane ? (wrun - lane + 63u) >> 6 : 0u;  // this lane's share: rows 0 .. r-1 of its column
#if !(KH_ABLR & 1)  /* timing experiment otherwise: no probing loop behind the straight-line first probe */
            if (KH_REGION_PULL_FP) region32_probe_pull<POW2>(wrun, s_q, s_pay, s_add, &s_fail, tid, rg, nd);
            else region32_probe_lean<POW2>(r, s_q, s_pay, s_add, &s_fail, tid, rg, nd);
#endif
        }
    }
    const uint32_t dw = (uint32_t)wave_sum((u64)nd);
    if ((tid & 63) == 0 && dw) atomicAdd(&s_new, dw);
    const uint32_t rw = (uint32_t)wave_sum((u64)nreal);
    if ((tid & 63) == 0 && rw) atomicAdd(&s_real, rw);
    __syncthreads();
    if (s_special) {  // (uniform: read behind the barrier; almost never -- and then the region pays two more barriers)
    if (tid == 0 && !s_fail) {
        // The payload equal to the free marker was only counted.  One lane places it now by plain
        // linear probing over the combined image (old keys from HBM, new claims from s_pay).
        const u64 key = Pay<uint32_t>::key(R32_FREE, p1, g);
        uint32_t off = rg.start(R32_FREE);
        uint32_t probes = 0;
        bool is_new = false;
        for (; probes < REGION_SLOTS; ++probes, off = (off + 1) & REGION_MASK) {
            bool old_is_key = false, old_is_free = true;  // what the table held in this slot before the batch
            if (!FRESH) {
                if (NARROW) {
                    const u64 ns = nreg[off];
                    old_is_free = (ns >> 32) == 0;
                    old_is_key = !old_is_free && (uint32_t)ns == R32_FREE;  // (in a narrow image the payload IS the key's identity)
                } else {
                    const u64 o = reg[off].key;
                    old_is_free = o == KH_EMPTY_KEY;
                    old_is_key = o == key;
                }
            }
            if (old_is_key) break;
            if (old_is_free && s_pay[off] == R32_FREE) {
                is_new = true;
                break;
            }
        }
        if (probes == REGION_SLOTS) {
            s_fail = 1;
        } else {
            s_sp_off = off;
            s_sp_new = is_new ? 1u : 0u;
            if (is_new) s_new += 1;
        }
    }
    __syncthreads();
    }
    const uint32_t sp_off = s_special ? s_sp_off : 0xFFFFFFFFu;
    // (Round 5, same-box A/B x 3: without those two barriers and the check below in a fresh pass 16.92 -> 16.65 ms at the
    //  headline, 17.62 -> 17.07 for the hg-shaped input's million regions of 2.9 K payloads.  Setting the image up and reading it
    //  back two slots at a time -- 8-byte LDS accesses, 16-byte stores -- on top of that: 16.69 / 17.26, nothing; four at a
    //  time, each store instruction writing every other 16 bytes of its lines: slower than one slot at a time.)
    // would a count leave 32 bits?  Then nothing of the region is written (uniform decision: two barriers).  Not in a FRESH
    // pass: its counts are the batch's deltas, and a bucket of 2^32 - 1 payloads or more was refused at the top.
    if (NARROW && !FRESH && !s_fail) {
        bool wide_cnt = false;
#pragma unroll
        for (int q = 0; q < SPL; ++q) {
            const uint32_t i = (uint32_t)q * NT + tid;
            const u64 cc = (oldn[q] >> 32) + s_add[i] + (i == sp_off ? s_special : 0u);
            wide_cnt |= cc > 0xFFFFFFFFull;
        }
        if (wide_cnt) s_fail = 2;
        __syncthreads();
    }
    if (s_fail) {
        if (FRESH && dirty) write_empty();
        if (tid == 0) {
            rfail[r] = (uint8_t)s_fail;
            rnew[r] = 0;
        }
        return;
    }
    uint4 *o4 = reinterpret_cast<uint4 *>(reg);
    uint32_t nheads = 0;
    bool too_wide = false;
    if (NARROW) {
        u64 *const nout = ntab + r * REGION_SLOTS;
#pragma unroll
        for (int q = 0; q < SPL; ++q) {
            const uint32_t i = (uint32_t)q * NT + tid;
            const uint32_t delta = s_add[i];
            u64 cc = (oldn[q] >> 32) + delta;
            uint32_t pp = (oldn[q] >> 32) ? (uint32_t)oldn[q] : s_pay[i];  // an old slot keeps its payload; a new one's is in the image
            if (i == sp_off) {
                if (s_sp_new) pp = R32_FREE;
                cc += s_special;
            }
            nout[i] = cc ? ((cc << 32) | pp) : 0ull;
            if (FRESH && head_cb && cc) {
                nheads += (uint32_t)((cc + (1ull << head_cb) - 1) >> head_cb);
                too_wide |= cc > (64ull << head_cb);
            }
        }
    } else
#pragma unroll
    for (int q = 0; q < SPL; ++q) {
        const uint32_t i = (uint32_t)q * NT + tid;
        u64 kk = old[q].key, cc = old[q].count;
        const uint32_t delta = s_add[i];
        if (delta) {
#if KH_ABLR & 4  /* timing experiment: no inverse hash in the write-back */
            if (kk == KH_EMPTY_KEY) kk = s_pay[i];
#else
            if (kk == KH_EMPTY_KEY) kk = Pay<uint32_t>::key(s_pay[i], p1, g);  // new key: invert the hash
#endif
            cc += delta;
        }
        if (i == sp_off) {
            if (s_sp_new) kk = Pay<uint32_t>::key(R32_FREE, p1, g);
            cc += s_special;
        }
        o4[i] = make_uint4((uint32_t)kk, (uint32_t)(kk >> 32), (uint32_t)cc, (uint32_t)(cc >> 32));
        if (FRESH && head_cb && kk != KH_EMPTY_KEY) {
            nheads += (uint32_t)((cc + (1ull << head_cb) - 1) >> head_cb);
            too_wide |= cc > (64ull << head_cb);
        }
    }
    if (tid == 0) {
        rnew[r] = s_new;
        rreal[r] = s_real;
    }
    if (FRESH && head_cb) {
        const uint32_t hw = (uint32_t)wave_sum((u64)nheads);
        if ((tid & 63) == 0 && hw) atomicAdd(&s_heads, hw);
        if (kh_any(too_wide) && (tid & 63) == 0) atomicOr((unsigned long long *)&ctr->heads_wide, 1ull);
        __syncthreads();
        if (tid == 0) rheads[r] = s_heads;
    }
}

// Folds the per-region results of one region_count pass into the context counters.
KH_GLOBAL __launch_bounds__(BLOCK) void region_reduce_kernel(const u64 *__restrict__ bstart, const uint8_t *__restrict__ rfail,
                                                              const uint32_t *__restrict__ rnew, const u64 *__restrict__ rreal,
                                                              u64 nregions, Counters *ctr) {
    const u64 stride = (u64)gridDim.x * BLOCK;
    u64 d = 0, km = 0, nf = 0;
    for (u64 r = (u64)blockIdx.x * BLOCK + threadIdx.x; r < nregions; r += stride) {
        if (rfail[r]) {
            ++nf;
        } else {
            d += rnew[r];
            km += rreal[r];  // k-mers of the bucket: its size minus the sentinels that pad its segments to whole units
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
// One workgroup per ORIGINAL region index; rfail/bstart/g refer to the geometry the buckets were
// built with, `tg` to the grown table.
template <typename PT>
__global__ __launch_bounds__(BLOCK) void failed_buckets_insert_kernel(TableGeom tg, PartGeom g, const PT *__restrict__ pays,
                                                                      const u64 *__restrict__ bstart, const u64 *__restrict__ bend,
                                                                      const uint8_t *__restrict__ rfail, Counters *ctr) {
    const u64 r = blockIdx.x;
    if (!rfail[r]) return;
    const u64 lo = bstart[r], hi = bend[r];
    const uint32_t p1 = part_div_b2(g, r), digit = (uint32_t)r - p1 * g.b2;
    uint32_t nd = 0, nf = 0;
    u64 real = 0;
    for (u64 i = lo + threadIdx.x; i < hi; i += BLOCK) {
        const PT v = pays[i];
        // 32-bit payloads: a payload of another bucket is a sentinel (line padding), not a k-mer
        if (sizeof(PT) == 4 && kh_bucket_of_x((uint32_t)v, g.b2) != digit) continue;
        if (sizeof(PT) == 8 && (u64)v == KH_EMPTY_KEY) continue;  // 64-bit payloads: the empty key pads the segments
        ++real;
        upsert(tg, Pay<PT>::key(v, p1, g), 1ull, nd, nf);
    }
    const u64 d = wave_sum((u64)nd), f = wave_sum((u64)nf), km = wave_sum(real);
    if (lane_id() == 0) {  // rare path (only regions that overflowed): plain counter atomics are fine
        if (d) atomicAdd(&ctr->distinct, d);
        if (f) atomicAdd(&ctr->failed, f);
        if (km) atomicAdd(&ctr->kmers, km);
    }
}

// ---- hot buckets ------------------------------------------------------------------------------------------------------
// The region pass gives every bucket to ONE workgroup, which takes ~1.4 G payloads/s: a bucket holding a thousandth of the
// batch costs as much as the whole pass (10 % poly-A reads, round 3: 65 M copies of one payload, 47.6 ms in one workgroup
// behind a 24 ms pass).  Such a bucket is dominated by one or a few keys -- the hash spreads distinct keys evenly -- so the
// region pass skips buckets above `cut` payloads (the same rule, in the same launch, as hot_list_kernel), and after it
// hot_buckets_kernel spreads each of them over the whole grid in slices: every workgroup adds up its slices in a small LDS
// table (payload -> count) and applies the sums to the table with device atomics, a few per workgroup instead of one per
// payload.  A hot bucket that is NOT dominated by few keys still comes out right: the LDS table is applied and cleared
// whenever it is half full.  (16-byte table only: kmerhip.hip widens an 8-byte image first.)
constexpr int HOT_SLICE = 16384;  // payloads per slice
constexpr int HOT_TAB = 4096;     // entries of the LDS table (applied when more than half are taken)
constexpr int HOT_GRID = 1024;

KH_GLOBAL __launch_bounds__(BLOCK) void hot_list_kernel(const u64 *__restrict__ bstart, const u64 *__restrict__ bend, u64 nregions, u64 cut,
                                                         uint32_t *__restrict__ list, Counters *ctr) {
    const u64 r = (u64)blockIdx.x * BLOCK + threadIdx.x;
    if (r < nregions && bend[r] - bstart[r] > cut) {
        list[atomicAdd(&ctr->hot, 1ull)] = (uint32_t)r;
        atomicAdd(&ctr->hot_total, bend[r] - bstart[r]);
    }
}

// NARROW: into the 8-byte image (count << 32 | payload) -- only where no count can leave 32 bits (the host checks the table's
// k-mer total), so that input with a few hot keys keeps its image: no widening pass, and the passes after it stay narrow.
template <typename PT, bool NARROW = false>
__global__ __launch_bounds__(BLOCK) void hot_buckets_kernel(TableGeom tg, PartGeom g, const PT *__restrict__ pays, const u64 *__restrict__ bstart,
                                                            const u64 *__restrict__ bend, const uint32_t *__restrict__ list, u64 nhot,
                                                            Counters *ctr, u64 *__restrict__ ntab = nullptr) {
    constexpr PT FREE = (PT)~(PT)0;  // (32-bit payloads: a legal value, counted apart; 64-bit: KH_EMPTY_KEY, the padding)
    __shared__ PT s_key[HOT_TAB];
    __shared__ uint32_t s_cnt[HOT_TAB];
    __shared__ uint32_t s_fill, s_free_cnt;
    const int tid = threadIdx.x;
    for (int i = tid; i < HOT_TAB; i += BLOCK) {
        s_key[i] = FREE;
        s_cnt[i] = 0;
    }
    if (tid == 0) {
        s_fill = 0;
        s_free_cnt = 0;
    }
    __syncthreads();
    uint32_t nd = 0, nf = 0;
    u64 real = 0;
    for (u64 b = 0; b < nhot; ++b) {
        const u64 r = list[b];
        const u64 lo = bstart[r], n = bend[r] - lo;
        const uint32_t p1 = part_div_b2(g, r), digit = (uint32_t)r - p1 * g.b2;
        auto apply = [&]() {  // the LDS table -> the table in HBM; leaves it empty (all threads)
            __syncthreads();
            for (int i = tid; i < HOT_TAB; i += BLOCK) {
                const uint32_t cnt = s_cnt[i];
                if (cnt) {
                    if constexpr (NARROW) (void)narrow_upsert(ntab + r * REGION_SLOTS, g, (uint32_t)s_key[i], (u64)cnt, 0ull, nd, nf);
                    else upsert(tg, Pay<PT>::key(s_key[i], p1, g), (u64)cnt, nd, nf);
                    s_key[i] = FREE;
                    s_cnt[i] = 0;
                }
            }
            if (tid == 0) {
                if (sizeof(PT) == 4 && s_free_cnt) {
                    if constexpr (NARROW) (void)narrow_upsert(ntab + r * REGION_SLOTS, g, 0xFFFFFFFFu, (u64)s_free_cnt, 0ull, nd, nf);
                    else upsert(tg, Pay<PT>::key(FREE, p1, g), (u64)s_free_cnt, nd, nf);
                }
                s_free_cnt = 0;
                s_fill = 0;
            }
            __syncthreads();
        };
        bool touched = false;
        for (u64 s = blockIdx.x; s * HOT_SLICE < n; s += gridDim.x) {
            touched = true;
            const u64 s0 = s * HOT_SLICE, s1 = s0 + HOT_SLICE < n ? s0 + HOT_SLICE : n;
            for (u64 c0 = s0; c0 < s1; c0 += 4 * BLOCK) {  // 1024 payloads between two looks at the fill
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u64 i = c0 + (u64)j * BLOCK + tid;
                    if (i >= s1) continue;
                    const PT v = pays[lo + i];
                    // a payload with another level-2 digit (32-bit) / the empty key (64-bit) pads the segments: not a k-mer
                    if (sizeof(PT) == 4 && kh_bucket_of_x((uint32_t)v, g.b2) != digit) continue;
                    if (sizeof(PT) == 8 && (u64)v == KH_EMPTY_KEY) continue;
                    ++real;
                    if (v == FREE) {
                        atomicAdd(&s_free_cnt, 1u);
                        continue;
                    }
                    uint32_t h = ((uint32_t)((u64)v ^ ((u64)v >> 29)) * 2654435761u) >> 20;  // 12 bits
                    for (;;) {
                        PT cur = s_key[h];
                        if (cur == FREE) {
                            cur = atomicCAS(&s_key[h], FREE, v);
                            if (cur == FREE) {
                                atomicAdd(&s_fill, 1u);
                                cur = v;
                            }
                        }
                        if (cur == v) {
                            atomicAdd(&s_cnt[h], 1u);
                            break;
                        }
                        h = (h + 1) & (HOT_TAB - 1);
                    }
                }
                __syncthreads();
                const bool full = s_fill > HOT_TAB / 2;  // (uniform: read between two barriers; at most 4 * BLOCK more before the next look)
                __syncthreads();
                if (full) apply();
            }
        }
        if (touched) apply();  // the next bucket has another region
    }
    const u64 d = wave_sum((u64)nd), f = wave_sum((u64)nf), km = wave_sum(real);
    if (lane_id() == 0) {
        if (d) atomicAdd(&ctr->distinct, d);
        if (f) atomicAdd(&ctr->failed, f);
        if (km) atomicAdd(&ctr->kmers, km);
    }
}


}  // namespace kh
