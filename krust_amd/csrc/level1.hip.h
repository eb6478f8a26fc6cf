// level1.hip.h -- level 1 of the partitioned path: bases -> canonical k-mers -> per-partition chunks of payloads.
// Compiled in its own translation units (level1_32.hip, level1_64.hip: one kernel per k for the written-out window,
// window.hip.h), reached from kmerhip.hip through level1_api.h.  Design notes: partition.hip.h and the banners below.
#pragma once
#include <utility>

#include "part_common.hip.h"
#include "window.hip.h"

namespace kh {

// ---------------------------------------------------------------------------------------------
// level 1, single pass: extraction + scatter into pool chunks, no counting pass
// ---------------------------------------------------------------------------------------------
struct ChunkDst {  // per partition, per batch: where staged element i (local index e = i - lofs) goes
    u64 a;         // e <  split: pool index = a + i   (the partition's current chunk)
    u64 b;         // e >= split: pool index = b + i   (freshly taken, consecutive chunks)
};
// a and b are biased by -lofs (and b by -split), so they wrap below zero for the first chunks of the pool:
// b lies in [-(RTILE + CHUNK_PAY), pool size) modulo 2^64, any value in there is a real destination
// (-1 included: it comes up once in a few hundred batches).  The "drop" marker sits far outside.
constexpr u64 CHUNK_DST_DROP = 1ull << 63;

// PT = uint32_t: one sorting round of 16 windows per lane per tile.  PT = u64 (k >= 22): TWO rounds of 8
// windows per lane, so that the staged payloads take the same 64 KiB of LDS and the per-partition runs
// the same 64 bytes; the extraction state (Roller) simply carries on between the rounds.
// KT: 0 = k is a run-time value; 21 / 31 = the kernel is compiled for that k (the BASELINE configurations):
// window masks, the revcomp insert position and the Feistel shifts become immediates, the 64-bit shift that
// splits the key into its halves becomes one v_alignbit, and for 21 the level-1 geometry (1024 partitions,
// payload = the low 32 hash bits) is fixed too.  Same values as the generic form (the tests run both).
template <bool QUAL, int MODE, bool FAST, typename PT, int KT>
__global__ __launch_bounds__(PART_NT) void part1_scatter_chunked_kernel(
    const uint8_t *__restrict__ abase, const uint8_t *__restrict__ qbase, int qaligned, u64 vbeg, u64 vend, u64 wlo,
    u64 tile0, u64 ntiles, uint32_t tiles_per_block, uint32_t k, uint32_t thr, PartGeom g, PT *__restrict__ pool,
    uint16_t *__restrict__ chunk_part, uint8_t *__restrict__ fill8, u64 *__restrict__ pool_next, u64 pool_chunks,
    Counters *ctr) {
    constexpr int ROUNDS = sizeof(PT) == 8 ? 2 : 1;
    constexpr int WPR = CHUNK / ROUNDS;         // windows per lane per round
    constexpr int RTILE = PART_NT * WPR;        // staged payloads per round
    __shared__ uint32_t s_code[2][PART_NT + 2];
    __shared__ uint16_t s_val[2][PART_NT + 2];
    __shared__ PT s_stage[RTILE + 2];           // 64 KiB (+ a trash slot for windows without a key)
    __shared__ uint16_t s_pid[RTILE + 2];       // 32 / 16 KiB
    __shared__ uint32_t s_cnt[MAX_P1];
    __shared__ uint32_t s_meta[MAX_P1];         // lofs | split << 16
    __shared__ ChunkDst s_dst[MAX_P1];          // 16 KiB
    __shared__ uint32_t s_wsum[4];
    __shared__ uint16_t s_lofs[MAX_P1];
    __shared__ u64 s_priv_next, s_priv_end;     // the workgroup's private range of chunk ids
    const int tid = threadIdx.x;
    if (KT) k = KT;
    const uint32_t p1b = (KT == 21 && FAST) ? 10u : g.p1_bits;
    s_cnt[tid] = 0;
    if (tid == 0) {
        s_priv_next = atomicAdd(pool_next, (u64)POOL_GRAB);
        s_priv_end = s_priv_next + POOL_GRAB;
    }
    // lane tid owns partition tid: its current chunk and how full it is
    u64 cur = 0;
    uint32_t fill = CHUNK_PAY;  // "full": the first payload takes a chunk
    bool have_chunk = false;
    const u64 tb = tile0 + (u64)blockIdx.x * tiles_per_block;
    u64 te = tb + tiles_per_block;
    if (te > tile0 + ntiles) te = tile0 + ntiles;
    int buf = 0;
    uint32_t lost = 0;
    __syncthreads();
    // The next tile's bases are requested right after this tile's first barrier and encoded into the other code buffer
    // BEFORE this tile's first write-out: a wait for them placed after stores is a wait for the stores' acknowledgement
    // (vmcnt counts both; see part1_bins_kernel).
    {
        const RawChunk raw0 = load_raw<QUAL>(abase, qbase, qaligned, chunk_pos<PART_NT>(tb, tid), vbeg, tb < te ? vend : 0);
        stage_encode<QUAL, PART_NT>(s_code, s_val, 0, true, tid, raw0, abase, qbase, qaligned, tb, vbeg, vend, thr);
    }
    for (u64 t = tb; t < te; ++t, buf ^= 1) {
        __syncthreads();
        const RawChunk raw = load_raw<QUAL>(abase, qbase, qaligned, chunk_pos<PART_NT>(t + 1, tid), vbeg, t + 1 < te ? vend : 0);
        const WinCtx w = stage_collect<PART_NT>(s_code, s_val, buf, tid, t);
        Roller roll;
        roll.init(w, k, wlo);
#pragma unroll
        for (int h = 0; h < ROUNDS; ++h) {
            PT pay[WPR];
            uint32_t tag[WPR];  // (p1 << 16) | rank-in-partition, 0xFFFFFFFF = no key
#pragma unroll
            for (int j = 0; j < WPR; ++j) {
                u64 key;
#if KH_ABL & 8  /* timing experiment: no window extraction */
                key = (u64)roll.code * (2 * j + 1) + tid;
                const bool ok = true;
#else
                const bool ok = roll.next(h * WPR + j, key);
#endif
                uint32_t p1 = 0;
                // Without quality masking nearly every window is valid (N is rare): hashing unconditionally
                // is cheaper than an exec-mask region per window.  With -Q ~40 % of the windows are masked,
                // there the branch pays.
                if (!QUAL || ok) {
                    if (sizeof(PT) == 8) {
                        const u64 H = kh_table_hash<MODE>(key, k) << g.shard_shift;
                        pay[j] = (PT)(H << g.p1_bits);  // (Pay<u64>::make: the hash below the level-1 digit)
                        p1 = p1_of_hash(H, g);
                    } else if (FAST) {
                        uint32_t pw;
#if KH_ABL & 1  /* timing experiment: no hash */
                        p1 = (uint32_t)key & 1023u;
                        pw = (uint32_t)(key >> 10);
#else
                        hash_p1_pay32<MODE>(k, p1b, key, p1, pw);
#endif
                        pay[j] = (PT)pw;
                    } else {
                        const u64 H = part_hash<MODE>(g, key);
                        pay[j] = (PT)Pay<uint32_t>::make(key, H, g);
                        p1 = p1_of_hash(H, g);
                    }
                }
                tag[j] = ok ? (p1 << 16) : 0xFFFFFFFFu;
            }
            // ranks in a second sweep: all LDS atomics in flight instead of one wait per window
#pragma unroll
            for (int j = 0; j < WPR; ++j)
#if KH_ABL & 16  /* timing experiment: no rank atomics (everything then collapses to an empty sort) */
                tag[j] &= 0xFFFF0000u;
#else
                if (tag[j] != 0xFFFFFFFFu) tag[j] |= atomicAdd(&s_cnt[tag[j] >> 16], 1u);
#endif
            __syncthreads();
            block_exclusive_scan_1024(s_cnt, s_lofs, s_wsum, tid);
            // Branch-free staging: every lane reads its run starts back to back (one wait instead of an
            // exposed LDS round trip behind a branch per window); windows without a key go to a trash slot.
            uint32_t rs[WPR];
#pragma unroll
            for (int j = 0; j < WPR; ++j) rs[j] = s_lofs[(tag[j] >> 16) & (MAX_P1 - 1)];
#pragma unroll
            for (int j = 0; j < WPR; ++j) {
                const uint32_t slot = tag[j] != 0xFFFFFFFFu ? rs[j] + (tag[j] & 0xFFFFu) : (uint32_t)RTILE;
                s_stage[slot] = pay[j];
                s_pid[slot] = (uint16_t)(tag[j] >> 16);
            }
            {  // lane tid places partition tid's run: the rest of its current chunk, then fresh chunks
                const uint32_t c = s_cnt[tid], lo = s_lofs[tid];
                const uint32_t space = CHUNK_PAY - fill;
                ChunkDst d;
                d.a = cur * CHUNK_PAY + fill - lo;
                d.b = 0;
                if (c > space) {
                    const uint32_t r = c - space;
                    const uint32_t nnew = (r + CHUNK_PAY - 1) / CHUNK_PAY;
                    u64 first = atomicAdd(&s_priv_next, (u64)nnew);  // LDS
                    if (first + nnew > s_priv_end) first = atomicAdd(pool_next, (u64)nnew);  // private range ran out (rare)
                    if (first + nnew > pool_chunks) {  // cannot happen with the host's pool sizing; never write past it
                        lost += r;
                        first = 0;
                        d.b = CHUNK_DST_DROP;  // marks "drop" for the write-out
                    } else {
                        for (uint32_t q = 0; q < nnew; ++q) chunk_part[first + q] = (uint16_t)tid;
                        d.b = first * CHUNK_PAY - space - lo;
                        cur = first + nnew - 1;
                        fill = r - (nnew - 1) * CHUNK_PAY;
                        have_chunk = true;
                    }
                } else {
                    fill += c;
                }
                s_dst[tid] = d;
                s_meta[tid] = lo | (space << 16);
            }
            const uint32_t total = (uint32_t)s_lofs[MAX_P1 - 1] + s_cnt[MAX_P1 - 1];
            __syncthreads();
            s_cnt[tid] = 0;  // ordered before the next atomics by the barrier below / the next tile's stage_tile() barrier
            if (tid == 0 && s_priv_next + POOL_LOW > s_priv_end) {  // refill the private range for the next round
                s_priv_next = atomicAdd(pool_next, (u64)POOL_GRAB);
                s_priv_end = s_priv_next + POOL_GRAB;
            }
            if (h == 0)  // tile t + 1's codes -> the other buffer, before any store of this tile
                stage_encode<QUAL, PART_NT>(s_code, s_val, buf ^ 1, false, tid, raw, abase, qbase, qaligned, t + 1, vbeg, vend, thr);
#if !(KH_ABL & 4)  /* timing experiment: no write-out */
#pragma unroll 2
            for (uint32_t i = tid; i < total; i += PART_NT) {
                const uint32_t p = s_pid[i];
                const uint32_t meta = s_meta[p];
                const ChunkDst d = s_dst[p];
                const uint32_t e = i - (meta & 0xFFFFu);
#if KH_ABL & 2  /* timing experiment: LDS side of the write-out only, no global stores */
                if (s_stage[i] == 0x12345678u && e == 77 && d.a == 5) pool[0] = 1;
#else
                if (e < (meta >> 16)) pool[d.a + i] = s_stage[i];
                else if (d.b != CHUNK_DST_DROP) pool[d.b + i] = s_stage[i];
#endif
            }
#endif
            // s_stage / s_dst / s_meta are rewritten only after the next round's / tile's barriers; the
            // counters, though, are hit by the next round's atomics right away
            if (h + 1 < ROUNDS) __syncthreads();
        }
    }
    if (have_chunk) fill8[cur] = (uint8_t)(fill - 1);
    const u64 l = wave_sum((u64)lost);
    if (lane_id() == 0 && l) atomicAdd(&ctr->failed, l);
}

constexpr uint32_t P1B_CAP = 32;                 // payloads per partition bin
constexpr uint32_t P1B_WORDS = MAX_P1 * P1B_CAP;   // 128 KiB

// ---------------------------------------------------------------------------------------------
// level 1 for 32-bit payloads: per-partition BINS in LDS, flushed in whole aligned 64-byte segments
// ---------------------------------------------------------------------------------------------
// What this kernel is shaped by was measured, not assumed (profiles/README.md r02c; tools/ubench/):
//  (1) THE STORE PATTERN.  scatter_runs.hip: 256 K lanes each appending to its own stream of 1-KiB pool chunks reach
//      1.3 TB/s in runs of 48 bytes and 2.8 TB/s in runs of one ALIGNED 64-BYTE SEGMENT (3.6 in 128-byte lines).  The
//      earlier level-1 kernels flushed every whole 16-byte unit of a partition every tile -- runs of ~56 bytes, 52 GB
//      of them per S100M batch: that was their whole 37 ms, whatever the instruction stream did.  Here a partition's
//      payloads collect in a 32-payload bin (128 KiB for 1024 partitions, which is what LDS there is), a flush writes
//      whole segments (16 payloads) and keeps up to 15 back, and to leave room for those the bins are flushed TWICE
//      per tile, after 8 windows per lane each (6.9 arrivals per partition on average: a bin overflows once in ~10^4
//      partition-flushes on well-mixed input).
//  (2) vmcnt COUNTS LOADS AND STORES ALIKE.  A wait for prefetched bases that sits after the flush -- where the
//      compiler puts it when the tile is staged at the top of the loop, or when the loaded registers are carried
//      around the loop (it copies them at the back edge) -- is a wait for the acknowledgement of every store just
//      issued.  So the next tile's bases are requested right after B0 and encoded into the other code buffer right
//      after the first B1 of the same iteration, before any store of it.
//  (3) No sorting pass: a payload's place is bin(p) + rank, known when the rank atomic returns -- no scan of the
//      1024 counts, no region table, two barriers per flush:
//          B0/B2  (codes staged / previous flush over)
//          8 windows: hash; rank = atomicAdd(&s_cnt[p], 1); s_bin[p][rank] = payload
//          B1
//          lane p (owner of partition p): whole segments of its bin -> the partition's chunk; the <= 15 payloads left
//          over move to the front of the bin; s_cnt[p] = that count
//      A window without a key bumps the lane's own waste counter, which starts every flush at 0x10000 (K21_WASTE0): the
//      rank it returns fails the "< 32" test that guards the store by itself, and "some real rank did not fit" is
//      (OR of all ranks) & 0xFF80 (ranks are kept as byte offsets, 4 x rank: see (6)).
//  (4) Overflow is exact, not a fallback to another kernel -- and it is the normal case on skewed input (a
//      homopolymer run sends a tile's 16384 payloads to ONE partition): a real rank >= 32 raises s_flag; after B1 the
//      owner, which sees the partition's full count c, reserves room for all c / 16 segments in the partition's chunk
//      sequence as usual, flushes the bin's two, and leaves in the (now free) bin where the others go; one more
//      barrier (taken only then), and the payloads that did not fit store themselves, 4 bytes each.  They do not keep
//      payload and rank in registers across the flush: a lane remembers WHICH of its windows they were, rolls over its
//      windows again and takes a second rank from the same counter, which the owner has restarted at -(whole
//      segments' worth of them): a negative rank is a position in the run, 0..14 a carried payload's bin slot, and
//      the counter ends at c % 16 as it must.  Chunk fill levels stay multiples of 16 until the end of the kernel:
//      same pool format, same reader.
//  (5) THE INSTRUCTION STREAM, for k = 21 at the headline geometry (1024 partitions).  valu_rates.hip: per wave, at
//      4 waves per SIMD, v_xor/and/or/add/sub/lshrrev/mov/not and v_bitop3 cost ~2.9 cycles; v_lshlrev, v_min/max,
//      v_bfe, v_alignbit, every fused three-operand form and every multiply (24- and 32-bit alike) ~4.9; v_cmp ~5.5;
//      a v_cndmask on a mask in an SGPR pair ~3.4; a v_cndmask reading a VCC that the instruction before it did not
//      just write ~20 (the usual "v_cmp_lt_u64 vcc; v_cndmask; v_cndmask" of a 64-bit min: 27).  The compiler's code
//      for a window adds up to ~215 such cycles, ~60 instructions (both strands rolled through registers, a second VCC
//      read in the canonical choice, left shifts and compares for tags and addresses).  Written out it was ~27
//      instructions, ~105 cycles (rounds 2-5; 22 since round 6's two-instruction Feistel rounds), one asm statement per window
//      (the compiler schedules the sixteen as units and allocates everything but five scratch registers):
//        * no rolling state: the lane's 48 bases are three words (w2:w1:w0, first base in the top bits) and their
//          reverse complements three more (c2:c1:c0 = 2-bit groups reversed and inverted, made once per tile); BOTH
//          strands of window J are 42-bit fields of those at fixed offsets (forward: bit 2 (15 - J); reverse
//          complement: bit 2 (J + 12)): v_alignbit + v_bfe each;
//        * canonical choice: v_cmp_lt_u64 into an SGPR pair, two v_cndmask on it;
//        * Feistel rounds of a multiply and ONE v_bitop3 (round 6; window.hip.h): the upper half A stays left-aligned, the lower
//          half B right-aligned, rounds 1 / 3 are (A ^ B x C) & top bits, rounds 2 / 4 B ^= mulhi(A, C) & low bits; no shifts,
//          no copies (rounds 2-5: v_mul_u32_u24, v_lshrrev, v_xor with the halves swapping by name);
//        * outputs are what the LDS instructions need, derived from A by shift-right + and: the counter's byte address
//          (A >> 20) & 0xFFC, the bin's (A >> 15) & 0x1FF80, payload (A << 10) | B; a window without a key gets the lane's
//          waste counter by a sign-extended v_bfe of its validity bit + v_bitop3.
//      Other k / other geometries take the same kernel with the window in C++ (Roller + hash_p1_pay32).
// 2-bit groups of x reversed and complemented: base m of a code word (bits 31-2m..30-2m) lands, complemented, at bits 2m..2m+1
__device__ __forceinline__ uint32_t rev2_complement(uint32_t x) {
    const uint32_t y = __builtin_bitreverse32(~x);
    return ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
}

typedef __attribute__((address_space(3))) uint32_t lds_u32;
// (6) LDS BANKS (round 3; profiles/README.md r03).  Round 2's SQ counters: 62.5 % of this kernel's LDS cycles were bank
//      conflicts.  Two sources, both structural: a bin is 128 bytes, so lane p's bin starts at bank 32 p mod 64 -- the owner
//      lanes' ds_read_b128 of "unit j of my bin" in a flush all land on the same two 4-bank groups (8-way conflicts on
//      every read of every flushed segment), and a payload store goes to bank (rank mod 32) whatever its partition.
//      Hence the 16-byte units of bin p are kept in the order u ^ (p & 7): a permutation INSIDE the bin (the payloads of a
//      partition are a multiset: order is free), which spreads equal unit indices of neighbouring bins over all banks.
//      It costs the window two instructions (the rotation is derived from the bin offset itself) because the counters
//      count BYTES (a rank comes back as 4 x rank, 8 x in the 64-bit kernel): the store address is bin ^ rank, one
//      instruction as before (it was bin + 4 rank).  KH_L1_SWIZZLE=0 builds the unswizzled layout (A/B).
//      Also: one waste counter per LANE of the workgroup (there were 64, shared by the 16 waves): with -Q more than half
//      of the windows have no key, the shared counters ran past the bits the "some real rank did not fit" test looks at,
//      and nearly every flush took the slow path's extra barrier for nothing.
#ifndef KH_L1_SWIZZLE
#define KH_L1_SWIZZLE 1
#endif
constexpr uint32_t L1_ROT_MASK = KH_L1_SWIZZLE ? 0x70u : 0u;       // byte-offset bits the unit permutation touches
constexpr uint32_t K21_CNT_OFF = 0;                                // 1024 counters + one waste counter per lane
constexpr uint32_t K21_BIN_OFF = (MAX_P1 + PART_NT) * 4;           // the bins, 128 KiB
constexpr uint32_t K21_TRASH_OFF = K21_BIN_OFF + P1B_WORDS * 4;    // one unit nobody reads
constexpr uint32_t K21_WASTE0 = 0x10000u;  // counters count bytes: a partition's real count stays below (8192 + 15) * 4 (8 in the 64-bit kernel: (4096 + 7) * 8)
// byte offset of the bin of partition p, with the partition's unit rotation in bits 4..6
__device__ __forceinline__ uint32_t l1_bin_offset(uint32_t p) { return (p << 7) | ((p << 4) & L1_ROT_MASK); }
// word index w of a 4-byte-payload bin (u64 index of an 8-byte-payload bin) of partition p -> where it is kept
__device__ __forceinline__ uint32_t l1_word(uint32_t p, uint32_t w) { return w ^ (((p << 4) & L1_ROT_MASK) >> 2); }
__device__ __forceinline__ uint32_t l1_word64(uint32_t p, uint32_t w) { return w ^ (((p << 4) & L1_ROT_MASK) >> 3); }
__device__ __forceinline__ uint32_t l1_unit(uint32_t p, uint32_t u) { return u ^ (((p << 4) & L1_ROT_MASK) >> 4); }

// KW = 11..21: the written-out window for that k (window.hip.h; 1024 partitions, no shard shift: the host checks);
// KW = 0: the C++ window, MODE / FAST as in part1_scatter_chunked_kernel, k and the geometry are run-time values.
template <bool QUAL, int MODE, bool FAST, int KW>
__global__ __launch_bounds__(PART_NT) void part1_bins_kernel(
    const uint8_t *__restrict__ abase, const uint8_t *__restrict__ qbase, int qaligned, u64 vbeg, u64 vend, u64 wlo,
    u64 tile0, u64 ntiles, uint32_t tiles_per_block, uint32_t k_rt, uint32_t thr, PartGeom g, uint32_t *__restrict__ pool,
    uint16_t *__restrict__ chunk_part, uint8_t *__restrict__ fill8, u64 *__restrict__ pool_next, u64 pool_chunks,
    Counters *ctr) {
    constexpr bool ASM = KW != 0;
    const uint32_t k = ASM ? (uint32_t)KW : k_rt, p1b = ASM ? 10u : g.p1_bits;
    constexpr int HALF = CHUNK / 2;          // windows per lane per flush
    constexpr uint32_t SEG = 16;             // payloads per 64-byte segment: what a flush writes is whole segments
    __shared__ __attribute__((aligned(16))) uint32_t s_mem[(K21_TRASH_OFF + 16) / 4];
    __shared__ uint32_t s_code[2][PART_NT + 2];
    __shared__ uint16_t s_val[2][PART_NT + 2];
    __shared__ uint32_t s_flag;              // some rank of this half-tile did not fit its bin
    __shared__ u64 s_priv_next, s_priv_end;  // the workgroup's private range of chunk ids
    uint32_t *const s_cnt = s_mem + K21_CNT_OFF / 4;
    uint32_t *const s_bin = s_mem + K21_BIN_OFF / 4;
    __attribute__((address_space(3))) char *const lds = (__attribute__((address_space(3))) char *)s_mem;
    const int tid = threadIdx.x;
    s_cnt[tid] = 0;                    // (the counters count BYTES: 4 x payloads)
    s_cnt[MAX_P1 + tid] = K21_WASTE0;  // the lane's own waste counter
    if (tid == 0) {
        s_flag = 0;
        s_priv_next = atomicAdd(pool_next, (u64)POOL_GRAB);
        s_priv_end = s_priv_next + POOL_GRAB;
    }
    // lane tid owns partition tid: its current chunk and how full it is (a multiple of 16 until the very end)
    u64 cur = 0;
    uint32_t fill = CHUNK_PAY;  // "full": the first segment takes a chunk
    uint32_t res = 0;           // payloads carried in the bin (== s_cnt[tid] between flushes)
    bool have_chunk = false;
    const u64 tb = tile0 + (u64)blockIdx.x * tiles_per_block;
    u64 te = tb + tiles_per_block;
    if (te > tile0 + ntiles) te = tile0 + ntiles;
    int buf = 0;
    uint32_t lost = 0;
    const uint32_t waste = K21_CNT_OFF + 4u * (MAX_P1 + (uint32_t)tid);
    uint32_t *const bin = s_bin + (uint32_t)tid * P1B_CAP;
    __syncthreads();
    auto take_chunk = [&](u64 &first) -> bool {
        first = atomicAdd(&s_priv_next, 1ull);  // LDS
        if (first + 1 > s_priv_end) first = atomicAdd(pool_next, 1ull);  // private range ran out (rare)
        if (first + 1 > pool_chunks) return false;  // cannot happen with the host's pool sizing; never write past it
        chunk_part[first] = (uint16_t)tid;
        return true;
    };
    // Lane tid flushes partition tid, in WHOLE ALIGNED 64-BYTE SEGMENTS: tools/ubench/scatter_runs.hip measures
    // what the memory system takes from 256 K lanes each appending to its own stream of 1-KiB chunks: 1.3 TB/s in
    // runs of 48 bytes, 2.8 TB/s in runs of one aligned 64-byte segment (3.6 in 128-byte lines) -- and the
    // 56-byte runs of a flush of every whole 16-byte unit, 52 GB of them per S100M batch, were the whole 37 ms of
    // this kernel, whatever the instruction stream did.  So a partition keeps up to 15 payloads back; to have room
    // for them in a 32-payload bin the bins are flushed twice per tile, after 8 windows per lane each (6.9
    // arrivals per partition on average; a bin overflows once in ~10^4 partition-flushes on well-mixed input).
    // c = what the bin's counter says (carried + new, possibly more than fit).  Returns the carried count.
    auto flush = [&](uint32_t c) -> uint32_t {
        const uint32_t nseg = c / SEG;                        // whole segments of the partition's run ...
        const uint32_t bseg = min(nseg, P1B_CAP / SEG);       // ... of which in the bin (the others: slow path)
        const uint32_t r = c % SEG;
        uint32_t nout = nseg;                                 // segments that find room in the pool
        u64 dst[P1B_CAP / SEG];                               // pool index of the bin's segments
        const uint32_t space = (CHUNK_PAY - fill) / SEG;      // segments left in the current chunk
        u64 ib = 0;                                           // first fresh chunk taken, if any (they are consecutive only
        uint32_t ntaken = 0;                                  //   when taken by one call: here one chunk at a time)
        // the run's segments fill the current chunk, then fresh chunks one after the other
        u64 run_a = cur * CHUNK_PAY + fill, run_b = 0;        // run position e < 16 space goes to run_a + e, else run_b + e
        if (nseg > space) {
            const uint32_t need = nseg - space;               // segments beyond the current chunk
            const uint32_t nnew = (need * SEG + CHUNK_PAY - 1) / CHUNK_PAY;
            u64 first = 0;
            bool ok = true;
            if (nnew == 1) ok = take_chunk(first);
            else {  // only a skewed batch does this: several consecutive chunks at once
                first = atomicAdd(pool_next, (u64)nnew);
                ok = first + nnew <= pool_chunks;
                if (ok) for (uint32_t q = 0; q < nnew; ++q) chunk_part[first + q] = (uint16_t)tid;
            }
            if (!ok) {
                lost += need * SEG;
                nout = space;
            } else {
                ib = first;
                ntaken = nnew;
                run_b = first * CHUNK_PAY - (u64)space * SEG;
                cur = first + nnew - 1;
                fill = need * SEG - (nnew - 1) * CHUNK_PAY;
                have_chunk = true;
            }
        } else {
            fill += nseg * SEG;
        }
        (void)ib; (void)ntaken;
#pragma unroll
        for (uint32_t sg = 0; sg < P1B_CAP / SEG; ++sg) dst[sg] = (sg < space ? run_a : run_b) + (u64)sg * SEG;
        const uint32_t nb = min(bseg, nout);
        uint4 *const bin4 = reinterpret_cast<uint4 *>(bin);  // the bin's eight 16-byte units, kept in the order u ^ (tid & 7)
#pragma unroll
        for (uint32_t sg = 0; sg < P1B_CAP / SEG; ++sg)
            if (sg < nb) {
                uint4 *d = reinterpret_cast<uint4 *>(pool + dst[sg]);
                const uint4 x0 = bin4[l1_unit(tid, 4 * sg)], x1 = bin4[l1_unit(tid, 4 * sg + 1)], x2 = bin4[l1_unit(tid, 4 * sg + 2)],
                            x3 = bin4[l1_unit(tid, 4 * sg + 3)];
                d[0] = x0; d[1] = x1; d[2] = x2; d[3] = x3;
            }
        if (c <= P1B_CAP) {
            if (bseg) {  // what does not fill a segment moves to the front of the bin, 16 bytes at a time
                const uint32_t nu = (r + 3u) / 4u;
                for (uint32_t i = 0; i < nu; ++i) bin4[l1_unit(tid, i)] = bin4[l1_unit(tid, 4 * bseg + i)];
            }
            s_cnt[tid] = r * 4u;
        } else {  // where the payloads that did not fit go: left in the second half of the emptied bin
            bin[l1_word(tid, 16)] = (uint32_t)run_a;
            bin[l1_word(tid, 17)] = (uint32_t)(run_a >> 32);
            bin[l1_word(tid, 18)] = (uint32_t)run_b;
            bin[l1_word(tid, 19)] = (uint32_t)(run_b >> 32);
            bin[l1_word(tid, 20)] = space * SEG;  // run positions before this one go to run_a + e, the others to run_b + e
            bin[l1_word(tid, 21)] = nout * SEG;   // ... if below this (less than the next only when the pool ran out)
            bin[l1_word(tid, 22)] = nseg * SEG;   // end of the run's whole segments
            // Those payloads take a second rank in the slow path, counted from -(their share of whole segments):
            // negative = run position 16 nseg + rank, 0..14 = carried in bin slot rank; the counter ends at c % 16.
            s_cnt[tid] = (r - (c - P1B_CAP)) * 4u;
        }
        s_cnt[MAX_P1 + tid] = K21_WASTE0;
        return r;
    };
    // The bases are fetched and encoded ONE tile ahead, between B0 and the first flush: vmcnt counts loads and
    // stores alike, so a wait for the prefetched bases placed after a flush (where the compiler puts it if the tile
    // is staged at the top of the loop, or if the loaded registers are carried around the loop: it copies them at
    // the back edge) is a wait for the acknowledgement of every store the flush has just issued.
    {
        const RawChunk raw0 = load_raw<QUAL>(abase, qbase, qaligned, chunk_pos<PART_NT>(tb, tid), vbeg, tb < te ? vend : 0);
        stage_encode<QUAL, PART_NT>(s_code, s_val, 0, true, tid, raw0, abase, qbase, qaligned, tb, vbeg, vend, thr);
    }
    for (u64 t = tb; t < te; ++t, buf ^= 1) {
        __syncthreads();  // B0: tile t's codes are in s_code[buf], the previous flush is over
        const RawChunk raw = load_raw<QUAL>(abase, qbase, qaligned, chunk_pos<PART_NT>(t + 1, tid), vbeg, t + 1 < te ? vend : 0);
        const WinCtx w = stage_collect<PART_NT>(s_code, s_val, buf, tid, t);
        const uint32_t good = window_good(w, k, wlo);
        const uint32_t w0 = (uint32_t)w.lo64, w1 = (uint32_t)(w.lo64 >> 32), w2 = w.hi;
        const uint32_t c0 = ASM ? rev2_complement(w2) : 0u, c1 = ASM ? rev2_complement(w1) : 0u, c2 = ASM ? rev2_complement(w0) : 0u;
        Roller roll;  // (the C++ window rolls through the lane's 16 windows in order, across both halves)
        if (!ASM) roll.init(w, k, wlo);
        // the C++ window: same outputs as the written-out one (payload, byte address of the counter, byte offset of the bin)
        auto window = [&](int j, uint32_t &pay, uint32_t &cnta, uint32_t &binb) {
            u64 key;
            const bool ok = roll.next(j, key);
            uint32_t p1 = 0;
            pay = 0;
            if (!QUAL || ok) {  // (see part1_scatter_chunked_kernel)
                if (FAST) {
                    hash_p1_pay32<MODE>(k, p1b, key, p1, pay);
                } else {
                    const u64 H = part_hash<MODE>(g, key);
                    pay = Pay<uint32_t>::make(key, H, g);
                    p1 = p1_of_hash(H, g);
                }
            }
            cnta = ok ? K21_CNT_OFF + 4u * p1 : waste;
            binb = l1_bin_offset(p1);
        };
#define KH_W21(J)                                                                                            \
    {                                                                                                        \
        uint32_t cnta;                                                                                       \
        if constexpr (ASM) {                                                                                 \
            uint32_t flo, fhi, rlo, rhi;                                                                     \
            win_fields<KW ? KW : 21, J>(w0, w1, w2, c0, c1, c2, flo, fhi, rlo, rhi);                         \
            win_hash32<KW ? KW : 21, J>(flo, fhi, rlo, rhi, good, waste, L1_ROT_MASK, pay[(J) % HALF], cnta, binb[(J) % HALF]); \
        } else window(J, pay[(J) % HALF], cnta, binb[(J) % HALF]);                                           \
        rk[(J) % HALF] = __hip_atomic_fetch_add((lds_u32 *)(lds + cnta), 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
    }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t omask = 0;  // bit j: window h * 8 + j has a key and its rank did not fit the bin
            {
                uint32_t pay[HALF], binb[HALF], rk[HALF];
                if (h == 0) { KH_W21(0) KH_W21(1) KH_W21(2) KH_W21(3) KH_W21(4) KH_W21(5) KH_W21(6) KH_W21(7) }
                else { KH_W21(8) KH_W21(9) KH_W21(10) KH_W21(11) KH_W21(12) KH_W21(13) KH_W21(14) KH_W21(15) }
                uint32_t racc = 0;
#pragma unroll
                for (int j = 0; j < HALF; ++j) {
                    const uint32_t r4 = rk[j];  // 4 x rank: the byte offset in the bin, before the unit permutation
                    *(lds_u32 *)(lds + K21_BIN_OFF + (r4 < 4u * P1B_CAP ? (binb[j] ^ r4) : K21_TRASH_OFF - K21_BIN_OFF)) = pay[j];
                    racc |= r4;
                }
                if (racc & (K21_WASTE0 - 4u * P1B_CAP)) {  // a real rank (below the waste counters' range) of 32 or more
                    s_flag = 1u;
#pragma unroll
                    for (int j = 0; j < HALF; ++j)
                        if (rk[j] >= 4u * P1B_CAP && rk[j] < K21_WASTE0) omask |= 1u << j;
                }
            }
            if (h == 1 && tid == 0 && s_priv_next + 2 * POOL_LOW > s_priv_end) {  // refill the private range (nobody takes
                s_priv_next = atomicAdd(pool_next, (u64)POOL_GRAB);               // chunks between a B0/B2 and the next B1;
                s_priv_end = s_priv_next + POOL_GRAB;                             // a vmcnt wait here finds only old stores)
            }
            __syncthreads();  // B1
            const bool slow = s_flag != 0u;  // uniform
            if (h == 0)  // tile t + 1's codes -> the other buffer (its bases were requested at B0)
                stage_encode<QUAL, PART_NT>(s_code, s_val, buf ^ 1, false, tid, raw, abase, qbase, qaligned, t + 1, vbeg, vend, thr);
            res = flush(s_cnt[tid] >> 2);
            if (slow) {
                __syncthreads();  // B2'
                if (omask) {  // the registers of the fast path are gone: roll over the lane's windows again
                    Roller again;
                    again.init(w, k, wlo);
#pragma unroll
                    for (int j = 0; j < CHUNK; ++j) {
                        u64 key;
                        again.next(j, key);
                        if (j / HALF == h && ((omask >> (j % HALF)) & 1u)) {
                            uint32_t p, pv;
                            if (FAST) {
                                hash_p1_pay32<MODE>(k, p1b, key, p, pv);
                            } else {
                                const u64 H = part_hash<MODE>(g, key);
                                pv = Pay<uint32_t>::make(key, H, g);
                                p = p1_of_hash(H, g);
                            }
                            uint32_t *const pbin = s_bin + p * P1B_CAP;
                            const int32_t r2 = (int32_t)atomicAdd(&s_cnt[p], 4u) >> 2;
                            if (r2 >= 0) {
                                pbin[l1_word(p, (uint32_t)r2)] = pv;
                            } else {
                                const u64 ra = ((u64)pbin[l1_word(p, 17)] << 32) | pbin[l1_word(p, 16)];
                                const u64 rb = ((u64)pbin[l1_word(p, 19)] << 32) | pbin[l1_word(p, 18)];
                                const uint32_t split = pbin[l1_word(p, 20)], lim = pbin[l1_word(p, 21)];
                                const uint32_t e = pbin[l1_word(p, 22)] + (uint32_t)r2;
                                if (e < lim) pool[(e < split ? ra : rb) + e] = pv;
                            }
                        }
                    }
                }
                if (tid == 0) s_flag = 0u;  // (everybody read it before B2'; it is set again after the next barrier)
            }
            if (h == 0) __syncthreads();  // B2: the first flush is over (after the second one: the next tile's B0)
        }
#undef KH_W21
    }
    __syncthreads();  // (the last slow path may have left carried payloads in other lanes' bins)
    res = s_cnt[tid] >> 2;
    // the payloads still carried: one by one into the partition's chunk
    if (res) {
        bool room = true;
        if (fill + res > CHUNK_PAY) {  // (fill is a multiple of 16, so this means fill == 256: a fresh chunk)
            u64 first;
            room = take_chunk(first);
            if (room) {
                cur = first;
                fill = 0;
                have_chunk = true;
            } else {
                lost += res;
            }
        }
        if (room) {
            for (uint32_t i = 0; i < res; ++i) pool[cur * CHUNK_PAY + fill + i] = bin[l1_word(tid, i)];
            fill += res;
        }
    }
    if (have_chunk) fill8[cur] = (uint8_t)(fill - 1);
    const u64 l = wave_sum((u64)lost);
    if (lane_id() == 0 && l) atomicAdd(&ctr->failed, l);
}

// ---------------------------------------------------------------------------------------------
// level 1 for 64-bit payloads (k >= 22, or KMERHIP_PAYLOAD=64): the bins recipe with 8-byte payloads
// ---------------------------------------------------------------------------------------------
// Round 2 left every k >= 22 on part1_scatter_chunked_kernel: a tile is counting-sorted in LDS and every partition's
// run appended payload by payload -- short unaligned runs, the store pattern that cost the 32-bit path 10 ms
// (tools/ubench/scatter_runs.hip: 1.3 TB/s in 48-byte runs against 2.8 in aligned 64-byte segments).  S100M, k = 31,
// -Q 20: 42.9 of the step's 95.8 ms, 93.5 GB written for 42 GB of payloads.
// Same recipe as part1_bins_kernel, resized for 8 bytes per payload and the same 128 KiB of bins:
//   * 1024 partitions (level 2's arena kernel takes <= 512 buckets per partition: 2^19 regions need all ten bits here),
//     so a bin holds 16 payloads = two 64-byte SEGMENTS of 8; a flush writes whole segments and keeps <= 7 back;
//   * to leave room for those, the bins are flushed every FW windows per lane: 4, or 8 where fewer than 0.55 of the windows
//     survive masking -- 3.5-4 arrivals per partition and flush (round 6, level1_64.hip: rounds 3-5 flushed twice as often and
//     paid a barrier pair and ~80 instructions per wave for every flush: k = 31 -Q 20 30.3 -> 25.8 ms, unmasked 48.6 -> 41.0);
//   * overflow is exact, as there: the owner reserves the run's whole segments, leaves in the emptied bin where the
//     payloads that did not fit go, and those lanes roll over their windows again and take a second rank.
// KT: 0 = k is a run-time value, otherwise the kernel is compiled for that k (window masks, Feistel shifts and the
// revcomp insert position become immediates).
constexpr uint32_t P1B64_CAP = 16;                     // payloads per partition bin (128 bytes)
constexpr uint32_t P1B64_SEG = 8;                      // payloads per 64-byte segment
constexpr uint32_t K64_BIN_OFF = (MAX_P1 + PART_NT) * 4;   // 1024 counters + one waste counter per lane in front of the bins
constexpr uint32_t K64_TRASH_OFF = K64_BIN_OFF + MAX_P1 * P1B64_CAP * 8;

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <int... Is, typename F>
__device__ __forceinline__ void static_for_seq(std::integer_sequence<int, Is...>, F &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_seq(std::make_integer_sequence<int, N>{}, f);
}

// KW = 22..32: the written-out window for that k (window.hip.h; needs 1024 partitions and no shard shift: the host
// checks); KW = 0: the C++ window, k and the geometry are run-time values.
template <bool QUAL, int MODE, int KW, int FW>
__global__ __launch_bounds__(PART_NT) void part1_bins64_kernel(
    const uint8_t *__restrict__ abase, const uint8_t *__restrict__ qbase, int qaligned, u64 vbeg, u64 vend, u64 wlo,
    u64 tile0, u64 ntiles, uint32_t tiles_per_block, uint32_t k_rt, uint32_t thr, PartGeom g, u64 *__restrict__ pool,
    uint16_t *__restrict__ chunk_part, uint8_t *__restrict__ fill8, u64 *__restrict__ pool_next, u64 pool_chunks,
    Counters *ctr) {
    constexpr bool ASM = KW != 0;
    const uint32_t k = ASM ? (uint32_t)KW : k_rt;
    constexpr uint32_t SEG = P1B64_SEG, CAP = P1B64_CAP;
    constexpr int NFLUSH = CHUNK / FW;
    __shared__ __attribute__((aligned(16))) uint32_t s_mem[(K64_TRASH_OFF + 16) / 4];
    __shared__ uint32_t s_code[2][PART_NT + 2];
    __shared__ uint16_t s_val[2][PART_NT + 2];
    __shared__ uint32_t s_flag;              // some rank of this flush interval did not fit its bin
    __shared__ u64 s_priv_next, s_priv_end;  // the workgroup's private range of chunk ids
    // ---- -Q: ONLY THE LIVE CHUNKS ARE WORKED ON (round 6) ----
    // The reference skips ahead past a masked base (src/run.rs:543-548); here a window is a lane's fixed place, and with -Q 20 on
    // configs[2]'s reads 56 % of the windows are dead and still ran the whole window: 31.2 ms for 5.25 G k-mers where k = 21
    // without masks spends 25.4 on 12.7 G.  Skipping a window when it is dead in EVERY lane of its wave (what round 5's review asked
    // for) meets no such window (0.0 % of them: a wave spans seven reads; the A/B build, KH_L1_WAVE_SKIP, is 3.5 ms SLOWER) --
    // but masked bases kill 31 windows in a row, and 47 % of the CHUNKS (a lane's sixteen windows) are dead from end to end.  So the
    // lanes whose chunk has a countable window publish it -- (chunk index, its 16-bit window mask): one word in a list, a ballot and an
    // LDS add per wave --, lane i takes list entry i (a chunk's code words are in LDS already), and the waves behind the list's
    // end sit the window phases out: they only keep the barriers and flush their partitions' bins.  The list of tile t + 1 is made
    // DURING tile t, behind the barrier that follows its codes' staging (two lists, taken in turn): no barrier of its own.
    __shared__ uint32_t s_live[QUAL ? 2 : 1][QUAL ? PART_NT : 1];
    __shared__ uint32_t s_nlive[2];
    uint32_t *const s_cnt = s_mem;
    u64 *const s_bin = reinterpret_cast<u64 *>(s_mem + K64_BIN_OFF / 4);
    __attribute__((address_space(3))) char *const lds = (__attribute__((address_space(3))) char *)s_mem;
    typedef __attribute__((address_space(3))) u64 lds_u64;
    const int tid = threadIdx.x;
    s_cnt[tid] = 0;                    // (the counters count BYTES: 8 x payloads)
    s_cnt[MAX_P1 + tid] = K21_WASTE0;  // the lane's own waste counter
    if (tid == 0) {
        s_flag = 0;
        s_nlive[0] = s_nlive[1] = 0;
        s_priv_next = atomicAdd(pool_next, (u64)POOL_GRAB);
        s_priv_end = s_priv_next + POOL_GRAB;
    }
    // lane tid owns partition tid: its current chunk and how full it is (a multiple of 8 until the very end)
    u64 cur = 0;
    uint32_t fill = CHUNK_PAY;  // "full": the first segment takes a chunk
    bool have_chunk = false;
    const u64 tb = tile0 + (u64)blockIdx.x * tiles_per_block;
    u64 te = tb + tiles_per_block;
    if (te > tile0 + ntiles) te = tile0 + ntiles;
    int buf = 0;
    uint32_t lost = 0;
    const uint32_t waste = 4u * (MAX_P1 + (uint32_t)tid);  // byte address of the lane's waste counter
    u64 *const bin = s_bin + (uint32_t)tid * CAP;
    __syncthreads();
    auto take_chunk = [&](u64 &first) -> bool {
        first = atomicAdd(&s_priv_next, 1ull);  // LDS
        if (first + 1 > s_priv_end) first = atomicAdd(pool_next, 1ull);  // private range ran out (rare)
        if (first + 1 > pool_chunks) return false;  // cannot happen with the host's pool sizing; never write past it
        chunk_part[first] = (uint16_t)tid;
        return true;
    };
    // lane tid flushes partition tid in whole aligned 64-byte segments; c = what the bin's counter says (carried + new,
    // possibly more than fit)
    auto flush = [&](uint32_t c) {
        const uint32_t nseg = c / SEG;                        // whole segments of the partition's run ...
        const uint32_t bseg = min(nseg, CAP / SEG);           // ... of which in the bin (the others: slow path)
        const uint32_t r = c % SEG;
        uint32_t nout = nseg;                                 // segments that find room in the pool
        const uint32_t space = (CHUNK_PAY - fill) / SEG;      // segments left in the current chunk
        u64 run_a = cur * CHUNK_PAY + fill, run_b = 0;        // run position e < 8 space goes to run_a + e, else run_b + e
        if (nseg > space) {
            const uint32_t need = nseg - space;               // segments beyond the current chunk
            const uint32_t nnew = (need * SEG + CHUNK_PAY - 1) / CHUNK_PAY;
            u64 first = 0;
            bool ok = true;
            if (nnew == 1) ok = take_chunk(first);
            else {  // only a skewed batch does this: several consecutive chunks at once
                first = atomicAdd(pool_next, (u64)nnew);
                ok = first + nnew <= pool_chunks;
                if (ok) for (uint32_t q = 0; q < nnew; ++q) chunk_part[first + q] = (uint16_t)tid;
            }
            if (!ok) {
                lost += need * SEG;
                nout = space;
            } else {
                run_b = first * CHUNK_PAY - (u64)space * SEG;
                cur = first + nnew - 1;
                fill = need * SEG - (nnew - 1) * CHUNK_PAY;
                have_chunk = true;
            }
        } else {
            fill += nseg * SEG;
        }
        const uint32_t nb = min(bseg, nout);
        uint4 *const bin4 = reinterpret_cast<uint4 *>(bin);  // the bin's eight 16-byte units, kept in the order u ^ (tid & 7)
#pragma unroll
        for (uint32_t sg = 0; sg < CAP / SEG; ++sg)
            if (sg < nb) {
                uint4 *d = reinterpret_cast<uint4 *>(pool + (sg < space ? run_a : run_b) + (u64)sg * SEG);
                const uint4 x0 = bin4[l1_unit(tid, 4 * sg)], x1 = bin4[l1_unit(tid, 4 * sg + 1)], x2 = bin4[l1_unit(tid, 4 * sg + 2)],
                            x3 = bin4[l1_unit(tid, 4 * sg + 3)];
                d[0] = x0; d[1] = x1; d[2] = x2; d[3] = x3;
            }
        if (c <= CAP) {
            if (bseg) {  // what does not fill a segment moves to the front of the bin, 16 bytes at a time
                const uint32_t nu = (r + 1u) / 2u;
                for (uint32_t i = 0; i < nu; ++i) bin4[l1_unit(tid, i)] = bin4[l1_unit(tid, 4 * bseg + i)];
            }
            s_cnt[tid] = r * 8u;
        } else {  // where the payloads that did not fit go: left in the second half of the emptied bin
            bin[l1_word64(tid, 8)] = run_a;
            bin[l1_word64(tid, 9)] = run_b;
            bin[l1_word64(tid, 10)] = (u64)(space * SEG) | ((u64)(nout * SEG) << 32);  // run positions below the first go to run_a + e, the others
            bin[l1_word64(tid, 11)] = (u64)(nseg * SEG);                               // to run_b + e if below the second; end of the whole segments
            // those payloads take a second rank in the slow path, counted from -(their share of whole segments):
            // negative = run position 8 nseg + rank, 0..6 = carried in bin slot rank; the counter ends at c % 8
            s_cnt[tid] = (r - (c - CAP)) * 8u;
        }
        s_cnt[MAX_P1 + tid] = K21_WASTE0;
    };
    // level-1 digit and payload (Pay<u64>::make: the hash below the digit, left-aligned) of a key
    auto p1_pay_of = [&](u64 key, u64 &pay) -> uint32_t {
        const u64 H = kh_table_hash<MODE>(key, k) << g.shard_shift;
        pay = H << g.p1_bits;
        return p1_of_hash(H, g);
    };
    {
        const RawChunk raw0 = load_raw<QUAL>(abase, qbase, qaligned, chunk_pos<PART_NT>(tb, tid), vbeg, tb < te ? vend : 0);
        stage_encode<QUAL, PART_NT>(s_code, s_val, 0, true, tid, raw0, abase, qbase, qaligned, tb, vbeg, vend, thr);
    }
#ifndef KH_L1_LIVE_LIST
#define KH_L1_LIVE_LIST 1  // (0: A/B builds -- every lane works on its own chunk, dead or not)
#endif
    constexpr bool LIVE = QUAL && KH_L1_LIVE_LIST != 0;
    // the lanes publish their live chunks of tile tt (its codes staged in s_code[cbuf], a barrier behind them) into list lb
    auto build_list = [&](int cbuf, u64 tt, int lb) {
        WinCtx wn;
        wn.hi = s_code[cbuf][tid];
        wn.lo64 = ((u64)s_code[cbuf][tid + 1] << 32) | s_code[cbuf][tid + 2];
        wn.V = ((u64)s_val[cbuf][tid] << 32) | ((u64)s_val[cbuf][tid + 1] << 16) | (u64)s_val[cbuf][tid + 2];
        wn.p0 = (u64)chunk_pos<PART_NT>(tt, tid);
        const uint32_t gd = window_good(wn, k, wlo);
        const bool live = gd != 0u;
        const u64 m = kh_ballot(live);
        uint32_t lbase = 0;
        if (m) {
            if ((int)lane_id() == __builtin_ctzll(m)) lbase = atomicAdd(&s_nlive[lb], (uint32_t)__builtin_popcountll(m));
            lbase = (uint32_t)__shfl((int)lbase, __builtin_ctzll(m), 64);
        }
        if (live) s_live[lb][lbase + mbcnt(m)] = (uint32_t)tid | (gd << 16);
    };
    if constexpr (LIVE) {  // the first tile's list (later ones are made a tile ahead)
        __syncthreads();
        build_list(0, tb, 0);
    }
    for (u64 t = tb; t < te; ++t, buf ^= 1) {
        __syncthreads();  // B0: tile t's codes are in s_code[buf] (and its list of live chunks whole), the previous flush is over
        // (the next tile's bases: requested here, encoded before this tile's first store -- see part1_bins_kernel)
        const RawChunk raw = load_raw<QUAL>(abase, qbase, qaligned, chunk_pos<PART_NT>(t + 1, tid), vbeg, t + 1 < te ? vend : 0);
        WinCtx w = stage_collect<PART_NT>(s_code, s_val, buf, tid, t);
        uint32_t good = LIVE ? 0u : window_good(w, k, wlo);
        bool wave_busy = true;  // (uniform per wave)
        const int lcur = (int)((t - tb) & 1);  // this tile's list; the other one is filled for tile t + 1 below
        if constexpr (LIVE) {   // the live chunks, compacted: see the banner at s_live
            const uint32_t nlive = s_nlive[lcur];
            if (tid == 0) s_nlive[lcur ^ 1] = 0;  // (read by everybody a tile ago; added to again two barriers from here)
#ifndef KH_L1_WAVE_ORDER
#define KH_L1_WAVE_ORDER 0  // (1: A/B builds -- the list fills the waves in the order 0, 4, 8, 12, 1, 5, ...)
#endif
            // Which 64 entries a wave takes: wave w the w-th.  A workgroup's sixteen waves are dealt to the CU's four SIMDs round
            // robin, so nine busy waves are 3 + 2 + 2 + 2 of them -- and what the list buys is the busiest SIMD's three instead of
            // four (measured, configs[2]: level 1 31.2 -> 30.4 ms; with the order 0, 4, 8, 12, 1, ... -- the nine on two SIMDs --
            // 33.9 ms: that is also what the list itself costs, a barrier and ~40 instructions per lane and tile).
            const uint32_t wv = (uint32_t)tid >> 6, seg = KH_L1_WAVE_ORDER ? ((wv & 3u) << 2) | (wv >> 2) : wv;
            const uint32_t li = (seg << 6) | ((uint32_t)tid & 63u);
            wave_busy = (seg << 6) < nlive;
            const uint32_t item = li < nlive ? s_live[lcur][li] : 0u;  // (a lane behind the list's end: no countable window)
            const uint32_t c = item & 0xFFFFu;
            good = item >> 16;
            w.hi = s_code[buf][c];
            w.lo64 = ((u64)s_code[buf][c + 1] << 32) | s_code[buf][c + 2];
            w.V = ((u64)s_val[buf][c] << 32) | ((u64)s_val[buf][c + 1] << 16) | (u64)s_val[buf][c + 2];
            w.p0 = (u64)chunk_pos<PART_NT>(t, (int)c);
        }
        const uint32_t w0 = (uint32_t)w.lo64, w1 = (uint32_t)(w.lo64 >> 32), w2 = w.hi;
        const uint32_t c0 = ASM ? rev2_complement(w2) : 0u, c1 = ASM ? rev2_complement(w1) : 0u, c2 = ASM ? rev2_complement(w0) : 0u;
        Roller roll;  // (the C++ window rolls through the lane's 16 windows in order, across the flushes)
        if (!ASM) {
            roll.init(w, k, wlo);
            if constexpr (QUAL) roll.good = good;  // (a lane behind the list's end holds chunk 0's words and no window)
        }
        static_for<NFLUSH>([&](auto hc) {
            constexpr int h = decltype(hc)::value;
            uint32_t omask = 0;  // bit j: window h * FW + j has a key and its rank did not fit the bin
            if (wave_busy) {
                uint32_t klo[FW], khi[FW], binb[FW], rk[FW];
                static_for<FW>([&](auto jc) {
                    constexpr int j = decltype(jc)::value, J = h * FW + j;
                    uint32_t cnta;
#ifndef KH_L1_WAVE_SKIP
#define KH_L1_WAVE_SKIP 0  // (1: A/B builds -- VERDICT r5 next-3: a window that is dead in EVERY lane of the wave branches around its
                           //  hash.  Measured, round 6: profiles/README.md r06 -- such windows are 0.0 % of configs[2]'s; level 1 unchanged)
#endif
                    if constexpr (ASM) {
                        if (KH_L1_WAVE_SKIP && QUAL && !kh_any(((good >> (15 - J)) & 1u) != 0u)) {
                            klo[j] = khi[j] = 0u;
                            cnta = waste;
                            binb[j] = 0u;
                        } else {
                            uint32_t flo, fhi, rlo, rhi;
                            win_fields<KW ? KW : 31, J>(w0, w1, w2, c0, c1, c2, flo, fhi, rlo, rhi);
                            win_hash64<KW ? KW : 31, J>(flo, fhi, rlo, rhi, good, waste, L1_ROT_MASK, klo[j], khi[j], cnta, binb[j]);
                        }
                    } else {
                        u64 key, pv = 0;
                        const bool ok = roll.next(J, key);
                        uint32_t p = 0;
                        if (!QUAL || ok) p = p1_pay_of(key, pv);  // (without -Q nearly every window has a key: hashing unconditionally is cheaper than a branch per window)
                        klo[j] = (uint32_t)pv;
                        khi[j] = (uint32_t)(pv >> 32);
                        cnta = ok ? 4u * p : waste;
                        binb[j] = l1_bin_offset(p);
                    }
                    rk[j] = __hip_atomic_fetch_add((lds_u32 *)(lds + cnta), 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                });
                uint32_t racc = 0;
#pragma unroll
                for (int j = 0; j < FW; ++j) {
                    const uint32_t r8 = rk[j];  // 8 x rank: the byte offset in the bin, before the unit permutation (a waste counter's: the trash slot)
                    *(lds_u64 *)(lds + (r8 < 8u * CAP ? K64_BIN_OFF + (binb[j] ^ r8) : K64_TRASH_OFF)) = ((u64)khi[j] << 32) | klo[j];
                    racc |= r8;
                }
                if (racc & (K21_WASTE0 - 8u * CAP)) {  // a real rank (below the waste counters' range) of 16 or more
                    s_flag = 1u;
#pragma unroll
                    for (int j = 0; j < FW; ++j)
                        if (rk[j] >= 8u * CAP && rk[j] < K21_WASTE0) omask |= 1u << j;
                }
            }
            if (h == NFLUSH - 1 && tid == 0 && s_priv_next + 2 * POOL_LOW > s_priv_end) {  // refill the private range (nobody takes
                s_priv_next = atomicAdd(pool_next, (u64)POOL_GRAB);                        // chunks between a B0/B2 and the next B1)
                s_priv_end = s_priv_next + POOL_GRAB;
            }
            __syncthreads();  // B1
            const bool slow = s_flag != 0u;  // uniform
            if (h == 0)  // tile t + 1's codes -> the other buffer (its bases were requested at B0)
                stage_encode<QUAL, PART_NT>(s_code, s_val, buf ^ 1, false, tid, raw, abase, qbase, qaligned, t + 1, vbeg, vend, thr);
            flush(s_cnt[tid] >> 3);
            if (slow) {
                __syncthreads();  // B2'
                if (omask) {  // the registers of the fast path are gone: roll over the lane's windows again
                    Roller again;
                    again.init(w, k, wlo);
#pragma unroll
                    for (int j = 0; j < CHUNK; ++j) {
                        u64 key;
                        again.next(j, key);
                        if (j / FW == h && ((omask >> (j % FW)) & 1u)) {
                            u64 pv;
                            const uint32_t p = p1_pay_of(key, pv);
                            u64 *const pbin = s_bin + p * CAP;
                            const int32_t r2 = (int32_t)atomicAdd(&s_cnt[p], 8u) >> 3;
                            if (r2 >= 0) {
                                pbin[l1_word64(p, (uint32_t)r2)] = pv;
                            } else {
                                const u64 ra = pbin[l1_word64(p, 8)], rb = pbin[l1_word64(p, 9)];
                                const u64 d10 = pbin[l1_word64(p, 10)];
                                const uint32_t split = (uint32_t)d10, lim = (uint32_t)(d10 >> 32);
                                const uint32_t e = (uint32_t)pbin[l1_word64(p, 11)] + (uint32_t)r2;
                                if (e < lim) pool[(e < split ? ra : rb) + e] = pv;
                            }
                        }
                    }
                }
                if (tid == 0) s_flag = 0u;  // (everybody read it before B2'; it is set again after the next barrier)
            }
            if (h + 1 < NFLUSH) __syncthreads();  // B2: this flush is over (after the last one: the next tile's B0)
            if constexpr (LIVE && h == 0 && NFLUSH > 1) build_list(buf ^ 1, t + 1, lcur ^ 1);  // (tile t + 1's codes were staged before the barrier just passed)
        });
        if constexpr (LIVE && NFLUSH == 1) {  // (one flush per tile: the staging has no barrier behind it yet)
            __syncthreads();
            build_list(buf ^ 1, t + 1, lcur ^ 1);
        }
    }
    __syncthreads();  // (the last slow path may have left carried payloads in other lanes' bins)
    const uint32_t res = s_cnt[tid] >> 3;
    // the payloads still carried: one by one into the partition's chunk
    if (res) {
        bool room = true;
        if (fill + res > CHUNK_PAY) {  // (fill is a multiple of 8, so this means fill == 256: a fresh chunk)
            u64 first;
            room = take_chunk(first);
            if (room) {
                cur = first;
                fill = 0;
                have_chunk = true;
            } else {
                lost += res;
            }
        }
        if (room) {
            for (uint32_t i = 0; i < res; ++i) pool[cur * CHUNK_PAY + fill + i] = bin[l1_word64(tid, i)];
            fill += res;
        }
    }
    if (have_chunk) fill8[cur] = (uint8_t)(fill - 1);
    const u64 l = wave_sum((u64)lost);
    if (lane_id() == 0 && l) atomicAdd(&ctr->failed, l);
}

}  // namespace kh
