// shard.hip.h -- multi-GPU merge of per-GPU tables by hash range (no reference
// counterpart: the reference is single-process, src/run.rs:500-503 is its only parallelism).
//
// Ownership = the top bits of the table hash, i.e. a contiguous range of table regions.  A rank's
// table can therefore be exported region by region, already grouped by owner and ordered by
// region, with two streaming kernels and no scatter; and the owner's shard is itself a table over
// the remaining hash bits (TableGeom::shard_shift), rebuilt region by region in LDS from the
// senders' region segments -- no partition passes and no global atomics on the merge either.
#pragma once
#include "kernels.hip.h"

#include <type_traits>

namespace kh {

// What an export reads: the 16-byte table, or (ntab != nullptr) its 8-byte image -- count << 32 | the 32-bit payload
// of the partition passes, whose bits below the level-2 digit ARE the hash bits below the region index that the packed
// and the heads formats carry (round 3: a rank's table can stay in the narrow form through the exchange).
// Round 4: tables of any multiple of 1024 regions -- the window is kernels.hip.h kh_below_region (for a power-of-two table
// the 32 hash bits below the region index, as before).
struct SlotSrc {
    const Slot *table;
    const u64 *ntab;
    RegionGeom geo;    // the table's geometry; narrow: the geometry the image's payloads are relative to (the same layout)
};
struct SlotVal {
    bool live;
    u64 count;
    u64 key_or_pay;  // wide: the key; narrow: the payload
};
__device__ __forceinline__ SlotVal slot_read(const SlotSrc &src, u64 i) {
    SlotVal v;
    if (src.ntab) {
        const u64 sl = src.ntab[i];
        v.count = sl >> 32;
        v.live = v.count != 0;
        v.key_or_pay = (uint32_t)sl;
    } else {
        const Slot s = src.table[i];
        v.live = s.key != KH_EMPTY_KEY;
        v.count = s.count;
        v.key_or_pay = s.key;
    }
    return v;
}
// The 32-bit window of hash bits below the region index (kernels.hip.h kh_below_region) of a slot of region r.  What
// depends on the region alone -- its bucket's first x -- is computed once per workgroup (one workgroup per region):
struct BelowCtx {
    uint32_t xlo, w;
};
__device__ __forceinline__ BelowCtx below_ctx(const SlotSrc &src, u64 r, uint32_t k) {
    BelowCtx b;
    b.xlo = kh_xlo_k((uint32_t)(r % src.geo.b2), src.geo.b2, kh_x_zero_bits(k, src.geo.p1_bits));
    b.w = kh_below_w(src.geo.b2);
    return b;
}
__device__ __forceinline__ uint32_t slot_hash_below_region(const SlotSrc &src, const SlotVal &v, const BelowCtx &bc, uint32_t k) {
    if (src.ntab) {  // the payload IS x, and no hash bits follow it (32-bit payloads: 2k - p1_bits <= 32)
        const uint32_t xoff = (uint32_t)v.key_or_pay - bc.xlo;
        return bc.w < 32 ? xoff << (32 - bc.w) : xoff;
    }
    const u64 H = kh_table_hash(v.key_or_pay, k);
    const uint32_t xoff = kh_x_of(H, src.geo.p1_bits) - bc.xlo;
    const uint32_t z = bc.w < 32 ? (uint32_t)((H << (src.geo.p1_bits + 32)) >> (32 + bc.w)) : 0u;
    return (bc.w < 32 ? xoff << (32 - bc.w) : xoff) | z;
}

// rcount[r] = live slots of region r.  One workgroup per region.
KH_GLOBAL __launch_bounds__(BLOCK) void region_live_count_kernel(SlotSrc src, uint32_t *__restrict__ rcount) {
    __shared__ uint32_t s_n;
    const u64 r = blockIdx.x;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    uint32_t n = 0;
    for (uint32_t i = threadIdx.x; i < REGION_SLOTS; i += BLOCK) n += slot_read(src, r * REGION_SLOTS + i).live;
    n = (uint32_t)wave_sum((u64)n);
    if (lane_id() == 0 && n) atomicAdd(&s_n, n);
    __syncthreads();
    if (threadIdx.x == 0) rcount[r] = s_n;
}

// Live pairs of region r go to [roff[r], roff[r+1]) (any order inside the region).
KH_GLOBAL __launch_bounds__(BLOCK) void region_compact_kernel(const Slot *__restrict__ table, const u64 *__restrict__ roff,
                                                               u64 *__restrict__ keys, u64 *__restrict__ counts) {
    __shared__ uint32_t s_cur;
    const u64 r = blockIdx.x;
    const u64 base = roff[r];
    if (roff[r + 1] == base) return;
    if (threadIdx.x == 0) s_cur = 0;
    __syncthreads();
    const Slot *reg = table + r * REGION_SLOTS;
    for (uint32_t i = threadIdx.x; i < REGION_SLOTS; i += BLOCK) {  // uniform trip count (4096 / 256)
        const Slot s = reg[i];
        const bool live = s.key != KH_EMPTY_KEY;
        const u64 m = kh_ballot(live);
        if (m == 0) continue;
        uint32_t wbase = 0;
        if ((int)lane_id() == __builtin_ctzll(m)) wbase = atomicAdd(&s_cur, (uint32_t)__builtin_popcountll(m));
        wbase = (uint32_t)__shfl((int)wbase, __builtin_ctzll(m), 64);
        if (live) {
            const u64 o = base + wbase + mbcnt(m);
            keys[o] = s.key;
            counts[o] = s.count;
        }
    }
}

// Packed form of the same export: ONE u64 per pair = count << 32 | bits [rbits, rbits + 32) of the
// table hash.  The receiver knows the region index of every segment, so these 32 bits identify the
// key whenever 2k - rbits <= 32 (the hash is a bijection); halves the bytes on the xGMI links.
// *wide is raised if a count does not fit 32 bits (the caller then uses the unpacked export).
KH_GLOBAL __launch_bounds__(BLOCK) void region_compact_packed_kernel(SlotSrc src, const u64 *__restrict__ roff,
                                                                      uint32_t k, u64 *__restrict__ pairs,
                                                                      u64 *__restrict__ wide) {
    __shared__ uint32_t s_cur;
    const u64 r = blockIdx.x;
    const u64 base = roff[r];
    if (roff[r + 1] == base) return;
    if (threadIdx.x == 0) s_cur = 0;
    __syncthreads();
    bool too_wide = false;
    const BelowCtx bc = below_ctx(src, r, k);
    for (uint32_t i = threadIdx.x; i < REGION_SLOTS; i += BLOCK) {
        const SlotVal s = slot_read(src, r * REGION_SLOTS + i);
        const bool live = s.live;
        const u64 m = kh_ballot(live);
        if (m == 0) continue;
        uint32_t wbase = 0;
        if ((int)lane_id() == __builtin_ctzll(m)) wbase = atomicAdd(&s_cur, (uint32_t)__builtin_popcountll(m));
        wbase = (uint32_t)__shfl((int)wbase, __builtin_ctzll(m), 64);
        if (live) {
            too_wide |= (s.count >> 32) != 0;
            pairs[base + wbase + mbcnt(m)] = (s.count << 32) | slot_hash_below_region(src, s, bc, k);
        }
    }
    if (kh_any(too_wide) && lane_id() == 0) atomicOr((unsigned long long *)wide, 1ull);
}

// ---- 32-bit "heads" -------------------------------------------------------------------------------
// The narrowest exchange unit: one u32 = [hb = kh_below_bits() hash bits (2k - log2 regions, rounded up) | cb = 32 - hb bits: addend - 1].
// A pair whose count exceeds 2^cb travels as several heads of the same key -- the receiver simply
// adds them up, so there is no escape mechanism -- and a table with a count above 64 x 2^cb is
// declared not representable (*wide), which bounds the blow-up.  S100M (k = 21, 2^19 regions):
// 23 hash bits + 9 count bits, 4 bytes per pair instead of 16.
__device__ __forceinline__ uint32_t heads_of(u64 count, uint32_t cb) { return (uint32_t)((count + (1ull << cb) - 1) >> cb); }

// `only` (optional): a byte per region -- regions whose byte is zero keep the count they have (batch.hip: the few regions the
// overflow list of a partitioned batch touched are counted again, the others' counts are the region pass's).
KH_GLOBAL __launch_bounds__(BLOCK) void region_head_count_kernel(SlotSrc src, uint32_t cb,
                                                                  uint32_t *__restrict__ rcount, u64 *__restrict__ wide,
                                                                  const uint8_t *__restrict__ only) {
    __shared__ uint32_t s_n;
    const u64 r = blockIdx.x;
    if (only && !only[r]) return;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    uint32_t n = 0;
    bool too_wide = false;
    for (uint32_t i = threadIdx.x; i < REGION_SLOTS; i += BLOCK) {
        const SlotVal s = slot_read(src, r * REGION_SLOTS + i);
        if (!s.live) continue;
        too_wide |= s.count > (64ull << cb);
        n += too_wide ? 1u : heads_of(s.count, cb);
    }
    n = (uint32_t)wave_sum((u64)n);
    if (lane_id() == 0 && n) atomicAdd(&s_n, n);
    if (kh_any(too_wide) && lane_id() == 0) atomicOr((unsigned long long *)wide, 1ull);
    __syncthreads();
    if (threadIdx.x == 0) rcount[r] = s_n;
}

// Heads of region r go to [roff[r], roff[r+1]) (any order inside the region).
KH_GLOBAL __launch_bounds__(BLOCK) void region_compact_heads_kernel(SlotSrc src, const u64 *__restrict__ roff,
                                                                     uint32_t k, uint32_t cb,
                                                                     uint32_t *__restrict__ heads) {
    __shared__ uint32_t s_cur;
    const u64 r = blockIdx.x;
    const u64 base = roff[r];
    if (roff[r + 1] == base) return;
    if (threadIdx.x == 0) s_cur = 0;
    __syncthreads();
    const uint32_t cmask = (1u << cb) - 1u;
    const BelowCtx bc = below_ctx(src, r, k);
    for (uint32_t i = threadIdx.x; i < REGION_SLOTS; i += BLOCK) {  // uniform trip count
        const SlotVal s = slot_read(src, r * REGION_SLOTS + i);
        const bool live = s.live;
        const uint32_t nh = live ? heads_of(s.count, cb) : 0u;
        // wave-inclusive prefix of nh
        uint32_t incl = nh;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if ((int)lane_id() >= off) incl += v;
        }
        const uint32_t wtotal = (uint32_t)__shfl((int)incl, 63, 64);
        if (wtotal == 0) continue;
        uint32_t wbase = 0;
        if (lane_id() == 63) wbase = atomicAdd(&s_cur, wtotal);
        wbase = (uint32_t)__shfl((int)wbase, 63, 64);
        if (live) {
            const uint32_t hw = slot_hash_below_region(src, s, bc, k) & ~cmask;  // the hb hash bits, top-aligned
            u64 o = base + wbase + (incl - nh);
            u64 left = s.count;
            for (uint32_t h = 0; h < nh; ++h) {
                const u64 take = left < (1ull << cb) ? left : (1ull << cb);
                heads[o + h] = hw | (uint32_t)(take - 1);
                left -= take;
            }
        }
    }
}

// ---- the same two compactions out of the 8-byte IMAGE (round 6) -----------------------------------------------------------
// What a rank exports right after its count is the image, and the loop above walks a region sixteen slots deep with a scan, an
// LDS atomic and a shuffle per step, one load in flight per lane: 1.9 ms per piece at configs[3]'s size, 3.6 TB/s.  Here a
// lane asks for its sixteen slots at once, ONE scan over the lanes' totals places them, the units are put together in LDS and
// leave in order as whole wave-wide stores; and the sender's digest of what it exports (exchange.hip: units, sum of counts,
// checksum per destination) is taken on the way -- rdig[2 r] = sum of the counts region r's units carry, rdig[2 r + 1] = the wrapping
// sum of its unit words -- instead of by a pass over the send buffer (export_digest_reduce_kernel folds them per owner).
template <bool HEADS>
__global__ __launch_bounds__(BLOCK) void region_compact_image_kernel(const u64 *__restrict__ ntab, RegionGeom geo, const u64 *__restrict__ roff, uint32_t k,
                                                                      uint32_t cb, void *__restrict__ out, u64 *__restrict__ rdig) {
    constexpr int SPL = REGION_SLOTS / BLOCK;
    typedef typename std::conditional<HEADS, uint32_t, u64>::type UT;
    __shared__ UT s_units[REGION_SLOTS];
    __shared__ uint32_t s_wsum[BLOCK / 64];
    __shared__ u64 s_dsum, s_dchk;
    const int tid = threadIdx.x;
    const u64 r = blockIdx.x;
    const u64 base = roff[r];
    const uint32_t total = (uint32_t)(roff[r + 1] - base);  // (<= 4096 x 64 units)
    if (total == 0) {  // (an empty region, or one outside the export's window)
        if (rdig && tid == 0) rdig[2 * r] = rdig[2 * r + 1] = 0;
        return;
    }
    u64 sl[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) sl[j] = ntab[r * REGION_SLOTS + (uint32_t)j * BLOCK + tid];
    if (tid == 0) s_dsum = s_dchk = 0;
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const u64 cnt = sl[j] >> 32;
        mine += cnt ? (HEADS ? heads_of(cnt, cb) : 1u) : 0u;
    }
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if ((int)lane_id() >= off) incl += v;
    }
    if (lane_id() == 63) s_wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t pos = incl - mine;
    for (int w = 0; w < (tid >> 6); ++w) pos += s_wsum[w];
    const uint32_t xlo = kh_xlo_k((uint32_t)(r % geo.b2), geo.b2, kh_x_zero_bits(k, geo.p1_bits)), w = kh_below_w(geo.b2);
    const uint32_t cmask = HEADS ? (1u << cb) - 1u : 0u;
    UT *const gout = reinterpret_cast<UT *>(out) + base;
    u64 dsum = 0, dchk = 0;
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        u64 left = sl[j] >> 32;
        if (!left) continue;
        const uint32_t xoff = (uint32_t)sl[j] - xlo;
        const uint32_t below = w < 32 ? xoff << (32 - w) : xoff;  // (the payload IS x: no hash bits follow it)
        if (HEADS) {
            const uint32_t hw = below & ~cmask;
            while (left) {
                const u64 take = left < (1ull << cb) ? left : (1ull << cb);
                const uint32_t h = hw | (uint32_t)(take - 1);
                if (pos < REGION_SLOTS) s_units[pos] = (UT)h;
                else gout[pos] = (UT)h;  // (more units than the staging holds: a region of many large counts)
                ++pos;
                dsum += take;
                dchk += h;
                left -= take;
            }
        } else {
            const u64 u = (left << 32) | below;
            s_units[pos++] = (UT)u;  // (one unit per live slot: always fits)
            dsum += left;
            dchk += u;
        }
    }
    if (rdig) {
        dsum = wave_sum(dsum);
        dchk = wave_sum(dchk);
        if (lane_id() == 0) {
            atomicAdd(&s_dsum, dsum);
            atomicAdd(&s_dchk, dchk);
        }
    }
    __syncthreads();
    const uint32_t staged = total < REGION_SLOTS ? total : REGION_SLOTS;
    for (uint32_t i = tid; i < staged; i += BLOCK) gout[i] = s_units[i];
    if (rdig && tid == 0) {
        rdig[2 * r] = s_dsum;
        rdig[2 * r + 1] = s_dchk;
    }
}
// out[3 p + {0, 1, 2}] += (units, sum of counts, checksum) of owner p's regions [p per, (p + 1) per); blockIdx.y = owner
KH_GLOBAL __launch_bounds__(BLOCK) void export_digest_reduce_kernel(const u64 *__restrict__ rdig, const u64 *__restrict__ roff, u64 per, u64 *__restrict__ out) {
    const u64 p = blockIdx.y, r0 = p * per;
    u64 sm = 0, ck = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < per; i += (u64)gridDim.x * BLOCK) {
        sm += rdig[2 * (r0 + i)];
        ck += rdig[2 * (r0 + i) + 1];
    }
    sm = wave_sum(sm);
    ck = wave_sum(ck);
    if (lane_id() == 0) {
        if (sm) atomicAdd(&out[3 * p + 1], sm);
        if (ck) atomicAdd(&out[3 * p + 2], ck);
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&out[3 * p], roff[r0 + per] - roff[r0]);
    }
}

// Region window of an export (kh_set_region_window): every owner's range of `per` regions is cut into
// pieces of `wper`; the counts of all pieces but `piece` become zero.
KH_GLOBAL __launch_bounds__(BLOCK) void region_window_mask_kernel(uint32_t *__restrict__ rcount, u64 nregions, u64 per, u64 wper,
                                                                   uint32_t piece) {
    const u64 stride = (u64)gridDim.x * BLOCK;
    for (u64 r = (u64)blockIdx.x * BLOCK + threadIdx.x; r < nregions; r += stride)
        if ((r % per) / wper != piece) rcount[r] = 0;
}

// ---- receiver side -----------------------------------------------------------------------------
constexpr int MAX_SENDERS = 64;
struct MergeSrc {
    const u64 *keys;    // the sender's pairs for this shard, ordered by the sender's region index
    const u64 *counts;
    const u64 *off;     // exclusive scan of the sender's per-region counts over this shard's region range (nr + 1)
};
struct MergeArgs {
    MergeSrc src[MAX_SENDERS];   // PACKED: keys = the packed pairs, counts unused
    uint32_t nsenders;
    int32_t dshift;  // target region t reads sender-local region t >> dshift (dshift >= 0), or the
                     // 2^-dshift sender-local regions starting at t << -dshift (dshift < 0)
    RegionGeom sgeo;      // packed formats: geometry of the senders' (unsharded) tables
    uint32_t head_cmask;  // heads: (1 << cb) - 1
    u64 src_region0;      // packed formats: the senders' region index of this shard's first region
};

// One workgroup per target region of the (sharded) receiver table.  FRESH: the table is empty.
// DIRECT: instead of rebuilding the region in LDS, upsert straight into HBM with device atomics --
// used only for the regions a first pass flagged as overflowing, after the table was grown.
// FMT: 0 = {u64 key, u64 count} in two arrays, 1 = packed u64 (count << 32 | 32 hash bits), 2 = u32 heads
template <bool FRESH, bool DIRECT, int FMT>
__global__ __launch_bounds__(1024, 8) void shard_merge_kernel(TableGeom tg, MergeArgs a, uint8_t *__restrict__ rfail,
                                                              uint32_t *__restrict__ rnew, u64 *__restrict__ radd,
                                                              const uint8_t *__restrict__ only_failed,
                                                              RegionGeom old_geo, Counters *ctr, uint32_t dirty,
                                                              uint32_t region0) {
    __shared__ u64 s_key[DIRECT ? 1 : REGION_SLOTS];
    __shared__ u64 s_cnt[DIRECT ? 1 : REGION_SLOTS];
    __shared__ uint32_t s_fail, s_new;
    __shared__ u64 s_add;  // sum of the counts this region took in (conservation: ctr->kmers, through radd / shard_reduce_kernel)
    __shared__ u64 s_seg_lo[MAX_SENDERS];
    __shared__ uint32_t s_seg_len[MAX_SENDERS];
    const int tid = threadIdx.x;
    // In DIRECT mode the grid still walks the ORIGINAL target regions (old_geo); tg is the grown table.
    const u64 t = (u64)blockIdx.x + region0;  // region0: first target region of this call's window (kh_set_region_window)
    if (DIRECT && !only_failed[t]) return;
    const RegionGeom match = DIRECT ? old_geo : rgeom(tg);  // the geometry whose region t this workgroup stands for
    auto target_of = [&](u64 H) -> u64 { return (u64)kh_p1_of(H, match.p1_bits) * match.b2 + kh_bucket_of_x(kh_x_of(H, match.p1_bits), match.b2); };
    Slot *reg = tg.table + t * REGION_SLOTS;
    if (!DIRECT) {
        const uint4 *g4 = reinterpret_cast<const uint4 *>(reg);
        for (uint32_t i = tid; i < REGION_SLOTS; i += 1024) {
            if (FRESH) {
                s_key[i] = KH_EMPTY_KEY;
                s_cnt[i] = 0;
            } else {
                const uint4 v = g4[i];
                s_key[i] = ((u64)v.y << 32) | v.x;
                s_cnt[i] = ((u64)v.w << 32) | v.z;
            }
        }
    }
    if (tid == 0) {
        s_fail = 0;
        s_new = 0;
        s_add = 0;
    }
    __syncthreads();
    uint32_t nd = 0, nf = 0;
    u64 nadd = 0;
    const u64 rl0 = a.dshift >= 0 ? (t >> a.dshift) : (t << -a.dshift);
    const u64 nrl = a.dshift >= 0 ? 1 : (1ull << -a.dshift);
    constexpr bool PACKED = FMT != 0;
    // What a sender region stands for in the packed formats: its level-1 digit at the top of the hash and the first x of
    // its bucket (kernels.hip.h kh_hash_of_below, with the divisions done once per segment)
    struct SegBase {
        u64 htop;
        uint32_t xlo;
    };
    const uint32_t sw = kh_below_w(a.sgeo.b2);
    auto seg_base = [&](u64 rs) -> SegBase {  // rs: the senders' (global) region index
        SegBase sb;
        const uint32_t p1 = (uint32_t)(rs / a.sgeo.b2), b = (uint32_t)(rs % a.sgeo.b2);
        sb.htop = a.sgeo.p1_bits ? (u64)p1 << (64 - a.sgeo.p1_bits) : 0ull;
        sb.xlo = kh_xlo_k(b, a.sgeo.b2, kh_x_zero_bits(tg.k, a.sgeo.p1_bits));
        return sb;
    };
    // one incoming unit: raw0 = key / packed pair / head, raw1 = count (FMT 0 only)
    auto take = [&](u64 raw0, u64 raw1, const SegBase &sb) {
        u64 key, H, addend;
        if (PACKED) {
            const uint32_t low = FMT == 1 ? (uint32_t)raw0 : ((uint32_t)raw0 & ~a.head_cmask);
            addend = FMT == 1 ? (raw0 >> 32) : (u64)((uint32_t)raw0 & a.head_cmask) + 1;
            const uint32_t x = sb.xlo + (sw < 32 ? low >> (32 - sw) : low);
            const u64 below = ((u64)x << 32) | (sw < 32 ? (u64)(uint32_t)(low << sw) : 0ull);
            const u64 Hs = sb.htop | (below >> a.sgeo.p1_bits);  // the sender's (unsharded) table hash
            H = Hs << tg.shard_shift;
            if (target_of(H) != t) return;  // the segment also feeds the sibling targets
            key = kh_table_unhash(Hs, tg.k);
        } else {
            key = raw0;
            H = table_hash(tg, key);
            if (target_of(H) != t) return;
            addend = raw1;
        }
        nadd += addend;
        if (DIRECT) {
            upsert(tg, key, addend, nd, nf);
            return;
        }
        uint32_t off = start_of(tg, H);
        uint32_t probes = 0;
        for (; probes < REGION_SLOTS; ++probes) {
            u64 cur = s_key[off];
            if (cur == KH_EMPTY_KEY) {
                cur = atomicCAS(&s_key[off], (u64)KH_EMPTY_KEY, key);
                if (cur == KH_EMPTY_KEY) {
                    ++nd;
                    cur = key;
                }
            }
            if (cur == key) {
                atomicAdd(&s_cnt[off], addend);
                break;
            }
            off = (off + 1) & REGION_MASK;
        }
        if (probes == REGION_SLOTS) s_fail = 1;
    };
    auto load0 = [&](const MergeSrc &src, u64 i) -> u64 {
        return FMT == 2 ? (u64)reinterpret_cast<const uint32_t *>(src.keys)[i] : src.keys[i];
    };
    if (nrl == 1) {
        // The usual shape (receiver table at least as large as the senders'): every sender contributes ONE
        // segment.  Walking the senders one after the other put, per sender, two dependent offset
        // loads and a unit load in series (~25 us per region).  So: all offsets first, then the units
        // of four senders in flight at a time (branch-free clamped loads).
        if (tid < (int)a.nsenders) {
            const u64 lo = a.src[tid].off[rl0];
            s_seg_lo[tid] = lo;
            s_seg_len[tid] = (uint32_t)(a.src[tid].off[rl0 + 1] - lo);  // a region holds <= 4096 keys x <= 64 heads
        }
        __syncthreads();
        const SegBase sb = PACKED ? seg_base(a.src_region0 + rl0) : SegBase{0ull, 0u};
        for (uint32_t s0 = 0; s0 < a.nsenders; s0 += 4) {
            uint32_t len[4], maxlen = 0;
            u64 lo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool on = s0 + q < a.nsenders;
                len[q] = on ? s_seg_len[s0 + q] : 0u;
                lo[q] = on ? s_seg_lo[s0 + q] : 0ull;
                maxlen = len[q] > maxlen ? len[q] : maxlen;
            }
            for (uint32_t base = 0; base < maxlen; base += 1024) {
                u64 r0[4], r1[4];
                const uint32_t idx = base + tid;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    r0[q] = r1[q] = 0;
                    if (len[q]) {  // uniform
                        const MergeSrc &src = a.src[(s0 + q) < a.nsenders ? (s0 + q) : 0];
                        const u64 i = lo[q] + (idx < len[q] ? idx : len[q] - 1);
                        r0[q] = load0(src, i);
                        if (FMT == 0) r1[q] = src.counts[i];
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (idx < len[q]) take(r0[q], r1[q], sb);
            }
        }
    } else {
        for (uint32_t s = 0; s < a.nsenders; ++s)
            for (u64 rl = rl0; rl < rl0 + (PACKED ? nrl : 1); ++rl) {  // PACKED: segment by segment (the region index is part of the key)
                const MergeSrc src = a.src[s];
                const u64 lo = src.off[rl], hi = src.off[PACKED ? rl + 1 : rl0 + nrl];
                const SegBase sb = PACKED ? seg_base(a.src_region0 + rl) : SegBase{0ull, 0u};
                for (u64 i = lo + tid; i < hi; i += 1024) take(load0(src, i), FMT == 0 ? src.counts[i] : 0ull, sb);
            }
    }
    if (DIRECT) {
        const u64 d = wave_sum((u64)nd), f = wave_sum((u64)nf), ad = wave_sum(nadd);
        if (lane_id() == 0) {
            if (d) atomicAdd(&ctr->distinct, d);
            if (f) atomicAdd(&ctr->failed, f);
            if (ad) atomicAdd(&ctr->kmers, ad);
        }
        return;
    }
    const uint32_t dw = (uint32_t)wave_sum((u64)nd);
    const u64 aw = wave_sum(nadd);
    if ((tid & 63) == 0 && dw) atomicAdd(&s_new, dw);
    if ((tid & 63) == 0 && aw) atomicAdd(&s_add, aw);
    __syncthreads();
    if (s_fail) {
        if (FRESH && dirty)  // lazily reset table (kernels.hip.h / kh_reset): leave an EMPTY region, not garbage
            for (uint32_t i = tid; i < REGION_SLOTS; i += 1024) reinterpret_cast<uint4 *>(reg)[i] = make_uint4(~0u, ~0u, 0u, 0u);
        if (tid == 0) {
            rfail[t] = 1;
            rnew[t] = 0;
            radd[t] = 0;  // (the DIRECT pass takes this region's units again, and counts them then)
        }
        return;
    }
    uint4 *o4 = reinterpret_cast<uint4 *>(reg);
    for (uint32_t i = tid; i < REGION_SLOTS; i += 1024) {
        const u64 kk = s_key[i], cc = s_cnt[i];
        o4[i] = make_uint4((uint32_t)kk, (uint32_t)(kk >> 32), (uint32_t)cc, (uint32_t)(cc >> 32));
    }
    if (tid == 0) {
        rfail[t] = 0;
        rnew[t] = s_new;
        radd[t] = s_add;
    }
}

// ---- the shard as an 8-byte image (round 6) ------------------------------------------------------------
// A FRESH merge of packed pairs or heads can leave the shard in the form a partitioned count leaves its table in: one
// u64 per slot, count << 32 | x, x = the 32 hash bits behind the shard table's level-1 digit (partition.hip.h: the 8-byte
// image; kh_finish / kh_result_* / kh_histogram / kh_lookup read it as it is, ensure_wide() converts) -- whenever those 32 bits
// hold everything the region index does not: 2k - shard_shift - p1_bits <= 32.  Round 5's kernel above rebuilt 16-byte slots:
// 43 GB of writes at configs[3]'s size behind a 64 KiB LDS image with 64-bit compare-and-swaps and an inverse hash per unit,
// 4.7 ms per piece of a four-piece merge.  Here: a 32 KiB LDS image (payload and count words, 32-bit claims), 512 lanes per
// workgroup, no inverse hash -- a unit's x is a few shifts of its own bits --, 8 bytes per slot written.
//   * a unit whose x equals the free marker (0xFFFFFFFF: possible only when all 32 bits are significant) is summed apart and
//     placed by one lane at the end (as region_count_kernel32 does);
//   * counts are 32-bit in this image: a target region that took in 2^32 occurrences or more fails with code 2 (a region that is
//     full: code 1); the host widens, grows and re-inserts those regions' units through the direct path, as for the wide form.
// WHAT A REGION COSTS IS INSTRUCTIONS.  The first version (one workgroup per region, the general arithmetic of the kernel above)
// took 3.1 ms per piece at configs[3]'s size against 1.4 ms for its bytes: SQ counters, 800 vector + 1000 scalar instructions
// per wave and region -- uniform 64-bit divisions (seg_base, kh_xlo, t / b2) done over by every wave for every region, 64-bit
// shifts per unit -- on SIMDs that issue one instruction per four cycles whatever its kind.  So:
//   * the geometry is the host's: NarrowK holds every shift, mask and magic number; no division on the device;
//   * ONE WORKGROUP WALKS 2^gshift CONSECUTIVE TARGETS: their (digit, bucket) and the sender region's (digit, bucket, first x) move on
//     by increments, all their segment bounds are fetched at once, and the units of target g + 1 are asked for before target g's
//     write-back.  At W ranks with equal tables the W siblings that read the same sender segments are consecutive targets: one
//     workgroup's, which finds the segments in the L1 / L2 from the second sibling on;
//   * a unit's place in this shard is 32-bit work: xs = the sender bucket's first x + the unit's offset, x = xs << o, its level-1
//     digit = (sender digit : xs) bits, o = shard_shift + p1_bits - the senders' p1_bits.  That needs the receiver's digit to end
//     inside or at the senders' x (0 <= o < 32), no hash bits behind the senders' x (2k - their p1_bits <= 32: what an image
//     exports), targets no coarser than the senders' regions -- every table of a real exchange; merge.hip sends the other
//     geometries to the kernel above.
// rdig (optional): [workgroup][sender][3] -- units this workgroup's targets took from that sender, the sum of their counts, the wrapping
// sum of their raw words: the arrival digest of exchange.hip (what unit_digest_kernel computes in a pass of its own), as partial
// sums per workgroup; every unit is taken by exactly one target, so their sum over the grid is the digest of what arrived.
constexpr int SHARD_NT = 512;
constexpr uint32_t SHARD_SEGS = 64;  // (targets per workgroup) x senders <= this: the bounds kept in LDS
struct NarrowK {
    uint32_t nsenders, pshift, gshift, dshift;  // P = 2^pshift waves per sender; 2^gshift targets per workgroup; target t reads sender-local region t >> dshift
    uint32_t b2r, b2s;                          // buckets per level-1 partition: the shard table's, the senders' tables'
    u64 magic_r, magic_s;                       // ceil(2^40 / b2): r / b2 == (r * magic) >> 40 for r < 2^22
    uint32_t sw_shr;                            // a unit's x offset = its window >> this ((32 - kh_below_w(b2s)) & 31)
    uint32_t o, o_shr, omask, rmask;            // see above; o_shr = (32 - o) & 31, omask = 2^o - 1, rmask = 2^p1_bits - 1
    uint32_t cmask;                             // heads: 2^cb - 1 (the count field); packed pairs: 0
    uint32_t zs_mask;                           // 2^zs - 1: the low bits of every x a k-mer's hash leaves zero (kh_xlo_k)
    uint32_t stepR;                             // 2^32 mod b2s ...
    u64 stepQ;                                  // ... and 2^32 / b2s: one step of the senders' bucket -> first x
    u64 src_region0;                            // the senders' region index of this shard's first region
    uint32_t region0;                           // first target of this launch (kh_set_region_window)
};
template <int FMT>
__global__ __launch_bounds__(SHARD_NT, 6) void shard_merge_narrow_kernel(NarrowK K, MergeArgs a, u64 *__restrict__ ntab, uint8_t *__restrict__ rfail,
                                                                          uint32_t *__restrict__ rnew, u64 *__restrict__ radd, u64 *__restrict__ rdig) {
    static_assert(FMT == 1 || FMT == 2, "packed pairs or heads");
    constexpr int NT = SHARD_NT;
    constexpr uint32_t FREE = 0xFFFFFFFFu, NW = NT / 64, DEPTH = 4;
    typedef typename std::conditional<FMT == 2, uint32_t, u64>::type UT;  // one exchange unit
    __shared__ __attribute__((aligned(16))) uint32_t s_pay[REGION_SLOTS];
    __shared__ __attribute__((aligned(16))) uint32_t s_add[REGION_SLOTS];
    __shared__ uint32_t s_fail, s_new, s_special, s_sp_off;
    __shared__ u64 s_sum;
    __shared__ u64 s_seg_lo[SHARD_SEGS];
    __shared__ uint32_t s_seg_len[SHARD_SEGS];
    __shared__ u64 s_dig[MAX_SENDERS * 3];
    const int tid = threadIdx.x;
    const uint32_t wave = (uint32_t)tid >> 6, lane = (uint32_t)tid & 63u;
    const uint32_t G = 1u << K.gshift, P = 1u << K.pshift, nitems = K.nsenders << K.pshift;
    u64 t = ((u64)blockIdx.x << K.gshift) + K.region0;  // the workgroup's first target
    if ((uint32_t)tid < G * K.nsenders) {  // all the bounds at once (host: G x senders <= SHARD_SEGS)
        const uint32_t g = (uint32_t)tid / K.nsenders, s = (uint32_t)tid - g * K.nsenders;
        const u64 rl = (t + g) >> K.dshift;
        const u64 lo = a.src[s].off[rl];
        s_seg_lo[tid] = lo;
        s_seg_len[tid] = (uint32_t)(a.src[s].off[rl + 1] - lo);  // a region holds <= 4096 keys x <= 64 heads
    }
    if (rdig && tid < 3 * (int)K.nsenders) s_dig[tid] = 0;
    // the first target's (digit, bucket); the sender region it reads: (digit, bucket) and floor / remainder of (bucket << 32) / b2s
    uint32_t tp1 = (uint32_t)((t * K.magic_r) >> 40), tb = (uint32_t)t - tp1 * K.b2r;
    u64 rl_have = t >> K.dshift;
    uint32_t sp1, sbk, sr;
    u64 sq;
    {
        const u64 rs = K.src_region0 + rl_have;
        sp1 = (uint32_t)((rs * K.magic_s) >> 40);
        sbk = (uint32_t)rs - sp1 * K.b2s;
        const uint32_t v = sbk * K.stepR, vq = (uint32_t)(((u64)v * K.magic_s) >> 40);  // (host: b2s <= 1024, so v < 2^20)
        sq = (u64)sbk * K.stepQ + vq;
        sr = v - vq * K.b2s;
    }
    // this wave's item: sender s (of P waves), every P-th group of 64 units.  The first DEPTH units of a target are asked for a
    // target ahead (branch-free: clamped index; a wave without an item, or an empty segment, reads unit 0 of sender 0 and ignores it)
    const bool has_item = wave < nitems;
    const uint32_t s0 = has_item ? wave >> K.pshift : 0u, part0 = has_item ? wave - (s0 << K.pshift) : 0u;
    UT pre[DEPTH];
    auto prefetch = [&](uint32_t g) {
        const uint32_t len = has_item ? s_seg_len[g * K.nsenders + s0] : 0u;
        const u64 lo = has_item ? s_seg_lo[g * K.nsenders + s0] : 0ull;
        const UT *const keys = reinterpret_cast<const UT *>(a.src[s0].keys);
#pragma unroll
        for (uint32_t q = 0; q < DEPTH; ++q) {
            const uint32_t idx = ((part0 + q * P) << 6) + lane;
            pre[q] = keys[len ? lo + (idx < len ? idx : len - 1) : 0ull];
        }
    };
    __syncthreads();  // (the bounds)
    prefetch(0);
    uint32_t dg_n = 0, dg_s = s0;  // digest of the units this lane took from sender dg_s, over all of the workgroup's targets
    u64 dg_sum = 0, dg_chk = 0;
    auto flush_digest = [&]() {  // -> that sender's digest words (LDS; three wave sums)
        const u64 n = wave_sum((u64)dg_n), sm = wave_sum(dg_sum), ck = wave_sum(dg_chk);
        if (lane == 0) {
            if (n) atomicAdd(&s_dig[3 * dg_s], n);
            if (sm) atomicAdd(&s_dig[3 * dg_s + 1], sm);
            if (ck) atomicAdd(&s_dig[3 * dg_s + 2], ck);
        }
        dg_n = 0;
        dg_sum = dg_chk = 0;
    };
    for (uint32_t g = 0; g < G; ++g, ++t) {
        {
            uint4 *p4 = reinterpret_cast<uint4 *>(s_pay), *a4 = reinterpret_cast<uint4 *>(s_add);
#pragma unroll
            for (uint32_t i = 0; i < REGION_SLOTS / 4 / NT; ++i) {
                p4[i * NT + tid] = make_uint4(FREE, FREE, FREE, FREE);
                a4[i * NT + tid] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        if (tid == 0) {
            s_fail = 0;
            s_new = 0;
            s_special = 0;
            s_sp_off = FREE;
            s_sum = 0;
        }
        if ((t >> K.dshift) != rl_have) {  // the next sender region (consecutive targets: its successor)
            ++rl_have;
            if (++sbk == K.b2s) {
                sbk = 0;
                ++sp1;
                sq = 0;
                sr = 0;
            } else {
                sq += K.stepQ;
                sr += K.stepR;
                if (sr >= K.b2s) {
                    sr -= K.b2s;
                    ++sq;
                }
            }
        }
        const uint32_t sxlo = ((uint32_t)sq + (sr != 0 ? 1u : 0u) + K.zs_mask) & ~K.zs_mask;  // kh_xlo_k of the sender region's bucket
        const uint32_t sp1o = (sp1 << K.o) & K.rmask;  // the sender digit's share of a unit's digit in this shard
        __syncthreads();
        uint32_t nd = 0, nadd_lo = 0, nadd_hi = 0;
        auto take = [&](UT raw) {
            const uint32_t low = FMT == 1 ? (uint32_t)raw : ((uint32_t)raw & ~K.cmask);
            const uint32_t addend = FMT == 1 ? (uint32_t)((u64)raw >> 32) : ((uint32_t)raw & K.cmask) + 1u;
            const uint32_t xs = sxlo + (low >> K.sw_shr);
            const uint32_t x = xs << K.o;
            const uint32_t dgt = sp1o | ((xs >> K.o_shr) & K.omask);
            const u64 prod = (u64)x * K.b2r;  // high word: the bucket; low word: the in-region start (kernels.hip.h)
            if (dgt != tp1 || (uint32_t)(prod >> 32) != tb) return;  // the segment also feeds the sibling targets
            ++dg_n;
            dg_sum += addend;
            dg_chk += (u64)raw;
            if (addend == 0) return;  // (a packed pair without a count carries nothing; 0 would read as a free slot)
            const uint32_t before = nadd_lo;
            nadd_lo += addend;
            nadd_hi += nadd_lo < before ? 1u : 0u;
            if (x == FREE) {
                atomicAdd(&s_special, addend);
                return;
            }
            uint32_t off = ((uint32_t)prod >> (32 - REGION_BITS)) & REGION_START_MASK;
            const uint32_t off0 = off;
            for (;;) {
                uint32_t cur = s_pay[off];
                if (cur == FREE) {
                    cur = atomicCAS(&s_pay[off], FREE, x);
                    if (cur == FREE) {
                        ++nd;
                        cur = x;
                    }
                }
                if (cur == x) {
                    atomicAdd(&s_add[off], addend);  // (wraps only where the region's total reaches 2^32: caught below)
                    break;
                }
                off = (off + 1) & REGION_MASK;
                if (off == off0) {  // every slot seen: region full
                    s_fail = 1;
                    break;
                }
            }
        };
        for (uint32_t it = wave; it < nitems; it += NW) {
            const uint32_t s = it >> K.pshift, part = it - (s << K.pshift);
            if (rdig && s != dg_s) {  // (uniform: a wave with several senders hands the last one's digest in before it goes on)
                flush_digest();
                dg_s = s;
            }
            const uint32_t len = s_seg_len[g * K.nsenders + s];
            const u64 lo = s_seg_lo[g * K.nsenders + s];
            const UT *const keys = reinterpret_cast<const UT *>(a.src[s].keys);
            for (uint32_t base = part << 6; base < len; base += (DEPTH << 6) << K.pshift) {
                if (it != wave || base != (part << 6)) {  // (uniform) every round but the one asked for a target ago
#pragma unroll
                    for (uint32_t q = 0; q < DEPTH; ++q) {
                        const uint32_t idx = base + ((q * P) << 6) + lane;
                        pre[q] = keys[lo + (idx < len ? idx : len - 1)];
                    }
                }
#pragma unroll
                for (uint32_t q = 0; q < DEPTH; ++q)
                    if (base + ((q * P) << 6) + lane < len) take(pre[q]);
            }
        }
        if (g + 1 < G) prefetch(g + 1);  // (in flight during the barriers, the write-back and the next target's set-up)
        const uint32_t dw = (uint32_t)wave_sum((u64)nd);
        const u64 aw = wave_sum(((u64)nadd_hi << 32) | nadd_lo);
        if (lane == 0 && dw) atomicAdd(&s_new, dw);
        if (lane == 0 && aw) atomicAdd(&s_sum, aw);
        __syncthreads();
        if (s_special) {  // (uniform; almost never)
            if (tid == 0 && !s_fail) {
                uint32_t off = kh_start_of_x(FREE, K.b2r), probes = 0;
                for (; probes < REGION_SLOTS && s_pay[off] != FREE; ++probes) off = (off + 1) & REGION_MASK;
                if (probes == REGION_SLOTS) s_fail = 1;
                else {
                    s_sp_off = off;
                    s_new += 1;
                }
            }
            __syncthreads();
        }
        if (tid == 0 && !s_fail && s_sum >= 0xFFFFFFFFull) s_fail = 2;  // a 32-bit count may have wrapped
        __syncthreads();
        uint4 *o4 = reinterpret_cast<uint4 *>(ntab + t * REGION_SLOTS);
        if (s_fail) {  // an EMPTY region of the image (it held another table's slots), the units again through the direct path
            for (uint32_t i = tid; i < REGION_SLOTS / 2; i += NT) o4[i] = make_uint4(0u, 0u, 0u, 0u);
            if (tid == 0) {
                rfail[t] = (uint8_t)s_fail;
                rnew[t] = 0;
                radd[t] = 0;
            }
        } else {
            const uint32_t sp_off = s_sp_off, sp_cnt = s_special;
#pragma unroll
            for (uint32_t j = 0; j < REGION_SLOTS / 2 / NT; ++j) {  // two slots per lane: 8-byte LDS reads, one 16-byte store
                const uint32_t i = j * NT + tid;
                const uint2 pp = reinterpret_cast<const uint2 *>(s_pay)[i], cc = reinterpret_cast<const uint2 *>(s_add)[i];
                uint32_t c0 = cc.x, c1 = cc.y;
                if (2 * i == sp_off) c0 = sp_cnt;      // (its payload word is the free marker already: that IS its payload)
                if (2 * i + 1 == sp_off) c1 = sp_cnt;
                o4[i] = make_uint4(c0 ? pp.x : 0u, c0, c1 ? pp.y : 0u, c1);
            }
            if (tid == 0) {
                rfail[t] = 0;
                rnew[t] = s_new;
                radd[t] = s_sum;
            }
        }
        if (++tb == K.b2r) {  // the next target's (level-1 digit, bucket)
            tb = 0;
            ++tp1;
        }
        __syncthreads();  // (the image and the flags are set up again from here)
    }
    if (rdig) {  // (a failed region's units too: they did arrive)
        flush_digest();
        __syncthreads();
        if (tid < 3 * (int)K.nsenders) rdig[(u64)blockIdx.x * 3 * K.nsenders + tid] = s_dig[tid];
    }
}

// out[c] += sum over the targets of rdig[target][c], c < cols = 3 x senders.  blockDim.x is a multiple of cols, so a lane stays in one
// column however far it strides.
KH_GLOBAL void digest_reduce_kernel(const u64 *__restrict__ rdig, u64 ntargets, uint32_t cols, u64 *__restrict__ out) {
    __shared__ u64 s_col[3 * MAX_SENDERS];
    const u64 total = ntargets * cols, stride = (u64)gridDim.x * blockDim.x;
    if (threadIdx.x < cols) s_col[threadIdx.x] = 0;
    __syncthreads();
    u64 acc = 0;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) acc += rdig[e];
    if (acc) atomicAdd(&s_col[threadIdx.x % cols], acc);
    __syncthreads();
    if (threadIdx.x < cols && s_col[threadIdx.x]) atomicAdd(&out[threadIdx.x], s_col[threadIdx.x]);
}

// distinct += sum(rnew), kmers += sum(radd), part_failed += number of failed target regions
KH_GLOBAL __launch_bounds__(BLOCK) void shard_reduce_kernel(const uint8_t *__restrict__ rfail, const uint32_t *__restrict__ rnew,
                                                             const u64 *__restrict__ radd, u64 nregions, Counters *ctr) {
    const u64 stride = (u64)gridDim.x * BLOCK;
    u64 d = 0, nf = 0, ad = 0;
    for (u64 r = (u64)blockIdx.x * BLOCK + threadIdx.x; r < nregions; r += stride) {
        if (rfail[r]) ++nf;
        else {
            d += rnew[r];
            ad += radd[r];
        }
    }
    d = wave_sum(d);
    nf = wave_sum(nf);
    ad = wave_sum(ad);
    if (lane_id() == 0) {
        if (d) atomicAdd(&ctr->distinct, d);
        if (nf) atomicAdd(&ctr->part_failed, nf);
        if (ad) atomicAdd(&ctr->kmers, ad);
    }
}

// ---- conservation digests of exchange units (round 5; exchange.hip) --------------------------------
// What a sender announces about every segment it sends, and what the receiver recomputes from the bytes that arrived:
// the number of units, the sum of the counts they carry (the same arithmetic as shard_merge_kernel's take()), and a wrapping
// sum of the raw unit words.  blockIdx.y = segment; out[3 * seg + {0, 1, 2}] is ADDED to (zeroed by the host).
// FMT as in shard_merge_kernel: 0 = keys + counts arrays, 1 = packed u64, 2 = u32 heads.
struct DigestSegs {
    u64 off[MAX_SENDERS];  // first unit of the segment (in units from `base`)
    u64 len[MAX_SENDERS];  // units
};
template <int FMT>
__global__ __launch_bounds__(BLOCK) void unit_digest_kernel(const void *__restrict__ base, const u64 *__restrict__ counts, DigestSegs segs,
                                                            uint32_t head_cmask, u64 *__restrict__ out) {
    const uint32_t seg = blockIdx.y;
    const u64 lo = segs.off[seg], n = segs.len[seg];
    const u64 stride = (u64)gridDim.x * BLOCK, t0 = (u64)blockIdx.x * BLOCK + threadIdx.x;
    u64 sum = 0, chk = 0;
    if (FMT == 2) {
        // 16-byte loads over the aligned middle of the segment (a first version read one head per lane: 2 TB/s, 5.8 ms of a
        // 39 ms merge leg at configs[3]'s size), single heads in front of it and behind
        const uint32_t *p = reinterpret_cast<const uint32_t *>(base) + lo;
        auto one = [&](uint32_t h) {
            sum += (u64)(h & head_cmask) + 1;
            chk += h;
        };
        const u64 pre = n < 4 ? n : (((16 - ((uintptr_t)p & 15)) & 15) >> 2);
        const u64 nvec = (n - pre) >> 2, rest0 = pre + (nvec << 2);
        if (t0 < pre) one(p[t0]);
        const uint4 *v = reinterpret_cast<const uint4 *>(p + pre);
        for (u64 i = t0; i < nvec; i += stride) {
            const uint4 x = v[i];
            one(x.x);
            one(x.y);
            one(x.z);
            one(x.w);
        }
        if (t0 < n - rest0) one(p[rest0 + t0]);
    } else if (FMT == 1) {
        const u64 *p = reinterpret_cast<const u64 *>(base) + lo;
        auto one = [&](u64 v) {
            sum += v >> 32;
            chk += v;
        };
        const u64 pre = n < 2 ? n : (((uintptr_t)p & 8) ? 1 : 0);
        const u64 nvec = (n - pre) >> 1, rest0 = pre + (nvec << 1);
        if (t0 < pre) one(p[t0]);
        const uint4 *v = reinterpret_cast<const uint4 *>(p + pre);
        for (u64 i = t0; i < nvec; i += stride) {
            const uint4 x = v[i];
            one(((u64)x.y << 32) | x.x);
            one(((u64)x.w << 32) | x.z);
        }
        if (t0 < n - rest0) one(p[rest0 + t0]);
    } else {
        for (u64 i = t0; i < n; i += stride) {
            const u64 key = reinterpret_cast<const u64 *>(base)[lo + i], cn = counts[lo + i];
            sum += cn;
            chk += key + 0x9E3779B97F4A7C15ull * cn;
        }
    }
    sum = wave_sum(sum);
    chk = wave_sum(chk);
    if (lane_id() == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && n) atomicAdd(&out[3 * seg + 0], n);
        if (sum) atomicAdd(&out[3 * seg + 1], sum);
        if (chk) atomicAdd(&out[3 * seg + 2], chk);
    }
}

}  // namespace kh
