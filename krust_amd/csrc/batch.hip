// batch.hip -- counting one device-resident range (count_device_range): the direct path, and the partitioned batch -- level 1,
// the table sized from its sample, level 2, the region pass (reference src/run.rs:494-582 is the loop this replaces).
#include "ctx.hip.h"
#include "level1_api.h"

namespace khi {

template <bool QUAL>
void launch_count(kh_ctx *c, const uint8_t *abase, const uint8_t *qbase, int qaligned, u64 vbeg, u64 vend,
                  u64 wlo, u64 tile0, u64 ntiles) {
    // contiguous tile ranges per workgroup so the k-1 look-back is carried in LDS
    u64 blocks = ntiles < (u64)GRID_CAP ? ntiles : (u64)GRID_CAP;
    uint32_t tpb = (uint32_t)((ntiles + blocks - 1) / blocks);
    blocks = (ntiles + tpb - 1) / tpb;
    uint32_t thr = 0;
    if (QUAL) {
        int t = c->minq + 33;  // saturating_add(33) on u8, run.rs:538
        thr = (uint32_t)(t > 255 ? 255 : t);
    }
    hipLaunchKernelGGL(kh::count_direct_kernel<QUAL>, dim3((unsigned)blocks), dim3(kh::BLOCK), 0, c->stream, abase,
                       qbase, qaligned, vbeg, vend, wlo, tile0, ntiles, tpb, c->k, thr, table_geom(c, c->table, c->cap), c->d_ctr);
}

struct RangeArgs {
    const uint8_t *abase, *qbase;
    int qaligned;
    bool use_qual;
    u64 vbeg, vend, wlo;
    // share of the range's windows expected to survive masking (1 = size the partition buffers for every window);
    // below 1 only for quality-masked ranges, from survival_sample_kernel -- see sized_for()
    double survive = 1.0;
};

// Internal result of partition_batch: the level-1 pool, sized from RangeArgs::survive, ran out -- nothing but the pool
// was written; the caller runs the same tiles again sized for every window.
constexpr int KH_RETRY_FULL_SIZE = 1000;
// ... and: the batch was sized on the assumption that level 2 narrows its 8-byte payloads (15 instead of 20 bytes per window), it
// cannot after all, and the 8-byte level-2 output does not fit beside the pool: the caller runs the same tiles again in smaller batches.
constexpr int KH_RETRY_WIDE = 1001;

// payloads to make room for when at most n windows exist and a share `survive` of them is expected to be countable:
// an eighth over the estimate plus a 64th of the windows (the sample is a 64th of the tiles)
u64 sized_for(u64 n, double survive) {
    if (survive >= 1.0) return n;
    const double e = (double)n * (survive * 1.125 + 1.0 / 64) + 65536.0;
    return e >= (double)n ? n : (u64)e;
}

uint32_t qual_thr(const kh_ctx *c) {
    int t = c->minq + 33;  // saturating_add(33) on u8, run.rs:538
    return (uint32_t)(t > 255 ? 255 : t);
}

// count bits of a 32-bit exchange head for this table (shard.hip.h), or -1 if the format does not apply
template <typename PT>
void launch_region(kh_ctx *c, const kh::PartGeom &g, u64 nregions, u64 hot, const u64 *bend, bool narrow, u64 skip, u64 expect);
// (fresh passes that will end at load <= 0.6: 512-lane workgroups, three per CU -- see launch_region<uint32_t> below)
bool region_small_groups(const kh_ctx *c, u64 expect, u64 nregions) {
    // 1024 lanes only where the probing loop is most of the kernel AND has the payloads to fill them: a table that ends
    // above load 0.6 (the hint's load, or -- without one -- as if every payload room was made for were a new key) with
    // more than 16 K payloads per bucket.  Measured: 125 M reads into 2^31 slots (0.61, 30 K per bucket) 36.4 vs 40.9 ms
    // with 512 lanes; an hg-shaped input in 2^32 slots (0.62, 2.9 K per bucket) 27.4 vs 20.5 ms.
    const int forced = c->knobs.region_nt;
    if (forced) return forced == 512;
    const double keys = c->est_keys ? (double)(c->distinct_known + c->est_keys) : c->hinted ? (double)c->hint_keys : (double)(c->distinct_known + expect);
    return !(keys > 0.6 * (double)c->cap && expect / nregions > 16384);
}
// The fresh region pass can leave every region's exchange-head count behind (a multi-GPU export right after it then skips its
// counting pass): worth its instructions -- a shift, an add and a compare per slot, a wave sum and a barrier per region --
// only where an export will come: a context with a communicator (kh_comm_init / kh_group_create come before the pushes), or
// one whose last table was exported (kh_export_regions_*: the Python harness, the tests).  Round 5, same-box A/B of a context without one: region pass 16.55 -> 16.37 ms at
// the headline, 16.94 -> 16.52 for the hg-shaped input.
static bool want_heads(const kh_ctx *c) { return c->comm != nullptr || c->exports_seen || c->knobs.heads_always; }
template <>
void launch_region<u64>(kh_ctx *c, const kh::PartGeom &g, u64 nregions, u64 hot, const u64 *bend, bool, u64 skip, u64 expect) {
    const kh::TableGeom tg = table_geom(c, c->table, c->cap);
#define KH_REGION64(FRESH, NT) \
    hipLaunchKernelGGL((kh::region_count_kernel64<FRESH, NT>), dim3((unsigned)nregions), dim3(NT), 0, c->stream, tg, g, \
                       (const u64 *)c->keysB, bend, (const u64 *)c->bstart, c->rfail, c->rnew, hot, (uint32_t)c->table_dirty, c->rreal, skip)
    if (c->table_empty && region_small_groups(c, expect, nregions)) KH_REGION64(true, 512);
    else if (c->table_empty) KH_REGION64(true, kh::REGION_NT);
    else KH_REGION64(false, kh::REGION_NT);
#undef KH_REGION64
}
template <>
void launch_region<uint32_t>(kh_ctx *c, const kh::PartGeom &g, u64 nregions, u64 hot, const u64 *bend, bool narrow, u64 skip, u64 expect) {
    const kh::TableGeom tg = table_geom(c, c->table, c->cap);
    const int cb = (c->table_empty && !c->shard_shift && want_heads(c)) ? head_count_bits(c, nregions) : -1;
    c->rheads_cb = cb > 0 ? (uint32_t)cb : 0u;
    // (a narrow FRESH pass must write every region of the image whatever the table held: dirty = 1)
    const bool pow2 = g.p2_bits != 0xFFFFFFFFu;  // (the power-of-two instances take digit and start by shifts: rounds 1-3's code)
#define KH_REGION32_P(FRESH, NARROW, NT, DIRTY, CB, RH, P2) \
    hipLaunchKernelGGL((kh::region_count_kernel32<FRESH, NARROW, NT, P2>), dim3((unsigned)nregions), dim3(NT), 0, c->stream, tg, g, \
                       (const uint32_t *)c->keysB, bend, (const u64 *)c->bstart, c->rfail, c->rnew, hot, DIRTY, CB, RH, c->d_ctr, c->rreal, c->ntab, skip, \
                       (bend == (const u64 *)c->bend && g.p1_bits <= kh::MAX_P1_BITS) ? (const uint8_t *)c->heavy : (const uint8_t *)nullptr)  /* (arena level 2: which partitions the exact kernels took; a narrowed batch's "partitions" are regions: no flags) */
#define KH_REGION32(FRESH, NARROW, NT, DIRTY, CB, RH)                     \
    do {                                                                  \
        if (pow2) KH_REGION32_P(FRESH, NARROW, NT, DIRTY, CB, RH, true);  \
        else KH_REGION32_P(FRESH, NARROW, NT, DIRTY, CB, RH, false);      \
    } while (0)
    // A fresh pass into a table that will end at load <= 0.6 runs in 512-lane workgroups, three per CU (at a higher load
    // the probing loop is most of the kernel and wants the waves of two 1024-lane workgroups; a pass over a filled table
    // keeps the old slots in registers: eight per lane would not fit).  Without a hint: the load it would end at if every
    // payload were a new key (a table sized for that ends far below 0.6; one capped by the memory -- an hg38-sized input
    // in 2^32 slots -- may not).  KMERHIP_REGION_NT=512|1024 forces one (A/B, tests).
    const bool small = region_small_groups(c, expect, nregions);
    if (c->table_empty && narrow && small) KH_REGION32(true, true, 512, 1u, c->rheads_cb, c->rheads);
    else if (c->table_empty && narrow) KH_REGION32(true, true, kh::REGION_NT, 1u, c->rheads_cb, c->rheads);
    else if (c->table_empty && small) KH_REGION32(true, false, 512, (uint32_t)c->table_dirty, c->rheads_cb, c->rheads);
    else if (c->table_empty) KH_REGION32(true, false, kh::REGION_NT, (uint32_t)c->table_dirty, c->rheads_cb, c->rheads);
    else if (narrow) KH_REGION32(false, true, kh::REGION_NT, 0u, 0u, (uint32_t *)nullptr);
    else KH_REGION32(false, false, kh::REGION_NT, 0u, 0u, (uint32_t *)nullptr);
#undef KH_REGION32
#undef KH_REGION32_P
}

// One partitioned batch: windows ending in PART_TILE tiles [tile0, tile0+ntiles).
// PT = payload type carried through the partition buffers (partition.hip.h).
// gc: the geometry of the partition passes.  Level 1 needs its p1_bits alone; `size_from_sample` (a fresh batch with 1024
// level-1 partitions): once level 1 has run, the batch's distinct keys are estimated from a few of its partitions
// (distinct_sample_kernel) and the table is made for THAT many keys -- gc.g.b2 is final only from there on.
// range_scale: windows of the whole range / windows of this batch (the estimate of one batch is scaled up to the range).
constexpr double LOAD_SIZED = 0.50;     // a table sized from the sample ends at this load, or a step below (round_cap rounds up) ...
constexpr double LOAD_KEEP_MAX = 0.53;  // ... an existing table is kept up to this load (and 2^31 slots -- 512 buckets per partition, the
constexpr double LOAD_KEEP_MIN = 0.36;  //     fast level-2 shape -- is preferred up to it), and down to this one
u64 policy_cap(double keys) {
    const u64 cap512 = 512ull * 1024 * kh::REGION_SLOTS;
    const double want = keys / LOAD_SIZED;
    if (want > (double)cap512 && keys / LOAD_KEEP_MAX <= (double)cap512) return cap512;
    return round_cap(std::max(want, (double)(2048ull * kh::REGION_SLOTS)));
}

// What follows level 2 -- the region pass, the overflow list, failed and hot buckets, growth -- for payloads of type PT in `bufB`,
// bucket r's in [bstart[r], bend[r]).  A function of its own since round 6: a batch whose level 2 NARROWED its 8-byte payloads to
// the 4 bytes below the region index (partition_batch) goes on here as a 32-bit batch over the virtual geometry g = (p1_bits =
// log2 regions, b2 = 1).
template <typename PT>
int finish_batch(kh_ctx *c, const kh::PartGeom &g, PT *bufB, const u64 *bend, u64 nregions, u64 n_all, u64 n_ub, u64 ovf_lim, bool heavy_exact) {
    int rc;
    const bool was_empty = c->table_empty;
    // The 8-byte table image (see kh_ctx::ntab): a fresh pass with 32-bit payloads writes it, a pass over a table that is
    // in that form updates it.  KMERHIP_NARROW=0: always the 16-byte table (A/B).
    const bool narrow_on = c->knobs.narrow;
    bool nar = sizeof(PT) == 4 && narrow_on && !c->narrow_banned && !c->shard_shift && (c->table_empty || c->narrow);
    if (nar && c->ntab_cap != c->cap) {
        if (c->ntab) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            (void)hipFree(c->ntab);
            c->ntab = nullptr;
            c->ntab_cap = 0;
        }
        if (hipMalloc((void **)&c->ntab, c->cap * sizeof(u64)) != hipSuccess) {  // (no room for the image: the 16-byte table it is)
            (void)hipGetLastError();
            c->ntab = nullptr;
            nar = false;
        } else {
            c->ntab_cap = c->cap;
        }
    }
    if (!nar && c->narrow && (rc = ensure_wide(c)) != KH_OK) return rc;
    if (!nar && (rc = need_table(c)) != KH_OK) return rc;  // (a fresh pass writes every region of it: uninitialised is fine)
    if (nar) c->narrow_g = g;
    // A bucket holding more than a thousandth of the batch (and a million payloads) would keep ONE workgroup of the region
    // pass busy for as long as the whole pass takes: the pass skips it, hot_buckets_kernel counts it afterwards (below).
    // Also 64 x the mean bucket: what makes a bucket hot is one key, not a table with too few regions for the batch (a
    // hint that was far too small: there every bucket is large, and the region pass's overflow handling is what sizes the
    // table) -- so at most a 64th of the buckets can be hot.  KMERHIP_HOT_CUT=n: another threshold (tests); 0 = no bucket is hot.
    const u64 hot_cut = [&] {
        if (c->knobs.hot_cut) return c->knobs.hot_cut;
        return std::max<u64>(std::max<u64>(n_all >> 10, 1ull << 20), 64 * (n_all / nregions));
    }();
    if (hot_cut != ~0ull) {
        StageTimer t(c, ST_MISC);
        hipLaunchKernelGGL(kh::hot_list_kernel, dim3((unsigned)((nregions + kh::BLOCK - 1) / kh::BLOCK)), dim3(kh::BLOCK), 0, c->stream,
                           (const u64 *)c->bstart, bend, (u64)nregions, hot_cut, c->hot_list, c->d_ctr);
    }
    {
        StageTimer t(c, ST_REGION);
        // buckets more than 4x the mean (upper bound) take the skew-guarded probing loop
        launch_region<PT>(c, g, nregions, 4 * (n_ub / nregions) + 4096, bend, nar, hot_cut, n_ub);
    }
    if (nar) c->narrow = true;
    bool heads_marked = false;
    {
        StageTimer t(c, ST_MISC);
        // (three same-address atomics per WAVE at ~10 ns each: a block per 256 regions -- 2048 blocks -- took 0.2 ms for 7 MB)
        hipLaunchKernelGGL(kh::region_reduce_kernel, dim3((unsigned)std::min<u64>(128, (nregions + kh::BLOCK - 1) / kh::BLOCK)), dim3(kh::BLOCK), 0, c->stream,
                           (const u64 *)c->bstart, (const uint8_t *)c->rfail, (const uint32_t *)c->rnew, (const u64 *)c->rreal, (u64)nregions, c->d_ctr);
        // (a long list is mostly copies -- bursts of a tandem repeat's payloads, a repeat family's: summed in LDS first; 4-byte
        //  payloads, and on the 8-byte image only where no count can leave 32 bits: the table's k-mers so far plus this batch's
        //  windows stay below 2^32.  KMERHIP_OVF_AGG=0: never; =1: for lists of any length -- tests)
        // (the head counts of the regions the list touches are made again after its insert: merge.hip)
        heads_marked = was_empty && sizeof(PT) == 4 && c->rheads_cb != 0 && c->ovf_pending && c->ovf_pending <= (64ull << 20) && !heavy_exact &&
                       mark_touched_regions(c, c->ovf_list, c->ovf, ovf_lim) == KH_OK;
        const int agg_env = c->knobs.ovf_agg;
        const bool ovf_agg = sizeof(PT) == 4 && agg_env != 0 && (agg_env == 1 || c->ovf_pending >= (1u << 16)) &&
                             (!nar || c->h_ctr->kmers + n_all < 0xFFFFFFFFull);
        if (c->ovf_pending && ovf_agg) {
            const unsigned grid = (unsigned)std::min<u64>(2048, (c->ovf_pending + 8191) / 8192);
            if (nar)
                hipLaunchKernelGGL((kh::ovf_agg_insert_kernel<true>), dim3(grid), dim3(kh::OVF_AGG_NT), 0, c->stream, table_geom(c, c->table, c->cap), g,
                                   (const kh::OvfEntry *)c->ovf_list, (const u64 *)c->ovf, ovf_lim, c->d_ctr, c->ntab);
            else
                hipLaunchKernelGGL((kh::ovf_agg_insert_kernel<false>), dim3(grid), dim3(kh::OVF_AGG_NT), 0, c->stream, table_geom(c, c->table, c->cap), g,
                                   (const kh::OvfEntry *)c->ovf_list, (const u64 *)c->ovf, ovf_lim, c->d_ctr, (u64 *)nullptr);
        } else if (c->ovf_pending) {  // what did not fit its arena / its bin: through the direct path, now that the table holds the rest
            if (nar)
                hipLaunchKernelGGL((kh::ovf_insert_kernel<PT, true>), dim3(grid_for(c->ovf_pending)), dim3(kh::BLOCK), 0, c->stream,
                                   table_geom(c, c->table, c->cap), g, c->ovf_list, (const u64 *)c->ovf, ovf_lim, c->d_ctr, c->ntab);
            else
                hipLaunchKernelGGL((kh::ovf_insert_kernel<PT, false>), dim3(grid_for(c->ovf_pending)), dim3(kh::BLOCK), 0, c->stream,
                                   table_geom(c, c->table, c->cap), g, c->ovf_list, (const u64 *)c->ovf, ovf_lim, c->d_ctr, (u64 *)nullptr);
        }
    }
    HIP_TRY(c, hipGetLastError());
    // a SHORT overflow list has just been inserted: the head counts of the regions it touched are made again, the others' stand
    // (merge.hip recount_touched_heads; a failed region, a narrow overflow or a long list drop the counts as before)
    bool heads_recounted = false;
    if (heads_marked) heads_recounted = recount_touched_heads(c, nregions) == KH_OK;
    c->table_empty = false;
    c->table_dirty = false;  // the FRESH region pass wrote every region
    c->launches++;
    c->part_batches++;

    // exact bookkeeping after every batch (batches are hundreds of ms; one sync is noise)
    rc = sync_counters(c);
    if (rc != KH_OK) return rc;
    // the per-region exchange-head counts of a FRESH 32-bit pass describe the whole table until
    // anything else touches it
    c->rheads_valid = was_empty && sizeof(PT) == 4 && c->rheads_cb != 0 && c->h_ctr->part_failed == 0 && (c->ovf_pending == 0 || heads_recounted) &&
                      !(nar && c->h_ctr->narrow_ovf);
    c->rheads_wide = c->h_ctr->heads_wide != 0;
    if (nar && c->h_ctr->narrow_ovf) {
        // overflow-list entries whose count would not fit the 8-byte image were left in the list: widen, insert them the
        // 16-byte way (the entries already applied are marked consumed), and keep this table wide from now on
        if ((rc = ensure_wide(c)) != KH_OK) return rc;
        c->narrow_banned = true;
        HIP_TRY(c, hipMemsetAsync(&c->d_ctr->narrow_ovf, 0, sizeof(u64), c->stream));
        hipLaunchKernelGGL((kh::ovf_insert_kernel<PT, false>), dim3(grid_for(c->ovf_pending)), dim3(kh::BLOCK), 0, c->stream,
                           table_geom(c, c->table, c->cap), g, c->ovf_list, (const u64 *)c->ovf, ovf_lim, c->d_ctr, (u64 *)nullptr);
        HIP_TRY(c, hipGetLastError());
        if ((rc = sync_counters(c)) != KH_OK) return rc;
    }
    if (c->h_ctr->part_failed) {
        // some regions overflowed: they were left untouched; grow, then insert their buckets directly.
        // Worst case every key of a failed bucket is new: size the grown table for that.
        std::vector<uint8_t> hf(nregions);
        std::vector<u64> hb(nregions), he(nregions);
        HIP_TRY(c, hipMemcpy(hf.data(), c->rfail, nregions, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(hb.data(), c->bstart, nregions * sizeof(u64), hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(he.data(), bend, nregions * sizeof(u64), hipMemcpyDeviceToHost));
        u64 failed_keys = 0;
        bool any_full = false, any_count = false;
        for (u64 r = 0; r < nregions; ++r)
            if (hf[r]) {
                failed_keys += he[r] - hb[r];
                any_full |= hf[r] == 1;
                any_count |= hf[r] == 2;  // (8-byte image: a count left 32 bits -- the region itself has room)
            }
        if (any_count) c->narrow_banned = true;
        if ((rc = ensure_wide(c)) != KH_OK) return rc;  // (the re-insert below goes through the 16-byte table)
        u64 newcap = c->cap;
        if (any_full) {
            newcap *= 2;
            while ((double)(c->distinct_known + failed_keys) > LOAD_HARD * (double)newcap ||
                   (double)c->distinct_known > LOAD_TARGET * (double)newcap)
                newcap *= 2;
            c->hinted = false;  // the capacity hint (if any) was too small: size later batches for the worst case
        } else {
            while ((double)(c->distinct_known + failed_keys) > LOAD_HARD * (double)newcap) newcap *= 2;
        }
        if (c->trace)
            fprintf(stderr, any_full ? "[kmerhip] %llu regions overflowed (%llu keys): growing and re-inserting them directly\n"
                                     : "[kmerhip] a count left 32 bits in %llu regions (%llu keys): 16-byte table from here on, re-inserting them directly\n",
                    (u64)c->h_ctr->part_failed, failed_keys);
        {
            StageTimer t(c, ST_GROW);
            if (newcap != c->cap) {
                rc = grow_to(c, newcap);
                if (rc != KH_OK) return rc;
            }
            hipLaunchKernelGGL(kh::failed_buckets_insert_kernel<PT>, dim3((unsigned)nregions), dim3(kh::BLOCK), 0, c->stream,
                               table_geom(c, c->table, c->cap), g, (const PT *)bufB, (const u64 *)c->bstart, bend,
                               (const uint8_t *)c->rfail, c->d_ctr);
            HIP_TRY(c, hipMemsetAsync(&c->d_ctr->part_failed, 0, sizeof(u64), c->stream));
        }
        HIP_TRY(c, hipGetLastError());
        rc = sync_counters(c);
        if (rc != KH_OK) return rc;
    }
    if (c->h_ctr->hot) {
        // the buckets the region pass skipped (see hot_cut above), spread over the whole grid, through device atomics: on the
        // 8-byte image where no count can leave 32 bits (the table has counted fewer than 2^32 k-mers, these included: an
        // hg38-sized input keeps its image), else on the 16-byte table, which the table then stays
        const u64 nhot = c->h_ctr->hot, hot_total = c->h_ctr->hot_total;
        const bool crowded = nhot > nregions / 64 && (double)(c->distinct_known + hot_total) > LOAD_HARD * (double)c->cap;  // (see below)
        const bool hot_narrow = sizeof(PT) == 4 && c->narrow && !crowded && c->h_ctr->kmers + hot_total < 0xFFFFFFFFull;
        if (c->trace)
            fprintf(stderr, "[kmerhip] %llu hot bucket(s) (more than %llu payloads each, %llu together) counted apart from the region pass%s\n", nhot,
                    hot_cut, hot_total, hot_narrow ? ", into the 8-byte image" : "");
        if (!hot_narrow && (rc = ensure_wide(c)) != KH_OK) return rc;
        // The hot kernel inserts through device atomics: a region without room is an error there, not a retry.  Hot buckets
        // are few (<= a 64th of the buckets with the default threshold) and hold few keys, and a table that is too small
        // shows in the OTHER regions first (they fail, the table grows: above).  Where most buckets were declared hot (a
        // forced threshold, tests) nothing has witnessed the table's size: make room for the worst case first.
        if (crowded) {
            u64 newcap = c->cap * 2;
            while ((double)(c->distinct_known + hot_total) > LOAD_TARGET * (double)newcap) newcap *= 2;
            StageTimer t(c, ST_GROW);
            if ((rc = grow_to(c, newcap)) != KH_OK) return rc;
        }
        {
            StageTimer t(c, ST_MISC);
            if (hot_narrow)
                hipLaunchKernelGGL((kh::hot_buckets_kernel<PT, sizeof(PT) == 4>), dim3(kh::HOT_GRID), dim3(kh::BLOCK), 0, c->stream, table_geom(c, c->table, c->cap), g,
                                   (const PT *)bufB, (const u64 *)c->bstart, bend, (const uint32_t *)c->hot_list, nhot, c->d_ctr, c->ntab);
            else
                hipLaunchKernelGGL((kh::hot_buckets_kernel<PT, false>), dim3(kh::HOT_GRID), dim3(kh::BLOCK), 0, c->stream, table_geom(c, c->table, c->cap), g,
                                   (const PT *)bufB, (const u64 *)c->bstart, bend, (const uint32_t *)c->hot_list, nhot, c->d_ctr, (u64 *)nullptr);
            HIP_TRY(c, hipMemsetAsync(&c->d_ctr->hot, 0, 2 * sizeof(u64), c->stream));  // hot + hot_total
        }
        HIP_TRY(c, hipGetLastError());
        c->rheads_valid = false;
        if ((rc = sync_counters(c)) != KH_OK) return rc;
    }
    if ((double)c->distinct_known > LOAD_PART * (double)c->cap) {
        u64 newcap = c->cap * 2;
        while ((double)c->distinct_known > LOAD_TARGET * (double)newcap) newcap *= 2;
        StageTimer t(c, ST_GROW);
        rc = grow_to(c, newcap);
        if (rc != KH_OK) return rc;
    }
    return KH_OK;
}

template <typename PT>
int partition_batch(kh_ctx *c, const RangeArgs &ra, GeomChoice &gc, u64 tile0, u64 ntiles, bool size_from_sample, double range_scale, bool sized_narrow, bool may_narrow) {
    constexpr bool CHUNKED = true;  // level 1 always goes into the chunk pool (partition.hip.h)
    kh::PartGeom &g = gc.g;
    const u64 P1 = 1ull << g.p1_bits;
    const u64 n_all = ntiles * kh::PART_TILE;  // every window of these tiles
    const bool estimated = ra.survive < 1.0;
    const u64 n_ub = sized_for(n_all, ra.survive);  // upper bound on keys (an estimate when `estimated`: checked after level 1)
    // chunk pool: every payload + one partial chunk per (workgroup, partition) + the unused tail of
    // every workgroup's private ranges
    const u64 pool_chunks = (n_ub / kh::CHUNK_PAY) + (n_ub / kh::CHUNK_PAY) / 24 + (u64)PART_G1 * (P1 + kh::POOL_GRAB) + 1024;
    const u64 max_blocks = pool_chunks / kh::CPB + P1 + 1;
    int rc;
    // ---- what level 1 needs: the pool and its metadata (independent of the table's size) ----
    if (!c->moff) {  // fixed-size scratch, allocated once
        u64 z = 0;
        z = 0;
        if ((rc = ensure_buf(c, &c->moff, &z, (u64)kh::MAX_P1 + 1, "hipMalloc(moff)")) != KH_OK) return rc;
        z = 0;
        if ((rc = ensure_buf(c, &c->nch, &z, (u64)kh::MAX_P1 + 1, "hipMalloc(nch)")) != KH_OK) return rc;
        z = 0;
        if ((rc = ensure_buf(c, &c->info, &z, 8, "hipMalloc(info)")) != KH_OK) return rc;
        z = 0;
        if ((rc = ensure_buf(c, &c->pcount, &z, (u64)kh::MAX_P1, "hipMalloc(pcount)")) != KH_OK) return rc;
        z = 0;
        if ((rc = ensure_buf(c, &c->pstart, &z, (u64)kh::MAX_P1 + 1, "hipMalloc(pstart)")) != KH_OK) return rc;
        z = 0;
        if ((rc = ensure_buf(c, &c->pool_next, &z, 1, "hipMalloc(pool_next)")) != KH_OK) return rc;
        z = 0;
        if ((rc = ensure_buf(c, &c->ptotal, &z, (u64)kh::MAX_P1, "hipMalloc(ptotal)")) != KH_OK) return rc;
        z = 0;
        if ((rc = ensure_buf(c, &c->pcap, &z, (u64)kh::MAX_P1, "hipMalloc(pcap)")) != KH_OK) return rc;
        z = 0;
        if ((rc = ensure_buf(c, &c->ovf, &z, 4, "hipMalloc(ovf)")) != KH_OK) return rc;
        z = 0;
        if ((rc = ensure_buf(c, &c->heavy, &z, (u64)kh::MAX_P1, "hipMalloc(heavy)")) != KH_OK) return rc;
    }
    if ((rc = ensure_buf(c, &c->blocks, &c->blocks_cap, max_blocks, "hipMalloc(blocks)")) != KH_OK) return rc;
    if (c->pool_cap < pool_chunks) {
        u64 z = c->chunk_part ? c->pool_cap : 0;
        if ((rc = ensure_buf(c, &c->chunk_part, &z, pool_chunks, "hipMalloc(chunk_part)")) != KH_OK) return rc;
        z = c->fill8 ? c->pool_cap : 0;
        if ((rc = ensure_buf(c, &c->fill8, &z, pool_chunks, "hipMalloc(fill8)")) != KH_OK) return rc;
        z = c->plist ? c->pool_cap : 0;
        if ((rc = ensure_buf(c, &c->plist, &z, pool_chunks, "hipMalloc(plist)")) != KH_OK) return rc;
        c->pool_cap = pool_chunks;
    }
    const u64 a_bytes = pool_chunks * kh::CHUNK_PAY * sizeof(PT);  // A: the level-1 pool
    if (c->key_cap < a_bytes) {  // (capacities in BYTES)
        u64 z = c->keysA ? c->key_cap : 0;
        if ((rc = ensure_buf(c, &c->keysA, &z, a_bytes, "hipMalloc(keysA)")) != KH_OK) return rc;
        c->key_cap = a_bytes;
    }
    // the sample: enough partitions for ~2 M payloads (one partition of a large batch), a set with room for all of them
    const uint32_t est_np = size_from_sample ? (uint32_t)std::min<u64>(16, std::max<u64>(1, (2ull << 20) / std::max<u64>(1, n_ub / P1))) : 0u;
    constexpr uint32_t EST_P0 = 517;  // (not partition 0: the hash of A^k is 0 -- its partition is the one a homopolymer makes heavy)
    u64 est_slots = 0;
    uint32_t est_sub = 0;  // ... and of a large batch's partition only the keys with est_sub zero bits behind the level-1 digit
    if (size_from_sample) {
        while (est_sub < 6 && (n_ub / P1) >> (est_sub + 1) >= (1ull << 19)) ++est_sub;
        est_slots = 1ull << 16;
        while (est_slots < 3 * (u64)est_np * ((n_ub / P1 >> est_sub) + 1)) est_slots *= 2;
        if ((rc = ensure_buf(c, &c->est_set, &c->est_set_cap, est_slots, "hipMalloc(distinct sample)")) != KH_OK) return rc;
    }

    const uint32_t tpb = (uint32_t)((ntiles + PART_G1 - 1) / PART_G1);
    const uint32_t thr = ra.use_qual ? qual_thr(c) : 0;
    const dim3 b1(kh::PART_NT);
    kh::ChunkSrc cs;
    cs.pay = c->keysA;
    cs.plist = c->plist;
    cs.fill8 = c->fill8;
    const uint32_t force_wide = c->knobs.p2_force_wide ? 1u : 0u;
    bool have_total = false;  // the host knows how many payloads level 1 produced (it synchronised to read them)
    u64 batch_total = 0;

    {
        {
            StageTimer t(c, ST_MISC);
            HIP_TRY(c, hipMemsetAsync(c->chunk_part, 0xFF, pool_chunks * sizeof(uint16_t), c->stream));
            HIP_TRY(c, hipMemsetAsync(c->fill8, 0xFF, pool_chunks, c->stream));
            HIP_TRY(c, hipMemsetAsync(c->pcount, 0, kh::MAX_P1 * sizeof(uint32_t), c->stream));
            HIP_TRY(c, hipMemsetAsync(c->pool_next, 0, sizeof(u64), c->stream));
        }
        {
            StageTimer t(c, ST_P1_SCATTER);
            // level 1 lives in translation units of its own (level1_api.h): one kernel per k for the written-out window
            kh::L1Launch l1;
            l1.stream = c->stream;
            l1.grid = (unsigned)PART_G1;
            l1.abase = ra.abase;
            l1.qbase = ra.qbase;
            l1.qaligned = ra.qaligned;
            l1.use_qual = ra.use_qual;
            l1.vbeg = ra.vbeg;
            l1.vend = ra.vend;
            l1.wlo = ra.wlo;
            l1.tile0 = tile0;
            l1.ntiles = ntiles;
            l1.tiles_per_block = tpb;
            l1.k = c->k;
            l1.thr = thr;
            l1.g = g;  // (level 1 reads p1_bits, k and the shard fields: not b2)
            l1.pool = c->keysA;
            l1.chunk_part = c->chunk_part;
            l1.fill8 = c->fill8;
            l1.pool_next = c->pool_next;
            l1.pool_chunks = pool_chunks;
            l1.ctr = c->d_ctr;
            // KMERHIP_GENERIC_K=1: the C++ window where a written-out one exists; KMERHIP_P1_BINS=0: round 1's tile-sorting kernel (both for A/B)
            l1.survive = ra.survive;
            l1.generic_k = c->knobs.generic_k;
            l1.legacy = c->knobs.p1_legacy;
            if (sizeof(PT) == 4) kh::launch_level1_32(l1, nullptr);
            else kh::launch_level1_64(l1, nullptr);
        }
#if KH_ABL
        if (c->knobs.stop_after_p1) {  // ablation builds only: time level 1 alone (its output is garbage)
            HIP_TRY(c, hipGetLastError());
            return sync_counters(c);
        }
#endif
        {
            StageTimer t(c, ST_MISC);
            HIP_TRY(c, hipMemsetAsync(c->ptotal, 0, kh::MAX_P1 * sizeof(u64), c->stream));
            hipLaunchKernelGGL(kh::chunk_hist_kernel, dim3(1024), dim3(1024), 0, c->stream, (const uint16_t *)c->chunk_part,
                               (const u64 *)c->pool_next, pool_chunks, c->pcount, (const uint8_t *)c->fill8, c->ptotal);
            if ((rc = device_scan(c, c->pcount, P1, c->pstart)) != KH_OK) return rc;
            // (the plan's moff / mbase depend on b2: it runs again below once that is final; this run sets the chunk list's cursors)
            hipLaunchKernelGGL(kh::part2_plan_chunked_kernel, dim3(1), dim3(1024), 0, c->stream, (const u64 *)c->pstart, g,
                               c->blocks, max_blocks, c->moff, c->nch, c->info, c->pcount, force_wide, (const uint8_t *)nullptr);
            hipLaunchKernelGGL(kh::chunk_list_kernel, dim3((unsigned)((pool_chunks + 16383) / 16384)), dim3(1024), 0, c->stream,
                               (const uint16_t *)c->chunk_part, (const u64 *)c->pool_next, pool_chunks, c->pcount, c->plist);
            if (size_from_sample) {
                u64 *est_out = c->info + 4;  // [distinct, payloads seen, no room]
                HIP_TRY(c, hipMemsetAsync(est_out, 0, 3 * sizeof(u64), c->stream));
                HIP_TRY(c, hipMemsetAsync(c->est_set, 0xFF, est_slots * sizeof(u64), c->stream));
                hipLaunchKernelGGL(kh::distinct_sample_kernel<PT>, dim3(1024), dim3(kh::BLOCK), 0, c->stream, cs, (const u64 *)c->pstart,
                                   EST_P0 % (uint32_t)(P1 - est_np + 1), est_np, est_sub, c->est_set, est_slots - 1, est_out);
            }
        }
        if (estimated || size_from_sample) {
            // Everything behind the pool is sized for n_ub payloads, an estimate: are there more?  (The pool itself has
            // slack -- a partial chunk per workgroup and partition -- so level 1 may well have found room for them: what
            // counts is the total, from chunk_hist_kernel; and payloads level 1 found no room for are in ctr->failed,
            // which is 0 on entry.)  Nothing but the pool and its chunk lists has been written yet.
            std::vector<u64> pt(kh::MAX_P1);
            u64 lost = 0, total = 0, est[3] = {0, 0, 0};
            HIP_TRY(c, hipMemcpyAsync(pt.data(), c->ptotal, kh::MAX_P1 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(&lost, &c->d_ctr->failed, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
            if (size_from_sample) HIP_TRY(c, hipMemcpyAsync(est, c->info + 4, sizeof(est), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            for (u64 p = 0; p < P1; ++p) total += pt[p];
            have_total = true;
            batch_total = total;
            if (estimated && (lost || total > n_ub)) {
                HIP_TRY(c, hipMemsetAsync(&c->d_ctr->failed, 0, sizeof(u64), c->stream));
                if (c->trace)
                    fprintf(stderr, "[kmerhip] sized for %.3f of the windows (%llu payloads), found %llu%s: again at full size\n", ra.survive, n_ub,
                            total, lost ? " and more that did not fit the pool" : "");
                return KH_RETRY_FULL_SIZE;
            }
            if (size_from_sample) {
                // distinct keys of the batch: the sample's, scaled by payloads (its partitions hold est[1] of `total`) -- and of the
                // range, scaled by windows: an upper bound (later batches repeat keys of this one)
                double keys = (double)total;  // no usable sample: as if every payload were a new key (round 3's sizing)
                // (scaled by KEY SPACE -- the sample is every occurrence of an exact 1 / (1024 / np x 2^sub) of it -- not by payloads: a
                //  sampled partition that holds a repeat family's heavy keys has more payloads, not more keys; an hg-shaped input
                //  came out 21 % low that way)
                if (est[2] == 0 && est[1] > 0) keys = std::min((double)total, (double)est[0] * ((double)P1 / (double)est_np) * (double)(1u << est_sub));
                c->est_keys = (u64)(keys * range_scale) + 1;
                const double load_now = (double)c->est_keys / (double)c->cap;
                u64 newcap = c->cap;
                // (a capacity hint is the caller's word on ALL the keys to come, this range being perhaps the first of many: a hinted
                //  table is never made smaller, only larger when the sample says the hint cannot be right)
                if (load_now > LOAD_KEEP_MAX || (!c->hinted && (load_now < LOAD_KEEP_MIN || c->cap < 2048ull * kh::REGION_SLOTS))) newcap = policy_cap((double)c->est_keys);
                if (c->hinted && newcap < c->cap) newcap = c->cap;
                // two levels of partitioning reach 1024 x 1024 regions: beyond that (more than ~3 G keys in one range) the table
                // grows by rehash after the batch and later batches take the direct path, as before
                newcap = std::min<u64>(newcap, (u64)kh::MAX_P1 * kh::MAX_B2 * kh::REGION_SLOTS);
                size_t fr = 0, tot = 0;
                if (hipMemGetInfo(&fr, &tot) == hipSuccess) {  // never beyond a third of what is free (the partition buffers of this batch come next)
                    u64 room = ((u64)fr + (c->table ? c->cap * sizeof(Slot) : 0) + (c->ntab ? c->ntab_cap * sizeof(u64) : 0)) / 3;
                    if (c->knobs.table_room_mb) room = c->knobs.table_room_mb << 20;  // (test build: as if that were all the room)
                    // (strictly smaller at every turn: round_cap() rounds UP, and 0.8 x a power of two -- every size up to 2^28 slots,
                    //  every size with KMERHIP_POW2_TABLE=1 -- rounds back to where it came from: ADVICE r4, a loop without an end)
                    while (newcap > c->cap && newcap * sizeof(Slot) > room) {
                        u64 next = round_cap((double)newcap * 0.8);
                        if (next >= newcap) next = round_cap((double)newcap * 0.5);
                        if (next >= newcap) break;
                        newcap = std::max(next, c->cap);
                    }
                }
                // 8-byte payloads (k >= 22): where a POWER-OF-TWO table of the next size up lets level 2 narrow them to the 4 bytes below
                // the region index (2k - log2 regions <= 32: see `can_narrow` below), take it -- a table at load 0.25-0.5 whose batches
                // then move half the bytes through level 2 and the region pass, and whose region pass is the 32-bit kernel (k = 25 on
                // S100M: 146 ms per step with 1024 x 640 regions, 8 bytes a payload)
                if (sizeof(PT) == 8 && c->knobs.l2_narrow && c->shard_shift == 0 && (newcap & (newcap - 1)) && !c->knobs.table_regions) {  // (a forced geometry stays)
                    u64 p2cap = 2048ull * kh::REGION_SLOTS;
                    uint32_t rb = 11;
                    while (p2cap < newcap) {
                        p2cap <<= 1;
                        ++rb;
                    }
                    const int below = 2 * (int)c->k - (int)rb;
                    size_t fr2 = 0, tot2 = 0;
                    // (its 8-byte image within a third of the free memory, and beside what this batch still has to allocate -- the arenas of
                    //  4-byte payloads and the overflow list, ~7 bytes per payload: the partition budget was drawn when the table was half
                    //  the size.  A batch that cannot narrow after all -- heavy partitions -- needs the 16-byte table: it comes back with
                    //  KH_RETRY_WIDE where that does not fit, and runs again in batches that leave it room.)
                    const u64 image_have = c->ntab ? c->ntab_cap * sizeof(u64) : 0, later = 7 * total > c->keyb_cap ? 7 * total - c->keyb_cap : 0;
                    // (a table that HAS that size -- the context's last pass made it so -- simply keeps it: nothing to allocate)
                    const bool fits = p2cap == c->cap ||
                                      (hipMemGetInfo(&fr2, &tot2) == hipSuccess && p2cap * sizeof(u64) <= ((u64)fr2 + image_have) / 3 &&
                                       (u64)fr2 + image_have >= p2cap * sizeof(u64) + later + (6ull << 30));
                    if (below >= 1 && below <= 32 && rb <= kh::MAX_P1_BITS + kh::MAX_P2_BITS && fits && (!c->hinted || p2cap >= c->cap)) newcap = p2cap;
                }
                if (c->trace)
                    fprintf(stderr, "[kmerhip] %llu payloads, ~%llu distinct (sample: %llu of %llu in %u partition(s)%s): table %llu -> %llu slots, load %.3f\n", total,
                            c->est_keys, est[0], est[1], est_np, est[2] ? ", VOID" : "", c->cap, newcap, (double)c->est_keys / (double)newcap);
                if (newcap != c->cap) {
                    // the table is empty (a lazily reset one may hold stale slots: the same to us): it just has another size now
                    resize_empty_table(c, newcap);
                    c->sized_by_sample = true;
                    const GeomChoice g2 = make_geom(c, c->cap);
                    if (!g2.ok || g2.g.p1_bits != g.p1_bits || g2.use32 != gc.use32) return fail(c, KH_ERR_STATE, "table geometry changed under a running batch");
                    gc = g2;
                }
            }
        }
    }
    // ---- what depends on the table's size ----
    // (level 2's output is sized from the payloads level 1 really produced where the host has just read that number)
    const u64 n_pay = have_total ? std::min(n_ub, batch_total) : n_ub;
    const u64 nregions = kh::part_regions(g);
    const u64 n2 = max_blocks * g.b2;
    if (c->h2_cap < n2) {  // H2 and O2 grow together
        u64 z = c->h2_cap;
        if ((rc = ensure_buf(c, &c->H2, &z, n2, "hipMalloc(H2)")) != KH_OK) return rc;
        z = c->O2 ? c->h2_cap + 1 : 0;
        if ((rc = ensure_buf(c, &c->O2, &z, n2 + 1, "hipMalloc(O2)")) != KH_OK) return rc;
        c->h2_cap = n2;
    }
    if ((rc = ensure_region_scratch(c, nregions)) != KH_OK) return rc;
    // 32-bit payloads with 2..512 buckets per partition: level 2 writes whole aligned lines, every (bucket,
    // workgroup) segment padded to a line with sentinels (KMERHIP_P2_LINES=0: the unpadded kernel, for A/B)
    const bool lines_on = c->knobs.p2_lines;
    const bool lines = lines_on && g.b2 >= 2 && g.b2 <= 512;
    // Level 2 without a counting pass (partition.hip.h, part2_arena_kernel): >= 256 level-1 partitions (one workgroup
    // each), 32..1024 buckets per partition.  KMERHIP_L2_ARENA=0: always the exact count -> scan -> scatter path.
    const bool arena_on = c->knobs.l2_arena;
    const bool arena = arena_on && g.p1_bits >= 8 && g.b2 >= 32 && g.b2 <= kh::MAX_B2;
    const u64 arena_pay = arena ? (n_pay + nregions) + ((n_pay + nregions) >> 2) + 1056ull * nregions : 0;  // upper bound of arena_plan_kernel's total
    // the overflow list: a sixteenth of the batch, plus what the workgroups RESERVE without using -- every workgroup that
    // overflows at all takes private 8192-entry segments (part2_arena_kernel, OVF_SEG), so a batch in which most of the
    // P1 partitions hold one moderately heavy bucket needs P1 segments before the first entry beyond them is "list full"
    const u64 ovf_need = arena ? n_pay / 16 + 2 * P1 * 8192ull + (1ull << 20) : 0;
    const u64 pad_ub = lines ? (max_blocks * g.b2) * (u64)(kh::P2L<PT>::UNIT - 1) : 0;  // sentinels at the segment ends
    // Heavy level-1 partitions (a homopolymer's, a satellite's: arena_plan_kernel) go through the exact kernels while the
    // others take the arenas; their buckets follow the arenas in the same buffer: room for an eighth of the batch there
    // (more than that in heavy partitions: the batch takes the exact path as a whole).
    const u64 heavy_room = arena ? n_pay / 8 : 0;
    const u64 heavy_base = (arena_pay + 31) & ~31ull;   // payload index behind the arenas (an upper bound of their total)
    const u64 heavy_pad = (arena && lines) ? ((heavy_room / (kh::CPB * kh::CHUNK_PAY) + 2 * P1) * g.b2) * (u64)(kh::P2L<PT>::UNIT - 1) : 0;
    // B: the level-2 output -- exact path: every payload + sentinel padding; arenas: a quarter more
    // LEVEL 2 NARROWS (round 6; partition.hip.h part2_arena_kernel, IT): 8-byte payloads whose hash bits below the REGION index fit 32
    // bits -- 2k - log2(regions) <= 32: k <= 25 in the headline's 2^19 regions -- leave level 2 as that 4-byte word, where the table's
    // regions are a power of two (the bits below the region index are then a bit field of the payload).  What follows is a 32-bit
    // batch over the virtual geometry (p1_bits = log2 regions, b2 = 1): the same table layout -- for powers of two any split of
    // the region index describes it --, half the bytes, the 32-bit region kernel and the 8-byte table image.  A batch that turns
    // out to have heavy partitions, or whose overflow list fills up, runs level 2 again un-narrowed (the pool is still whole).
    const int below_region_bits = 2 * (int)c->k - (int)g.shard_shift - (int)g.p1_bits - (g.p2_bits != 0xFFFFFFFFu ? (int)g.p2_bits : 64);
    const bool can_narrow = sizeof(PT) == 8 && arena && c->knobs.l2_narrow && may_narrow && g.p2_bits != 0xFFFFFFFFu && g.p2_bits >= 5 && g.b2 <= kh::MAX_B2 &&
                            below_region_bits >= 1 && below_region_bits <= 32 && g.shard_shift == 0;
    const u64 b_bytes_wide = std::max((n_pay + pad_ub) * (u64)sizeof(PT), arena ? (heavy_base + heavy_room + heavy_pad + 64) * (u64)sizeof(PT) : 0);
    const u64 b_bytes = can_narrow ? (arena_pay + 64) * (u64)sizeof(uint32_t) : b_bytes_wide;  // (a narrowed batch: arenas of 4-byte payloads, nothing behind them)
    // room for the 8-byte output after all (the batch cannot narrow, or its narrowed level 2 has to be run again wide): what is there, what can
    // be allocated -- or the caller makes smaller batches
    auto room_for_wide = [&]() -> int {
        if (c->keyb_cap >= b_bytes_wide) return KH_OK;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) {
            (void)hipGetLastError();
            fr = 0;
        }
        // (8-byte payloads end in the 16-byte table: its room too, where it does not exist yet)
        if (sized_narrow && ((u64)fr + c->keyb_cap < b_bytes_wide + (c->table ? 0 : c->cap * sizeof(Slot)) + (4ull << 30) || c->knobs.l2_no_room_wide)) return KH_RETRY_WIDE;
        u64 z = c->keysB ? c->keyb_cap : 0;
        const int r = ensure_buf(c, &c->keysB, &z, b_bytes_wide, "hipMalloc(keysB)");
        if (r == KH_OK) c->keyb_cap = b_bytes_wide;
        return r;
    };
    if (!can_narrow) {
        if ((rc = room_for_wide()) != KH_OK) return rc;
    } else if (c->keyb_cap < b_bytes) {
        u64 z = c->keysB ? c->keyb_cap : 0;
        if ((rc = ensure_buf(c, &c->keysB, &z, b_bytes, "hipMalloc(keysB)")) != KH_OK) return rc;
        c->keyb_cap = b_bytes;
    }
    PT *bufA = reinterpret_cast<PT *>(c->keysA), *bufB = reinterpret_cast<PT *>(c->keysB);  // (bufB: again after a re-allocation)
    if (arena && c->ovf_cap < ovf_need) {
        u64 z = c->ovf_list ? c->ovf_cap : 0;
        if ((rc = ensure_buf(c, &c->ovf_list, &z, ovf_need, "hipMalloc(ovf_list)")) != KH_OK) return rc;
        c->ovf_cap = ovf_need;
    }
    const u64 *bend = c->bstart + 1;  // end of region r's data: the next region's start (exact path) or c->bend[r] (arenas)
    bool arena_done = false, heavy_exact = false;
    const u64 ovf_test_cap = c->knobs.l2_ovf_cap;
    const u64 ovf_lim = std::min(c->ovf_cap, ovf_test_cap);
    bool narrowed = false;
    if (arena) {
        u64 hov[4] = {0, 0, 0, 0};
        for (int attempt = can_narrow ? 0 : 1; attempt < 2; ++attempt) {
            const bool nw = attempt == 0;
            StageTimer t(c, ST_P2_SCATTER);
            // (test knobs: KMERHIP_L2_SKEW_X = how many times the mean a partition may hold before it counts as heavy, 0 = no
            //  limit; KMERHIP_L2_OVF_CAP = entries the overflow list may take; KMERHIP_L2_HEAVY_ROOM = payloads of room for heavy partitions)
            const uint32_t skew_x = c->knobs.l2_skew_x >= 0 ? (uint32_t)c->knobs.l2_skew_x : 2u;
            const u64 room = std::min<u64>(heavy_room, c->knobs.l2_heavy_room);
            hipLaunchKernelGGL(kh::arena_plan_kernel, dim3((unsigned)std::min<u64>(64, (nregions + 1023) / 1024)), dim3(1024), 0, c->stream, (const u64 *)c->ptotal, g, c->bstart, c->pcap,
                               c->ovf, skew_x, c->heavy, room);
#define KH_ARENA(UB, NBK, P2)                                                                                                              \
    hipLaunchKernelGGL((kh::part2_arena_kernel<PT, UB, NBK, P2>), dim3((unsigned)P1), dim3(kh::P2L_NT), 0, c->stream, cs, (const u64 *)c->pstart, g, \
                       (const u64 *)c->bstart, (const uint32_t *)c->pcap, bufB, c->bend, c->ovf_list, c->ovf, ovf_lim, (const uint8_t *)c->heavy)
            const bool pow2 = g.p2_bits != 0xFFFFFFFFu;
            constexpr int UB512 = sizeof(PT) == 4 ? KH_ARENA_UNITB : 64;
#define KH_ARENA_NARROW(UB, NBK)                                                                                                           \
    hipLaunchKernelGGL((kh::part2_arena_kernel<uint32_t, UB, NBK, true, u64>), dim3((unsigned)P1), dim3(kh::P2L_NT), 0, c->stream, cs, (const u64 *)c->pstart, g, \
                       (const u64 *)c->bstart, (const uint32_t *)c->pcap, (uint32_t *)c->keysB, c->bend, c->ovf_list, c->ovf, ovf_lim, (const uint8_t *)c->heavy)
            if (nw) {
                if constexpr (sizeof(PT) == 8) {
                    if (g.b2 > 512) KH_ARENA_NARROW(64, 1024);
                    else KH_ARENA_NARROW(KH_ARENA_UNITB, 512);
                }
            } else
            if (g.b2 > 768) {  // 769 .. 1024 buckets per partition: the 128 KiB of bins shared out among them, 64-byte units, four buckets per lane group
                if (pow2) KH_ARENA(64, 1024, true);
                else KH_ARENA(64, 1024, false);
            } else if (g.b2 > 512) {  // 513 .. 768: three buckets per lane group
#ifndef KH_ARENA_UNITB_768
#define KH_ARENA_UNITB_768 128  // whole lines while a bin holds >= 52 payloads of the 768-bucket instance's 144 KiB, i.e. up to 682 buckets (64: A/B builds)
#endif
                if (KH_ARENA_UNITB_768 == 128 && sizeof(PT) == 4 && g.b2 <= 682) KH_ARENA(128, 768, false);
                else KH_ARENA(64, 768, false);
            } else if (pow2) {
                KH_ARENA(UB512, 512, true);
            } else {
                KH_ARENA(UB512, 512, false);
            }
#undef KH_ARENA
#undef KH_ARENA_NARROW
            t.stop();
            HIP_TRY(c, hipMemcpyAsync(hov, c->ovf, sizeof(hov), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (nw && (hov[1] != 0 || hov[2] != 0)) {  // (heavy partitions' buckets come from the exact kernels, 8 bytes a payload: the batch stays wide)
                if (c->trace) fprintf(stderr, "[kmerhip] level 2 narrowed to 4-byte payloads met %s: again with 8-byte payloads\n", hov[1] ? "a full overflow list" : "heavy partitions");
                if ((rc = room_for_wide()) != KH_OK) return rc;
                bufB = reinterpret_cast<PT *>(c->keysB);
                continue;
            }
            narrowed = nw;
            break;
        }
        if (hov[1] == 0) {
            arena_done = true;
            heavy_exact = hov[2] != 0;
            bend = c->bend;
            c->ovf_pending = std::min<u64>(hov[0], ovf_lim);  // (the cursor moves in whole segments: it may end beyond the list)
            HIP_TRY(c, hipMemsetAsync(c->rfail, 0, nregions, c->stream));
            if (c->trace && c->ovf_pending)
                fprintf(stderr, "[kmerhip] level 2: %llu entries handed out to the overflow list (%.2f %% of the batch's payload room)\n", (u64)c->ovf_pending,
                        100.0 * (double)c->ovf_pending / (double)n_ub);
            if (heavy_exact && c->trace)
                fprintf(stderr, "[kmerhip] %llu heavy level-1 partition(s) (%llu payloads) take the exact level-2 kernels, the others the arenas\n", hov[2], hov[3]);
        } else if (c->trace) {
            fprintf(stderr, hov[1] == 2 ? "[kmerhip] heavy level-1 partitions hold %llu payloads, more than the room behind the arenas: this batch takes the exact level-2 path\n"
                                        : "[kmerhip] level-2 overflow list full (%llu entries): this batch takes the exact path\n", (u64)(hov[1] == 2 ? hov[3] : hov[0]));
        }
    }
    if (!arena_done || heavy_exact) {
    // the exact kernels: over every partition (the plan above), or over the heavy ones of an arena batch alone -- their
    // buckets then go behind the arenas, and their (small) counting pass is booked under "misc": stage_ms[P2_COUNT] == 0
    // still says "this batch's level 2 was the arena kernel"
    PT *const outB = heavy_exact ? bufB + heavy_base : bufB;
    {
        // (the exact path's plan and its cleared histogram matrix are made HERE since round 5: a batch the arenas took whole --
        //  every batch of the bench -- never reads them.  The plan runs again because b2 is final only now -- the matrix offsets
        //  depend on it -- or for the heavy partitions alone.)
        StageTimer t(c, ST_MISC);
        if (heavy_exact || size_from_sample)
            hipLaunchKernelGGL(kh::part2_plan_chunked_kernel, dim3(1), dim3(1024), 0, c->stream, (const u64 *)c->pstart, g,
                               c->blocks, max_blocks, c->moff, c->nch, c->info, c->pcount, force_wide, heavy_exact ? (const uint8_t *)c->heavy : (const uint8_t *)nullptr);
        HIP_TRY(c, hipMemsetAsync(c->H2, 0, n2 * sizeof(uint32_t), c->stream));
    }
    {
        StageTimer t(c, heavy_exact ? ST_MISC : ST_P2_COUNT);
        hipLaunchKernelGGL((kh::part2_count_kernel<PT, CHUNKED>), dim3((unsigned)max_blocks), b1, 0, c->stream, (const PT *)bufA, cs,
                           (const kh::Part2Block *)c->blocks, (const u64 *)c->info, g, c->H2, lines ? (uint32_t)kh::P2L<PT>::UNIT : 1u);
    }
    {
        StageTimer t(c, ST_MISC);
        if ((rc = device_scan(c, c->H2, n2, c->O2)) != KH_OK) return rc;
    }
    {
        StageTimer t(c, ST_P2_SCATTER);
        const uint32_t fallback = lines ? 1u : 0u;  // behind the unit kernel the unaligned one runs only where that stood down
        if (lines)  // whole aligned 64-byte units only (32-bit payloads, 2..512 buckets per partition)
            hipLaunchKernelGGL((kh::part2_scatter_lines_kernel<PT, CHUNKED>), dim3((unsigned)max_blocks), dim3(kh::P2L_NT), 0, c->stream,
                               (const PT *)bufA, cs, (const kh::Part2Block *)c->blocks, (const u64 *)c->info, g, (const u64 *)c->O2, outB);
        if (g.b2 <= 512)  // <= 512 buckets per partition: the small-LDS variant, two workgroups per CU
            hipLaunchKernelGGL((kh::part2_scatter_kernel<PT, CHUNKED, 512>), dim3((unsigned)max_blocks), dim3(kh::PART2_NT), 0, c->stream,
                               (const PT *)bufA, cs, (const kh::Part2Block *)c->blocks, (const u64 *)c->info, g, (const u64 *)c->O2, outB, fallback);
        else
            hipLaunchKernelGGL((kh::part2_scatter_kernel<PT, CHUNKED, 1024>), dim3((unsigned)max_blocks), dim3(kh::PART2_NT), 0, c->stream,
                               (const PT *)bufA, cs, (const kh::Part2Block *)c->blocks, (const u64 *)c->info, g, (const u64 *)c->O2, outB, fallback);
    }
#if KH_ABL2 || KH_ABL3
    if (c->knobs.stop_after_p2) {  // ablation builds only: time level 2 alone (its output is garbage)
        HIP_TRY(c, hipGetLastError());
        return sync_counters(c);
    }
#endif
    {
        StageTimer t(c, ST_MISC);
        if (heavy_exact) {
            hipLaunchKernelGGL(kh::bucket_bounds_heavy_kernel, dim3((unsigned)((nregions + 255) / 256)), dim3(256), 0, c->stream,
                               (const u64 *)c->O2, (const u64 *)c->moff, (const uint32_t *)c->nch, g, (const uint8_t *)c->heavy, heavy_base,
                               c->bstart, c->bend);
        } else {
            hipLaunchKernelGGL(kh::bucket_bounds_kernel, dim3((unsigned)((nregions + 256) / 256)), dim3(256), 0, c->stream,
                               (const u64 *)c->O2, (u64)n2, (const u64 *)c->moff, (const uint32_t *)c->nch, g, c->bstart);
            HIP_TRY(c, hipMemsetAsync(c->rfail, 0, nregions, c->stream));
        }
    }
    if (!heavy_exact) c->ovf_pending = 0;
    }  // exact kernels
    if (narrowed) {  // a 32-bit batch from here on: regions as "partitions" of one bucket each, the payload = the 32 hash bits behind the region index
        kh::PartGeom g2 = g;
        g2.p1_bits = g.p1_bits + g.p2_bits;
        g2.b2 = 1;
        g2.b2_magic = kh::part_magic_of(1);
        g2.p2_bits = 0;
        g2.defer = 0;
        c->narrow2_refused = false;
        if (c->trace) fprintf(stderr, "[kmerhip] level 2 narrowed the batch's payloads to the %d bits below the region index\n", below_region_bits);
        return finish_batch<uint32_t>(c, g2, reinterpret_cast<uint32_t *>(c->keysB), bend, nregions, n_all, n_ub, ovf_lim, false);
    }
    return finish_batch<PT>(c, g, bufB, bend, nregions, n_all, n_ub, ovf_lim, heavy_exact);
}

int direct_range(kh_ctx *c, const RangeArgs &ra, u64 first_tile, u64 end_tile) {
    {
        int rc = ensure_wide(c);  // (device atomics work on the 16-byte slots)
        if (rc == KH_OK) rc = clear_if_dirty(c);
        if (rc != KH_OK) return rc;
    }
    u64 t = first_tile;
    u64 sub = SUB_TILES;
    while (t < end_tile) {
        u64 nt = std::min(sub, end_tile - t);
        bool smaller = false;
        int rc = ensure_room(c, nt * kh::TILE, nt > SUB_TILES_MIN, &smaller);
        if (rc != KH_OK) return rc;
        if (smaller) {
            sub = std::max(SUB_TILES_MIN, nt / 4);
            continue;
        }
        {
            StageTimer tm(c, ST_DIRECT);
            if (ra.use_qual) launch_count<true>(c, ra.abase, ra.qbase, ra.qaligned, ra.vbeg, ra.vend, ra.wlo, t, nt);
            else launch_count<false>(c, ra.abase, nullptr, 0, ra.vbeg, ra.vend, ra.wlo, t, nt);
        }
        HIP_TRY(c, hipGetLastError());
        c->table_empty = false;
        c->rheads_valid = false;
        c->launches++;
        c->pending_bound += nt * kh::TILE;
        t += nt;
    }
    return KH_OK;
}

// bytes the two partition buffers (and the overflow list) of a batch may take: decided at the context's first partitioned range
void ensure_part_budget(kh_ctx *c) {
    if (c->part_budget) return;
    size_t fr = 0, tot = 0;
    // (up to 0.80 of what is free: the 8-byte table image -- 8 bytes per slot, allocated after level 2 -- and the small arrays
    //  take the rest.  Round 3 stopped at 160 GiB / 0.75: configs[3]'s 125 M reads then ran as two batches, the second one a
    //  pass over a filled table that re-reads and re-writes all of it: 36 ms of region pass where one fresh pass takes 24)
    // (A rank of a multi-GPU merge gets the same: kh_merge_across gives the partition buffers back before it allocates its
    //  send / receive buffers and the shard's 16-byte table -- release_part_buffers.  For an hour of round 4 such a rank kept
    //  0.55 instead, and configs[3]'s share ran as two batches there.)
    u64 budget = 224ull << 30;
    const double share = 0.80;  // (round 5: 0.78 -- two RCCL communicators' worth of device memory tipped configs[3]'s 125 M reads, which need 207.6 of
                                //  the 288 GB by the 11-bytes-per-window rule, into two batches: 100 ms of kernels instead of 80, bench.py --force-merge)
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
        budget = std::min<u64>(budget, (u64)((double)(fr + c->key_cap + c->keyb_cap) * share));
        // (buffers given back for a merge: the table image and the shard's table they left room for exist by now -- the same budget
        //  again where it still fits, so that a rank that counted in one batch goes on counting in one)
        // (... unless kh_set_shard's low-memory path freed the image: its 8 bytes per slot must still fit beside the budget,
        //  or the next count falls back to the 16-byte table and a second batch)
        const u64 image = c->ntab ? 0 : 8ull * c->cap;
        if (c->prev_part_budget > budget && (u64)fr + c->key_cap + c->keyb_cap >= c->prev_part_budget + image + (3ull << 30)) budget = c->prev_part_budget;
    } else (void)hipGetLastError();
    if (c->knobs.part_budget_gb > 0) budget = (u64)(c->knobs.part_budget_gb * (double)(1ull << 30));
    c->part_budget = std::max<u64>(budget, 64ull << 20);
}

// Count all windows of the device buffer [d_bases, d_bases+n) that end at offset >= wlo_off.
int count_device_range(kh_ctx *c, const uint8_t *d_bases, const uint8_t *d_qual, u64 n, u64 wlo_off) {
    if (n == 0) return KH_OK;
    const uintptr_t addr = (uintptr_t)d_bases;
    const u64 lead = addr & 15;
    RangeArgs ra;
    ra.abase = d_bases - lead;
    ra.vbeg = lead;
    ra.vend = lead + n;
    ra.wlo = lead + wlo_off;
    ra.use_qual = (d_qual != nullptr) && (c->minq >= 0);
    ra.qbase = nullptr;
    ra.qaligned = 0;
    if (ra.use_qual) {
        ra.qbase = d_qual - lead;  // same virtual coordinates as the bases
        ra.qaligned = (((uintptr_t)ra.qbase) & 15) == 0;
    }
    const u64 windows = ra.vend - ra.wlo;  // upper bound on k-mers of this range

    // Path choice.  Partitioned cost ~ 32 B per key of HBM traffic + one read and one write of the
    // whole table (32 B per slot); direct cost ~ one memory-side atomic per key (~18.5 G/s), i.e.
    // ~270 B per key at streaming rate.  So partition when the table is < ~7x the batch.
    bool part = false;
    if (c->path_mode == 2) part = true;
    else if (c->path_mode == 0) part = windows >= PART_MIN_WINDOWS && (double)c->cap <= 7.0 * (double)windows;
    if (part) ensure_part_budget(c);
    // (an unmasked range whose every window fits one batch with room to spare needs no estimate of the survivors: the sample
    //  costs a kernel and a host round trip, 0.3 ms of the headline's 68)
    const bool tight = part && (double)windows * 11.0 > 0.85 * (double)c->part_budget;
    if (part && (ra.use_qual || (windows >= (64ull << 20) && (tight || !c->hinted)))) {
        // A quality-masked range: most windows may be gone (-Q 20 on typical reads keeps 0.4 of them at k = 31) -- and so may
        // those of an unhinted one (FASTQ text as the device scanner leaves it: headers and quality lines are masked positions,
        // 0.4 of the windows are k-mers; the table of an unhinted context is sized from the windows).  Count the
        // survivors of every 64th 4096-position tile and size pool, arenas and batches from that instead of from "every
        // window" -- configs[2] then runs as one batch instead of two.  KMERHIP_SURVIVAL=x: use x instead of the sample
        // (tests: a far too small x exercises the retry); =1: size for every window.
        if (c->knobs.survival > 0) {
            ra.survive = std::min(1.0, c->knobs.survival);
        } else {
            const u64 t0 = ra.wlo / kh::TILE, t1 = (ra.vend + kh::TILE - 1) / kh::TILE, stride = 64;
            const u64 nsamp = (t1 - t0 + stride - 1) / stride;
            u64 *d_out = &c->d_ctr->cursor;
            u64 good = 0;
            StageTimer tm(c, ST_MISC);
            HIP_TRY(c, hipMemsetAsync(d_out, 0, sizeof(u64), c->stream));
            if (ra.use_qual)
                hipLaunchKernelGGL(kh::survival_sample_kernel<true>, dim3((unsigned)std::min<u64>(nsamp, 2048)), dim3(kh::BLOCK), 0, c->stream, ra.abase, ra.qbase,
                                   ra.qaligned, ra.vbeg, ra.vend, ra.wlo, t0, t1 - t0, stride, c->k, qual_thr(c), d_out);
            else
                hipLaunchKernelGGL(kh::survival_sample_kernel<false>, dim3((unsigned)std::min<u64>(nsamp, 2048)), dim3(kh::BLOCK), 0, c->stream, ra.abase, (const uint8_t *)nullptr,
                                   0, ra.vbeg, ra.vend, ra.wlo, t0, t1 - t0, stride, c->k, 0u, d_out);
            HIP_TRY(c, hipMemcpyAsync(&good, d_out, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            ra.survive = std::min(1.0, (double)good / (double)(nsamp * kh::TILE));
        }
        if (c->trace) fprintf(stderr, "[kmerhip] %s range: %.3f of the windows expected to survive\n", ra.use_qual ? "quality-masked" : "unhinted", ra.survive);
    }
    // A fresh range with 1024 level-1 partitions ahead of it: partition_batch sizes the table itself, from the distinct keys of
    // a few level-1 partitions, once level 1 has run (round 4) -- hinted or not.  All that is needed here is a table of more
    // than 1024 regions, so that level 1 gets its 10-bit digit.
    const bool sample = part && c->estimate_on && c->table_empty && c->shard_shift == 0 && c->k >= kh::MAX_P1_BITS && windows >= PART_MIN_WINDOWS;
    c->est_keys = 0;
    if (sample && c->cap < 2048ull * kh::REGION_SLOTS) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        resize_empty_table(c, 2048ull * kh::REGION_SLOTS);
    }
    if (part && !c->hinted && !sample) {
        // No capacity hint: this batch may bring up to `windows` NEW keys.  A region pass that overflows
        // falls back to re-inserting the overflowing buckets through device atomics -- correct, but
        // ~30x slower than the pass itself -- so room for the worst case is made first: an empty table
        // is simply re-allocated, a live one rehashed (cheap next to a failed pass).  Never beyond a
        // quarter of the device memory; past that the fallback remains the safety net.
        if (c->pending_bound) {
            int rc = sync_counters(c);
            if (rc != KH_OK) return rc;
        }
        u64 limit = c->cap;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
            const u64 bytes = ((u64)fr + (c->table ? c->cap * sizeof(Slot) : 0)) / 4;
            u64 lim = MIN_CAP;
            while (lim * 2 * sizeof(Slot) <= bytes) lim *= 2;
            limit = std::max(limit, lim);
        }
        // ... the worst case for the first range; after that, half again of what the last range brought per window (reads of
        // one file arrive in file order: the rate changes slowly and mostly falls), at least a 16th of the windows.  A range
        // that brings more overflows some regions and takes the fallback for those.
        u64 expect = sized_for(windows, ra.survive);
        if (c->new_rate >= 0.0) expect = std::min<u64>(expect, (u64)((double)windows * std::max(c->new_rate * 1.5, 1.0 / 16)) + (1u << 20));
        const u64 want = std::min(round_cap((double)(c->distinct_known + expect) / LOAD_PART), limit);
        if (want > c->cap) {
            if (c->table_empty) {
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                resize_empty_table(c, want);
            } else {
                StageTimer t(c, ST_GROW);
                int rc = grow_to(c, want);
                if (rc != KH_OK) c->poisoned = false;
            }
        }
    }
    if (part && !make_geom(c, c->cap).ok) part = false;  // table beyond 2 levels of partitioning

    if (!part) return direct_range(c, ra, ra.wlo / kh::TILE, (ra.vend + kh::TILE - 1) / kh::TILE);

    if (c->pending_bound) {  // exact counters before switching paths
        int rc = sync_counters(c);
        if (rc != KH_OK) return rc;
    }
    if (c->borrow_on) {  // the partition buffers are about to be written: whatever a merge borrowed of them moves out
        int rc = end_borrow(c);
        if (rc != KH_OK) return rc;
    }
    ensure_part_budget(c);
    const u64 first_tile = ra.wlo / kh::PART_TILE;
    const u64 end_tile = (ra.vend + kh::PART_TILE - 1) / kh::PART_TILE;
    const u64 distinct_before = c->distinct_known;
    bool no_narrow = false;  // (KH_RETRY_WIDE: this range's batches are sized for 8-byte level-2 output from here on)
    for (u64 t = first_tile; t < end_tile;) {
        const GeomChoice gc = make_geom(c, c->cap);  // re-evaluated per batch: the table may have grown
        if (!gc.ok) return direct_range(c, ra, t * (kh::PART_TILE / kh::TILE), (ra.vend + kh::TILE - 1) / kh::TILE);
        // bytes per key over the two buffers and the overflow list: pool (1.04 x payload) + arenas (1.25 x + 1) + 1.  8-byte payloads
        // that level 2 will narrow (partition_batch: a fresh table takes a power-of-two size for them; an existing one must be one):
        // 8-byte pool, 4-byte arenas.  A batch sized that way that cannot narrow after all comes back with KH_RETRY_WIDE.
        const int rb_now = (int)kh::kh_floor_log2((uint32_t)std::min<u64>(c->cap / kh::REGION_SLOTS, 1u << 20));
        const bool expect_narrow = !gc.use32 && !no_narrow && !c->narrow2_refused && c->knobs.l2_narrow && c->shard_shift == 0 && c->knobs.l2_arena &&
                                   (c->table_empty ? (sample && 2 * (int)c->k - 20 <= 32) : (cap_is_pow2(c->cap) && rb_now >= 15 && 2 * (int)c->k - rb_now <= 32 && 2 * (int)c->k - rb_now >= 1));
        const u64 per_key = gc.use32 ? 11 : (expect_narrow ? 15 : 20);
        const u64 left = end_tile - t;
        // (a range sized from its survival rate: so many windows per batch that the expected payloads fit the budget)
        const double share = (double)sized_for(left * kh::PART_TILE, ra.survive) / (double)(left * kh::PART_TILE);
        u64 batch_tiles = std::max<u64>(1, (u64)((double)(c->part_budget / per_key / kh::PART_TILE) / share));
        const u64 nb = (left + batch_tiles - 1) / batch_tiles;  // equal-sized batches
        batch_tiles = (left + nb - 1) / nb;
        const u64 nt = std::min(batch_tiles, left);
        GeomChoice gcb = gc;
        const bool from_sample = sample && t == first_tile && gcb.g.p1_bits == kh::MAX_P1_BITS;
        const double range_scale = (double)(end_tile - first_tile) / (double)nt;
        int rc = gcb.use32 ? partition_batch<uint32_t>(c, ra, gcb, t, nt, from_sample, range_scale, false, false)
                           : partition_batch<u64>(c, ra, gcb, t, nt, from_sample, range_scale, expect_narrow, !no_narrow);
        if (rc == KH_RETRY_FULL_SIZE) {  // the sample misjudged these tiles: the rest of the range is sized for every window
            ra.survive = 1.0;
            continue;
        }
        if (rc == KH_RETRY_WIDE) {  // sized for a narrowing level 2 that is not to be: the same tiles again, 20 bytes per window
            no_narrow = true;
            c->narrow2_refused = true;  // (... and this context's later ranges are sized for 8-byte output from the start, until one narrows)
            {   // ... in batches that leave the 16-byte table its room (the budget was drawn before the table had its size)
                size_t fr = 0, tot = 0;
                if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
                    const u64 have = (u64)fr + c->key_cap + c->keyb_cap, tab = c->table ? 0 : c->cap * sizeof(Slot);
                    if (have > tab + (8ull << 30)) c->part_budget = std::min<u64>(c->part_budget, (u64)(0.9 * (double)(have - tab - (8ull << 30))));
                } else (void)hipGetLastError();
            }
            if (c->trace) fprintf(stderr, "[kmerhip] the batch was sized for 4-byte level-2 payloads and cannot have them: again in smaller batches\n");
            continue;
        }
        if (rc != KH_OK) return rc;
        t += nt;
    }
    c->new_rate = (double)(c->distinct_known - distinct_before) / (double)windows;
    return KH_OK;
}

// Staging copy host -> pinned.  One thread moves ~10 GB/s, PCIe takes ~55 GB/s: large copies are split
// over a few short-lived threads (the caller's buffer is pageable memory we cannot DMA from directly).

}  // namespace khi
