// merge.hip -- exports and merges of ONE context: (key, count) pairs by owner, the dense form of small k, and the
// region-ordered units the exchange (exchange.hip) moves between contexts.
#include "ctx.hip.h"
#include "shard.hip.h"

using namespace khi;

// =============================================================================================
// multi-GPU merge support
// =============================================================================================
extern "C" uint32_t kh_owner(uint64_t key, uint32_t k, uint32_t nparts) {
    return (nparts && k >= 1 && k <= 32) ? kh_owner_of(key, k, nparts) : 0;
}

extern "C" int kh_export_by_owner_device(kh_ctx *c, uint32_t nparts, uint64_t *d_keys, uint64_t *d_counts,
                                         uint64_t cap, uint64_t *part_counts) {
    int rc = enter(c);
    if (rc != KH_OK) return rc;
    if (nparts < 1 || nparts > kh::MAX_PARTS || !part_counts) return fail(c, KH_ERR_BAD_ARG, "bad nparts/part_counts");
    rc = sync_counters(c);
    if (rc != KH_OK) return rc;
    u64 *d_parts = nullptr;
    if (hipMalloc((void **)&d_parts, nparts * sizeof(u64)) != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, KH_ERR_OOM, "hipMalloc(parts)");
    }
    std::vector<u64> h(nparts, 0);
    hipError_t e = hipMemsetAsync(d_parts, 0, nparts * sizeof(u64), c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(kh::owner_count_kernel, dim3(grid_for(c->cap)), dim3(kh::BLOCK), 0, c->stream, c->table,
                           c->cap, c->k, nparts, d_parts);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_parts, nparts * sizeof(u64), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    u64 total = 0;
    std::vector<u64> offs(nparts, 0);
    for (uint32_t p = 0; p < nparts; ++p) {
        offs[p] = total;
        total += h[p];
        part_counts[p] = h[p];
    }
    if (e == hipSuccess && total > cap) {
        (void)hipFree(d_parts);
        return fail(c, KH_ERR_RANGE, "export arrays too small");
    }
    if (e == hipSuccess && total) {
        if (!d_keys || !d_counts) {
            (void)hipFree(d_parts);
            return fail(c, KH_ERR_BAD_ARG, "NULL output");
        }
        e = hipMemcpyAsync(d_parts, offs.data(), nparts * sizeof(u64), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(kh::owner_scatter_kernel, dim3(grid_for(c->cap)), dim3(kh::BLOCK), 0, c->stream,
                               c->table, c->cap, c->k, nparts, d_parts, (u64 *)d_keys, (u64 *)d_counts, (u64)cap);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(d_parts);
    if (e != hipSuccess) return fail(c, KH_ERR_HIP, "kh_export_by_owner_device", e);
    return KH_OK;
}

extern "C" int kh_merge_pairs_device(kh_ctx *c, const uint64_t *d_keys, const uint64_t *d_counts, uint64_t n) {
    int rc = enter(c);
    if (rc != KH_OK) return rc;
    if (n == 0) return KH_OK;
    if (!d_keys || !d_counts) return fail(c, KH_ERR_BAD_ARG, "NULL argument");
    const u64 step = SUB_TILES * kh::TILE;
    for (u64 off = 0; off < n;) {
        u64 m = std::min(step, n - off);
        bool smaller = false;
        rc = ensure_room(c, m, false, &smaller);
        if (rc != KH_OK) return rc;
        hipLaunchKernelGGL(kh::table_merge_pairs_kernel, dim3(grid_for(m)), dim3(kh::BLOCK), 0, c->stream,
                           table_geom(c, c->table, c->cap), (const u64 *)d_keys + off, (const u64 *)d_counts + off, m, c->d_ctr);
        HIP_TRY(c, hipGetLastError());
        c->table_empty = false;
        c->rheads_valid = false;
        c->pending_bound += m;
        off += m;
    }
    return KH_OK;
}

extern "C" int kh_merge_pairs(kh_ctx *c, const uint64_t *keys, const uint64_t *counts, uint64_t n) {
    int rc = enter(c);
    if (rc != KH_OK) return rc;
    if (n == 0) return KH_OK;
    if (!keys || !counts) return fail(c, KH_ERR_BAD_ARG, "NULL argument");
    uint64_t *dk = nullptr, *dc = nullptr;
    if (hipMalloc((void **)&dk, n * sizeof(u64)) != hipSuccess || hipMalloc((void **)&dc, n * sizeof(u64)) != hipSuccess) {
        (void)hipGetLastError();
        if (dk) (void)hipFree(dk);
        return fail(c, KH_ERR_OOM, "hipMalloc(merge)");
    }
    hipError_t e = hipMemcpyAsync(dk, keys, n * sizeof(u64), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dc, counts, n * sizeof(u64), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) rc = kh_merge_pairs_device(c, dk, dc, n);
    hipError_t e2 = hipStreamSynchronize(c->stream);
    (void)hipFree(dk);
    (void)hipFree(dc);
    if (e != hipSuccess || e2 != hipSuccess) return fail(c, KH_ERR_HIP, "kh_merge_pairs", e != hipSuccess ? e : e2);
    return rc;
}

// ---- dense form (small k): export for an all-reduce(sum), merge back by owner ---------------------
extern "C" int kh_export_dense_device(kh_ctx *c, uint64_t *d_dense, uint64_t n_entries) {
    int rc = enter(c);
    if (rc != KH_OK) return rc;
    if (2 * c->k > 26) return fail(c, KH_ERR_RANGE, "the dense form needs 2k <= 26");
    if (!d_dense || n_entries != (1ull << (2 * c->k))) return fail(c, KH_ERR_BAD_ARG, "d_dense must hold 4^k entries");
    if (c->shard_shift) return fail(c, KH_ERR_STATE, "table is already a shard");
    HIP_TRY(c, hipMemsetAsync(d_dense, 0, n_entries * sizeof(u64), c->stream));
    hipLaunchKernelGGL(kh::table_to_dense_kernel, dim3(grid_for(c->cap)), dim3(kh::BLOCK), 0, c->stream, (const Slot *)c->table,
                       c->cap, (u64 *)d_dense, (u64)n_entries);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return KH_OK;
}

extern "C" int kh_merge_dense_device(kh_ctx *c, const uint64_t *d_dense, uint64_t n_entries, uint32_t owner, uint32_t nparts) {
    int rc = enter(c);
    if (rc != KH_OK) return rc;
    if (2 * c->k > 26) return fail(c, KH_ERR_RANGE, "the dense form needs 2k <= 26");
    if (!d_dense || n_entries != (1ull << (2 * c->k)) || nparts == 0 || owner >= nparts)
        return fail(c, KH_ERR_BAD_ARG, "bad dense array / owner");
    if (c->shard_shift) return fail(c, KH_ERR_STATE, "a hash-range shard takes kh_merge_regions_*; the dense merge fills a full-geometry table");
    // at most every canonical key is new: 4^k / 2 plus the palindromes
    bool smaller = false;
    rc = ensure_room(c, n_entries / 2 + (1ull << c->k), false, &smaller);
    if (rc != KH_OK) return rc;
    hipLaunchKernelGGL(kh::table_merge_dense_kernel, dim3(grid_for(n_entries)), dim3(kh::BLOCK), 0, c->stream,
                       table_geom(c, c->table, c->cap), (const u64 *)d_dense, (u64)n_entries, owner, nparts, c->d_ctr);
    HIP_TRY(c, hipGetLastError());
    c->table_empty = false;
    c->rheads_valid = false;
    c->pending_bound += n_entries / 2 + (1ull << c->k);
    return sync_counters(c);
}

// ---- hash-range sharding: region-ordered export and LDS merge ------------------------------------
extern "C" int kh_set_shard(kh_ctx *c, uint32_t index, uint32_t count) {
    int rc = enter(c, true, false);  // touches no slot: a lazily reset table stays lazily reset
    if (rc != KH_OK) return rc;
    if (count == 0 || (count & (count - 1)) || index >= count || count > (uint32_t)kh::MAX_SENDERS)
        return fail(c, KH_ERR_BAD_ARG, "shard count must be a power of two (<= 64) and index < count");
    if (!c->table_empty) return fail(c, KH_ERR_STATE, "kh_set_shard needs an empty table (call kh_reset first)");
    uint32_t sh = 0;
    while ((1u << sh) < count) ++sh;
    if (sh >= 2 * c->k) return fail(c, KH_ERR_BAD_ARG, "more shards than k-mers");
    c->shard_shift = sh;
    c->shard_index = index;
    if (c->ntab) {
        // A shard table is never kept as the 8-byte image: where the shard's 16-byte table would not fit beside it, its 8 bytes
        // per slot are the room.  (Not always: a host that counts and merges in a loop would free and allocate the image -- 21 GB
        // at configs[3]'s size -- every round, and large allocations are what stalls on a nearly full device.)
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) {
            (void)hipGetLastError();
            fr = 0;
        }
        if (borrow_room(c) < 16ull * c->cap && (u64)fr < 16ull * (c->cap >> sh) + (8ull << 30)) {  // (round 5: not where the table will be borrowed from the partition buffers)
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            (void)hipFree(c->ntab);
            c->ntab = nullptr;
            c->ntab_cap = 0;
        }
    }
    return KH_OK;
}

namespace khi {


// fmt XF_PACKED64: one u64 per pair into d_keys (d_counts unused); XF_HEADS32: u32 heads into d_keys
// what the export kernels read: the 16-byte table, or its 8-byte image while that holds the counts
kh::SlotSrc slot_src(const kh_ctx *c) {
    kh::SlotSrc s;
    s.table = c->table;
    s.ntab = c->narrow ? c->ntab : nullptr;
    s.geo = c->narrow ? kh::RegionGeom{c->narrow_g.p1_bits, c->narrow_g.b2} : geom_of_cap(c->cap);
    return s;
}

// The region pass of a FRESH batch leaves every region's exchange-head count behind (rheads): a later heads export needs no
// counting pass over the table (4 ms at configs[3]'s size).  The batch's overflow list -- a few entries on well-mixed input --
// is inserted after that pass and changes the counts of the regions it touches: those, and only those, are counted again
// (round 5; until then ANY entry dropped the counts of the whole table).  rtouch, a byte per region, is all zero between uses.
// Two steps around the list's insert, which marks the entries it has applied as consumed (region = ~0): the touched regions are
// noted BEFORE it, counted AFTER it.  (The first version looked at the list after the insert, found every entry consumed, counted
// nothing again -- and kh_merge_across's conservation check refused the export that followed: 157,486 of 15.9 G counts short.)
int mark_touched_regions(kh_ctx *c, const void *ovf_list, const u64 *d_ovf, u64 ovf_lim) {
    hipLaunchKernelGGL(kh::ovf_touch_kernel, dim3(grid_for(c->ovf_pending)), dim3(kh::BLOCK), 0, c->stream, (const kh::OvfEntry *)ovf_list, d_ovf, ovf_lim,
                       c->rtouch);
    HIP_TRY(c, hipGetLastError());
    return KH_OK;
}
int recount_touched_heads(kh_ctx *c, u64 nregions) {
    const int cb = head_count_bits(c, nregions);
    int rc = KH_OK;
    if (cb < 0 || (uint32_t)cb != c->rheads_cb) rc = KH_ERR_RANGE;  // (not the unit the counts were made for: the caller drops them)
    else
        hipLaunchKernelGGL(kh::region_head_count_kernel, dim3((unsigned)nregions), dim3(kh::BLOCK), 0, c->stream, slot_src(c), (uint32_t)cb, c->rheads,
                           &c->d_ctr->heads_wide, (const uint8_t *)c->rtouch);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemsetAsync(c->rtouch, 0, nregions, c->stream));
    return rc;
}

int export_regions(kh_ctx *c, int fmt, uint32_t nparts, void *d_keys, uint64_t *d_counts, uint64_t cap,
                   uint32_t *d_region_counts, uint64_t region_cap, uint64_t *part_counts, uint64_t *table_regions, u64 *d_digest, bool *digest_done) {
    if (digest_done) *digest_done = false;
    int rc = enter(c, true, true, false, fmt != XF_WIDE);  // (packed and heads come straight out of the 8-byte image)
    if (rc != KH_OK) return rc;
    const bool packed = fmt != XF_WIDE;
    const u64 nregions = c->cap / kh::REGION_SLOTS;
    if (fmt == XF_PACKED64 && kh::kh_below_bits(c->k, 0, geom_of_cap(c->cap)) > 32)
        return fail(c, KH_ERR_RANGE, "packed export needs 2k - log2(table regions) <= 32");
    const int cb = fmt == XF_HEADS32 ? head_count_bits(c, nregions) : 0;
    if (cb < 0) return fail(c, KH_ERR_RANGE, "32-bit heads need 1 <= 2k - log2(table regions) <= 28");
    if (table_regions) *table_regions = nregions;
    if (nparts < 1 || nparts > (uint32_t)kh::MAX_SENDERS || (nparts & (nparts - 1)) || nparts > nregions || nregions % nparts || !part_counts ||
        !d_region_counts)
        return fail(c, KH_ERR_BAD_ARG, "bad nparts / NULL argument");
    if (c->shard_shift) return fail(c, KH_ERR_STATE, "table is already a shard");
    if (region_cap < nregions) return fail(c, KH_ERR_RANGE, "region count array too small");
    if (c->win_n > 1 && (nregions / nparts) % c->win_n) return fail(c, KH_ERR_BAD_ARG, "region window: fewer regions per owner than pieces");
    rc = sync_counters(c);
    if (rc != KH_OK) return rc;
    bool counted_by_region_pass = false, keep_counts = false;
    if (fmt == XF_HEADS32 && c->rheads_valid && c->rheads_cb == (uint32_t)cb) {
        // the FRESH region pass that built this table left the head count of every region behind:
        // no counting pass over the 34 GB table
        if (c->rheads_wide) return fail(c, KH_ERR_RANGE, "a count is too large for 32-bit heads");
        HIP_TRY(c, hipMemcpyAsync(d_region_counts, c->rheads, nregions * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
        counted_by_region_pass = true;
    } else if (fmt == XF_HEADS32) {
        rc = zero_cursors(c);
        if (rc != KH_OK) return rc;
        hipLaunchKernelGGL(kh::region_head_count_kernel, dim3((unsigned)nregions), dim3(kh::BLOCK), 0, c->stream,
                           slot_src(c), (uint32_t)cb, d_region_counts, &c->d_ctr->big, (const uint8_t *)nullptr);
        // (kept for the exports that follow -- the other pieces of a pipelined exchange, its unit counts: one pass over the table
        //  instead of one per call; valid once the "too wide" flag below has come back clear, until anything touches the table)
        if (c->rheads && c->region_cap >= nregions) {
            HIP_TRY(c, hipMemcpyAsync(c->rheads, d_region_counts, nregions * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
            keep_counts = true;
        }
    } else {
        hipLaunchKernelGGL(kh::region_live_count_kernel, dim3((unsigned)nregions), dim3(kh::BLOCK), 0, c->stream,
                           slot_src(c), d_region_counts);
    }
    HIP_TRY(c, hipGetLastError());
    if (c->win_n > 1) {
        // One piece of every owner's region range: the other regions count as empty, so the offsets, the
        // per-owner totals and the compaction (which skips empty ranges) all follow.
        hipLaunchKernelGGL(kh::region_window_mask_kernel, dim3(grid_for(nregions)), dim3(kh::BLOCK), 0, c->stream, d_region_counts,
                           nregions, nregions / nparts, (nregions / nparts) / c->win_n, c->win_piece);
        HIP_TRY(c, hipGetLastError());
    }
    // offsets of every region in the export (device scan), and the per-owner totals (host)
    u64 z = c->merge_off_cap;
    rc = ensure_buf(c, &c->merge_off, &z, nregions + 1, "hipMalloc(merge_off)");
    c->merge_off_cap = z;
    if (rc != KH_OK) return rc;
    rc = device_scan(c, d_region_counts, nregions, c->merge_off);
    if (rc != KH_OK) return rc;
    std::vector<u64> bounds(nparts + 1);
    const u64 per = nregions / nparts;
    for (uint32_t p = 0; p <= nparts; ++p)
        HIP_TRY(c, hipMemcpyAsync(&bounds[p], c->merge_off + (u64)p * per, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    if (fmt == XF_HEADS32 && !counted_by_region_pass) {
        u64 wide = 0;
        rc = read_cursor(c, nullptr, &wide);
        if (rc != KH_OK) return rc;
        if (wide) return fail(c, KH_ERR_RANGE, "a count is too large for 32-bit heads");
        if (keep_counts) {
            c->rheads_valid = true;
            c->rheads_wide = false;
            c->rheads_cb = (uint32_t)cb;
        }
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (uint32_t p = 0; p < nparts; ++p) part_counts[p] = bounds[p + 1] - bounds[p];
    const u64 total = bounds[nparts];
    if (total > cap) return fail(c, KH_ERR_RANGE, "export arrays too small");
    if (total && (!d_keys || (!packed && !d_counts))) return fail(c, KH_ERR_BAD_ARG, "NULL output");
    if (total && packed && c->narrow) {
        // out of the 8-byte image (what a rank exports right after its count): shard.hip.h region_compact_image_kernel, which also
        // takes the sender's digest of what it writes where the caller wants one
        const kh::SlotSrc ss = slot_src(c);
        u64 *rdig = nullptr;
        if (d_digest && digest_done && ensure_buf(c, &c->rdig, &c->rdig_cap, 2 * nregions, "hipMalloc(digest partials)") == KH_OK) rdig = c->rdig;
        if (fmt == XF_HEADS32)
            hipLaunchKernelGGL((kh::region_compact_image_kernel<true>), dim3((unsigned)nregions), dim3(kh::BLOCK), 0, c->stream, ss.ntab, ss.geo, (const u64 *)c->merge_off, c->k,
                               (uint32_t)cb, d_keys, rdig);
        else
            hipLaunchKernelGGL((kh::region_compact_image_kernel<false>), dim3((unsigned)nregions), dim3(kh::BLOCK), 0, c->stream, ss.ntab, ss.geo, (const u64 *)c->merge_off, c->k,
                               0u, d_keys, rdig);
        HIP_TRY(c, hipGetLastError());
        if (rdig) {
            HIP_TRY(c, hipMemsetAsync(d_digest, 0, (size_t)3 * nparts * sizeof(u64), c->stream));
            hipLaunchKernelGGL(kh::export_digest_reduce_kernel, dim3((unsigned)std::min<u64>(256, (per + kh::BLOCK - 1) / kh::BLOCK), nparts), dim3(kh::BLOCK), 0, c->stream,
                               (const u64 *)rdig, (const u64 *)c->merge_off, per, d_digest);
            HIP_TRY(c, hipGetLastError());
            *digest_done = true;
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    } else if (total && fmt == XF_HEADS32) {
        hipLaunchKernelGGL(kh::region_compact_heads_kernel, dim3((unsigned)nregions), dim3(kh::BLOCK), 0, c->stream,
                           slot_src(c), (const u64 *)c->merge_off, c->k, (uint32_t)cb,
                           (uint32_t *)d_keys);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    } else if (total && !packed) {
        hipLaunchKernelGGL(kh::region_compact_kernel, dim3((unsigned)nregions), dim3(kh::BLOCK), 0, c->stream,
                           (const Slot *)c->table, (const u64 *)c->merge_off, (u64 *)d_keys, (u64 *)d_counts);  // (XF_WIDE: enter() widened)
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    } else if (total) {
        rc = zero_cursors(c);
        if (rc != KH_OK) return rc;
        hipLaunchKernelGGL(kh::region_compact_packed_kernel, dim3((unsigned)nregions), dim3(kh::BLOCK), 0, c->stream,
                           slot_src(c), (const u64 *)c->merge_off, c->k, (u64 *)d_keys,
                           &c->d_ctr->big);
        HIP_TRY(c, hipGetLastError());
        u64 wide = 0;
        rc = read_cursor(c, nullptr, &wide);
        if (rc != KH_OK) return rc;
        if (wide) return fail(c, KH_ERR_RANGE, "a count does not fit the packed export (>= 2^32)");
    }
    return KH_OK;
}
}  // namespace khi

// Phase one of an export on its own: how many exchange units every region holds (whole range, whatever
// the window) -- what a pipelined exchange needs to announce the sizes of ALL its pieces up front.
extern "C" int kh_region_unit_counts_device(kh_ctx *c, uint32_t unit_bytes, uint32_t *d_region_counts, uint64_t region_cap,
                                            uint64_t *table_regions) {
    int rc = enter(c, true, true, false, true);
    if (rc != KH_OK) return rc;
    if (unit_bytes != 4 && unit_bytes != 8 && unit_bytes != 16) return fail(c, KH_ERR_BAD_ARG, "unit_bytes is 4 (heads), 8 (packed) or 16 (pairs)");
    const u64 nregions = c->cap / kh::REGION_SLOTS;
    if (table_regions) *table_regions = nregions;
    if (!d_region_counts) return fail(c, KH_ERR_BAD_ARG, "NULL argument");
    if (region_cap < nregions) return fail(c, KH_ERR_RANGE, "region count array too small");
    if (c->shard_shift) return fail(c, KH_ERR_STATE, "table is already a shard");
    rc = sync_counters(c);
    if (rc != KH_OK) return rc;
    if (unit_bytes == 4) {
        const int cb = head_count_bits(c, nregions);
        if (cb < 0) return fail(c, KH_ERR_RANGE, "32-bit heads need 1 <= 2k - log2(table regions) <= 28");
        if (c->rheads_valid && c->rheads_cb == (uint32_t)cb) {
            if (c->rheads_wide) return fail(c, KH_ERR_RANGE, "a count is too large for 32-bit heads");
            HIP_TRY(c, hipMemcpyAsync(d_region_counts, c->rheads, nregions * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
        } else {
            rc = zero_cursors(c);
            if (rc != KH_OK) return rc;
            hipLaunchKernelGGL(kh::region_head_count_kernel, dim3((unsigned)nregions), dim3(kh::BLOCK), 0, c->stream,
                               slot_src(c), (uint32_t)cb, d_region_counts, &c->d_ctr->big, (const uint8_t *)nullptr);
            HIP_TRY(c, hipGetLastError());
            const bool keep = c->rheads && c->region_cap >= nregions;
            if (keep) HIP_TRY(c, hipMemcpyAsync(c->rheads, d_region_counts, nregions * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
            u64 wide = 0;
            rc = read_cursor(c, nullptr, &wide);
            if (rc != KH_OK) return rc;
            if (wide) return fail(c, KH_ERR_RANGE, "a count is too large for 32-bit heads");
            if (keep) {  // (as in export_regions)
                c->rheads_valid = true;
                c->rheads_wide = false;
                c->rheads_cb = (uint32_t)cb;
            }
        }
    } else {
        hipLaunchKernelGGL(kh::region_live_count_kernel, dim3((unsigned)nregions), dim3(kh::BLOCK), 0, c->stream,
                           slot_src(c), d_region_counts);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return KH_OK;
}

extern "C" int kh_set_region_window(kh_ctx *c, uint32_t piece, uint32_t npieces) {
    if (!c) return KH_ERR_BAD_ARG;
    if (npieces < 1 || npieces > 64 || (npieces & (npieces - 1)) || piece >= npieces)
        return fail(c, KH_ERR_BAD_ARG, "region window: npieces must be a power of two <= 64, piece < npieces");
    c->win_piece = piece;
    c->win_n = npieces;
    return KH_OK;
}

extern "C" int kh_export_regions_device(kh_ctx *c, uint32_t nparts, uint64_t *d_keys, uint64_t *d_counts, uint64_t cap,
                                        uint32_t *d_region_counts, uint64_t region_cap, uint64_t *part_counts,
                                        uint64_t *table_regions) {
    return export_regions(c, XF_WIDE, nparts, d_keys, d_counts, cap, d_region_counts, region_cap, part_counts, table_regions);
}

extern "C" int kh_export_regions_packed_device(kh_ctx *c, uint32_t nparts, uint64_t *d_pairs, uint64_t cap,
                                               uint32_t *d_region_counts, uint64_t region_cap, uint64_t *part_counts,
                                               uint64_t *table_regions) {
    return export_regions(c, XF_PACKED64, nparts, d_pairs, nullptr, cap, d_region_counts, region_cap, part_counts, table_regions);
}

extern "C" int kh_export_regions_heads_device(kh_ctx *c, uint32_t nparts, uint32_t *d_heads, uint64_t cap,
                                              uint32_t *d_region_counts, uint64_t region_cap, uint64_t *part_counts,
                                              uint64_t *table_regions) {
    if (c) c->exports_seen = true;  // (from now on this context's fresh passes leave the head counts behind: batch.hip want_heads)
    return export_regions(c, XF_HEADS32, nparts, d_heads, nullptr, cap, d_region_counts, region_cap, part_counts, table_regions);
}

namespace khi {
int merge_regions(kh_ctx *c, int fmt, uint32_t nsenders, uint64_t sender_regions, const void *const *d_keys,
                  const uint64_t *const *d_counts, const uint32_t *const *d_region_counts, u64 *d_digest, bool *digest_done) {
    if (digest_done) *digest_done = false;
    // a FRESH merge rewrites every region of a lazily reset table; in pieces (kh_set_region_window), the
    // pieces still to come stay unwritten until then (win_open)
    const bool windowed = c && c->win_n > 1;
    // (narrow_ok: a fresh merge of packed pairs / heads builds the shard as the 8-byte image, piece by piece -- decided below)
    int rc = enter(c, true, false, windowed && c->win_open && c->win_open_n == c->win_n && !(c->win_mask & (1ull << c->win_piece)), true);
    if (rc != KH_OK) return rc;
    const bool packed = fmt != XF_WIDE;
    if (nsenders < 1 || nsenders > (uint32_t)kh::MAX_SENDERS || !d_keys || (!packed && !d_counts) || !d_region_counts)
        return fail(c, KH_ERR_BAD_ARG, "bad nsenders / NULL argument");
    // The senders' tables: any geometry a table can have (a power of two, or 1024 x b2 regions) whose regions split evenly
    // among the shards -- for 1024 x b2 that means b2 is a multiple of the shard count: a shard's range of sender regions then
    // nests in any receiver table of nr x 2^d regions (target t <-> sender-local region t >> d), exactly as bit fields do
    // for powers of two.  (Proof sketch: with C = b2 / shards, the sender-local region of a key is p1' C + floor(xr C / 2^(32 - s)),
    // p1' and xr being the shard table's own level-1 digit and the bits behind it; a receiver with C 2^d buckets per p1' has
    // t = p1' C 2^d + floor(xr C 2^d / 2^(32 - s)), and t >> d is the former.)
    const kh::RegionGeom sgeo = kh::kh_geom_of_regions(sender_regions);
    if (!kh::kh_regions_valid(sender_regions) || (sender_regions >> c->shard_shift) == 0 || (sender_regions & ((1ull << c->shard_shift) - 1)) ||
        (sgeo.b2 > 1 && (sgeo.b2 & (sgeo.b2 - 1)) && sgeo.b2 % (1u << c->shard_shift)))
        return fail(c, KH_ERR_BAD_ARG, "sender_regions must be a table geometry (a power of two, or a multiple of 1024) that splits evenly among the shards");
    const u64 nr = sender_regions >> c->shard_shift;  // sender regions inside this shard's hash range
    if (windowed && nr % c->win_n) return fail(c, KH_ERR_BAD_ARG, "region window: fewer sender regions in the shard than pieces");
    // per-sender offsets of every region segment (device scans), and the incoming total (host)
    u64 z = c->merge_off_cap;
    rc = ensure_buf(c, &c->merge_off, &z, (u64)nsenders * (nr + 1), "hipMalloc(merge_off)");
    c->merge_off_cap = z;
    if (rc != KH_OK) return rc;
    std::vector<u64> totals(nsenders);
    for (uint32_t s = 0; s < nsenders; ++s) {
        rc = device_scan(c, d_region_counts[s], nr, c->merge_off + (u64)s * (nr + 1));
        if (rc != KH_OK) return rc;
        HIP_TRY(c, hipMemcpyAsync(&totals[s], c->merge_off + (u64)s * (nr + 1) + nr, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    u64 incoming = 0;
    for (u64 t : totals) incoming += t;
    if (incoming == 0) return KH_OK;
    // every incoming pair may be a new key: make room up front (an empty table is simply re-allocated)
    if (c->pending_bound) {
        rc = sync_counters(c);
        if (rc != KH_OK) return rc;
    }
    // a first piece sizes for all of them (pieces are equal shares of the hash range)
    const u64 expect = (windowed && c->table_empty) ? incoming * c->win_n : incoming;
    // the receiver's table must NEST with the senders' regions: nr x 2^d regions for some d (negative: coarser)
    auto nests = [&](u64 cap) {
        const u64 nt = cap / kh::REGION_SLOTS;
        if (!kh::kh_regions_valid(nt)) return false;
        const u64 hi = std::max(nt, nr), lo = std::min(nt, nr);
        return hi % lo == 0 && ((hi / lo) & (hi / lo - 1)) == 0;
    };
    if ((double)(c->distinct_known + expect) > LOAD_HARD * (double)c->cap || !nests(c->cap)) {
        const double need = (double)(c->distinct_known + expect) / LOAD_HARD;
        u64 newcap = nr * kh::REGION_SLOTS;
        while ((double)newcap < need || newcap < c->cap || newcap < MIN_CAP) newcap *= 2;
        while (newcap / 2 >= MIN_CAP && (double)(newcap / 2) >= need && newcap / 2 >= c->cap && nests(newcap / 2)) newcap /= 2;
        if (c->win_open) {  // growing rehashes the whole table: the unwritten pieces must be empty first
            rc = close_fresh_window(c);
            if (rc != KH_OK) return rc;
        }
        if (c->table_empty) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            resize_empty_table(c, newcap);
        } else {
            rc = grow_to(c, newcap);
            if (rc != KH_OK) return rc;
        }
    }
    // ---- the shard as the 8-byte image (round 6; shard.hip.h shard_merge_narrow_kernel) ----
    // A FRESH merge of packed pairs or heads -- all of it, or the pieces of a window one after the other -- into a table whose
    // 32-bit payload (the hash bits behind its level-1 digit) holds everything the region index does not: count << 32 | payload
    // per slot, what a partitioned count leaves behind too; the readers take it as it is, everything else widens it (enter()).
    // The count's own image is the room: every export of this merge has been made by now.
    const bool fresh_now = windowed ? (c->table_empty || (c->win_open && !(c->win_mask & (1ull << c->win_piece)))) : c->table_empty;
    const kh::RegionGeom rgc = geom_of_cap(c->cap);
    const int xbits = 2 * (int)c->k - (int)c->shard_shift - (int)rgc.p1_bits;
    // ... and the geometry the kernel's 32-bit arithmetic covers (shard.hip.h NarrowK): targets no coarser than the senders' regions, the
    // shard's level-1 digit ending inside or at the senders' x, no hash bits behind that x -- every table of a real exchange; the rest
    // (receivers far smaller than the senders, tiny tables) goes through the 16-byte form as before
    const int o_bits = (int)c->shard_shift + (int)rgc.p1_bits - (int)sgeo.p1_bits;
    const bool geo_fast = c->cap / kh::REGION_SLOTS >= nr && o_bits >= 0 && o_bits < 32 && 2 * (int)c->k - (int)sgeo.p1_bits <= 32 && sgeo.b2 <= 1024 &&
                          sender_regions < (1ull << 22) && c->cap / kh::REGION_SLOTS < (1ull << 22) && nsenders <= kh::SHARD_SEGS;
    bool nar = packed && fresh_now && c->knobs.narrow && !c->narrow_banned && xbits >= 1 && xbits <= 32 && geo_fast && (c->table_empty || c->narrow);
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
    if (!nar && c->narrow) {  // (a merge into a shard that IS an image: units it cannot take that way -- wide pairs, a second merge)
        if ((rc = close_fresh_window(c)) != KH_OK || (rc = ensure_wide(c)) != KH_OK) return rc;
    }
    if (!nar && (rc = need_table(c)) != KH_OK) return rc;  // (uninitialised if new: a fresh merge writes every region -- `dirty` below)
    const kh::TableGeom tg = table_geom(c, c->table, c->cap);
    const u64 nregions = c->cap / kh::REGION_SLOTS;
    if ((rc = ensure_region_scratch(c, nregions)) != KH_OK) return rc;
    kh::MergeArgs a;
    memset(&a, 0, sizeof(a));
    a.nsenders = nsenders;
    a.dshift = 0;  // target t <-> sender-local region t >> dshift (nests(): the ratio is a power of two)
    for (u64 q = nregions; q > nr; q >>= 1) ++a.dshift;
    for (u64 q = nr; q > nregions; q >>= 1) --a.dshift;
    for (uint32_t s = 0; s < nsenders; ++s) {
        a.src[s].keys = (const u64 *)d_keys[s];
        a.src[s].counts = packed ? nullptr : (const u64 *)d_counts[s];
        a.src[s].off = c->merge_off + (u64)s * (nr + 1);
    }
    a.sgeo = sgeo;
    a.src_region0 = (u64)c->shard_index * nr;
    if (fmt == XF_PACKED64 && kh::kh_below_bits(c->k, 0, sgeo) > 32) return fail(c, KH_ERR_BAD_ARG, "packed pairs need 2k - log2(sender_regions) <= 32");
    if (fmt == XF_HEADS32) {
        const int cb = head_count_bits(c, sender_regions);
        if (cb < 0) return fail(c, KH_ERR_BAD_ARG, "32-bit heads need 1 <= 2k - log2(sender_regions) <= 28");
        a.head_cmask = (1u << cb) - 1u;
    }
    // the target regions this call covers: all of them, or the window's contiguous share
    u64 region0 = 0, nwin = nregions;
    bool fresh = c->table_empty;
    if (windowed) {
        // (targets coarser than the senders' regions are fine: nr / win_n sender regions are then still
        // whole target regions, as both counts are powers of two and nregions >= win_n)
        if (nregions < c->win_n) return fail(c, KH_ERR_BAD_ARG, "region window: the shard table has fewer regions than pieces");
        nwin = nregions / c->win_n;
        region0 = (u64)c->win_piece * nwin;
        if (c->table_empty) {  // first piece of a FRESH merge
            c->win_open = true;
            c->win_open_n = c->win_n;
            c->win_mask = 0;
            c->win_dirty = nar ? true : c->table_dirty;  // (the image holds the slots of the count that was exported: always stale)
        }
        fresh = c->win_open && !(c->win_mask & (1ull << c->win_piece));  // (enter() closed a window this piece does not fit)
    }
    {
        StageTimer t(c, ST_REGION);
        const dim3 mg((unsigned)nwin), mb(1024);
        const uint8_t *none = nullptr;
        const uint32_t dirty = (uint32_t)(windowed ? (fresh && c->win_dirty) : c->table_dirty);
#define KH_MERGE_LAUNCH(FRESH, FMT) \
    hipLaunchKernelGGL((kh::shard_merge_kernel<FRESH, false, FMT>), mg, mb, 0, c->stream, tg, a, c->rfail, c->rnew, c->radd, none, kh::RegionGeom{0u, 1u}, c->d_ctr, \
                       FRESH ? dirty : 0u, (uint32_t)region0)
        if (nar) {
            kh::PartGeom ng;
            memset(&ng, 0, sizeof(ng));
            ng.p1_bits = tg.p1_bits;
            ng.b2 = tg.b2;
            ng.b2_magic = kh::part_magic_of(tg.b2);
            ng.p2_bits = kh::part_p2_bits_of(tg.b2);
            ng.k = c->k;
            ng.shard_shift = c->shard_shift;
            ng.shard_index = c->shard_index;
            // a workgroup walks 2^gshift consecutive targets (shard.hip.h): up to eight, as long as their segment bounds fit the kernel's
            // table and the grid keeps every CU busy
            kh::NarrowK K;
            memset(&K, 0, sizeof(K));
            while (K.gshift < 3 && nwin % (2ull << K.gshift) == 0 && (2u << K.gshift) * nsenders <= kh::SHARD_SEGS && nwin >> (K.gshift + 1) >= 2048) ++K.gshift;
            K.nsenders = nsenders;
            if ((nsenders & (nsenders - 1)) == 0)
                while ((nsenders << (K.pshift + 1)) <= (uint32_t)kh::SHARD_NT / 64) ++K.pshift;
            K.dshift = (uint32_t)a.dshift;
            K.b2r = tg.b2;
            K.b2s = sgeo.b2;
            K.magic_r = kh::part_magic_of(tg.b2);
            K.magic_s = kh::part_magic_of(sgeo.b2);
            K.sw_shr = (32u - kh::kh_below_w(sgeo.b2)) & 31u;
            K.o = (uint32_t)o_bits;
            K.o_shr = (32u - K.o) & 31u;
            K.omask = (1u << K.o) - 1u;
            K.rmask = tg.p1_bits ? (1u << tg.p1_bits) - 1u : 0u;
            K.cmask = a.head_cmask;
            K.zs_mask = (1u << kh::kh_x_zero_bits(c->k, sgeo.p1_bits)) - 1u;
            K.stepQ = (1ull << 32) / sgeo.b2;
            K.stepR = (uint32_t)((1ull << 32) - K.stepQ * sgeo.b2);
            K.src_region0 = a.src_region0;
            K.region0 = (uint32_t)region0;
            const dim3 mgn((unsigned)(nwin >> K.gshift));
            // the arrival digests on the way (exchange.hip): per-workgroup partial sums, folded by a small kernel
            u64 *rdig = nullptr;
            const uint32_t cols = 3 * nsenders;
            if (d_digest && digest_done && ensure_buf(c, &c->rdig, &c->rdig_cap, (nwin >> K.gshift) * cols, "hipMalloc(digest partials)") == KH_OK) rdig = c->rdig;
            if (fmt == XF_PACKED64)
                hipLaunchKernelGGL((kh::shard_merge_narrow_kernel<1>), mgn, dim3(kh::SHARD_NT), 0, c->stream, K, a, c->ntab, c->rfail, c->rnew, c->radd, rdig);
            else
                hipLaunchKernelGGL((kh::shard_merge_narrow_kernel<2>), mgn, dim3(kh::SHARD_NT), 0, c->stream, K, a, c->ntab, c->rfail, c->rnew, c->radd, rdig);
            const uint32_t gshift = K.gshift;
            if (rdig) {
                HIP_TRY(c, hipMemsetAsync(d_digest, 0, cols * sizeof(u64), c->stream));
                const unsigned bs = cols * std::max(1u, 256u / cols);
                hipLaunchKernelGGL(kh::digest_reduce_kernel, dim3((unsigned)std::min<u64>(1024, ((nwin >> gshift) * cols + bs - 1) / bs)), dim3(bs), 0, c->stream, (const u64 *)rdig,
                                   (u64)(nwin >> gshift), cols, d_digest);
                *digest_done = true;
            }
            c->narrow = true;
            c->narrow_g = ng;
        } else if (fresh) {
            if (fmt == XF_WIDE) KH_MERGE_LAUNCH(true, 0);
            else if (fmt == XF_PACKED64) KH_MERGE_LAUNCH(true, 1);
            else KH_MERGE_LAUNCH(true, 2);
        } else {
            if (fmt == XF_WIDE) KH_MERGE_LAUNCH(false, 0);
            else if (fmt == XF_PACKED64) KH_MERGE_LAUNCH(false, 1);
            else KH_MERGE_LAUNCH(false, 2);
        }
#undef KH_MERGE_LAUNCH
        hipLaunchKernelGGL(kh::shard_reduce_kernel, dim3(grid_for(nwin)), dim3(kh::BLOCK), 0, c->stream,
                           (const uint8_t *)c->rfail + region0, (const uint32_t *)c->rnew + region0, (const u64 *)c->radd + region0, (u64)nwin, c->d_ctr);
    }
    HIP_TRY(c, hipGetLastError());
    c->table_empty = false;
    c->rheads_valid = false;
    c->table_dirty = false;
    if (windowed && c->win_open) {
        c->win_mask |= 1ull << c->win_piece;
        if (c->win_mask == (c->win_open_n == 64 ? ~0ull : (1ull << c->win_open_n) - 1)) c->win_open = false;  // every region written
    }
    rc = sync_counters(c);
    if (rc != KH_OK) return rc;
    if (c->h_ctr->part_failed) {  // some target regions overflowed: grow, then insert their pairs directly
        const kh::RegionGeom old_geo{tg.p1_bits, tg.b2};
        StageTimer t(c, ST_GROW);
        rc = close_fresh_window(c);  // growing rehashes the whole table
        if (rc != KH_OK) return rc;
        rc = grow_to(c, c->cap * 2);
        if (rc != KH_OK) return rc;
#define KH_MERGE_DIRECT(FMT) \
    hipLaunchKernelGGL((kh::shard_merge_kernel<false, true, FMT>), dim3((unsigned)nwin), dim3(1024), 0, c->stream, \
                       table_geom(c, c->table, c->cap), a, c->rfail, c->rnew, c->radd, (const uint8_t *)c->rfail, old_geo, c->d_ctr, 0u, \
                       (uint32_t)region0)
        if (fmt == XF_WIDE) KH_MERGE_DIRECT(0);
        else if (fmt == XF_PACKED64) KH_MERGE_DIRECT(1);
        else KH_MERGE_DIRECT(2);
#undef KH_MERGE_DIRECT
        HIP_TRY(c, hipMemsetAsync(&c->d_ctr->part_failed, 0, sizeof(u64), c->stream));
        HIP_TRY(c, hipGetLastError());
        rc = sync_counters(c);
        if (rc != KH_OK) return rc;
    }
    return KH_OK;
}
}  // namespace khi

extern "C" int kh_merge_regions_device(kh_ctx *c, uint32_t nsenders, uint64_t sender_regions,
                                       const uint64_t *const *d_keys, const uint64_t *const *d_counts,
                                       const uint32_t *const *d_region_counts) {
    return merge_regions(c, XF_WIDE, nsenders, sender_regions, (const void *const *)d_keys, d_counts, d_region_counts);
}

extern "C" int kh_merge_regions_packed_device(kh_ctx *c, uint32_t nsenders, uint64_t sender_regions,
                                              const uint64_t *const *d_pairs, const uint32_t *const *d_region_counts) {
    return merge_regions(c, XF_PACKED64, nsenders, sender_regions, (const void *const *)d_pairs, nullptr, d_region_counts);
}

extern "C" int kh_merge_regions_heads_device(kh_ctx *c, uint32_t nsenders, uint64_t sender_regions,
                                             const uint32_t *const *d_heads, const uint32_t *const *d_region_counts) {
    return merge_regions(c, XF_HEADS32, nsenders, sender_regions, (const void *const *)d_heads, nullptr, d_region_counts);
}

