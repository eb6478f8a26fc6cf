// kmerhip.hip -- C-ABI implementation (include/kmerhip.h) over the gfx950 kernels.
//
// Host-side orchestration of the device path that replaces KmerMap::build /
// build_with_quality / into_hashmap (reference src/run.rs:494-582).  No CPU fallback exists:
// every counting entry point needs a HIP device and fails with KH_ERR_NO_DEVICE otherwise.
#include "ctx.hip.h"

namespace khi {

void read_knobs(Knobs &k) {
    k = Knobs();
    if (const char *e = env_of("KMERHIP_TRACE")) k.trace = e[0] != '0';
    if (const char *e = env_of("KMERHIP_PATH")) k.path = !strcmp(e, "direct") ? 1 : !strcmp(e, "partition") ? 2 : 0;
    if (const char *e = env_of("KMERHIP_PART_BUDGET_GB")) k.part_budget_gb = atof(e);
    if (const char *e = env_of("KMERHIP_ACC_MAX_MB")) k.acc_max_mb = strtoull(e, nullptr, 10);
    if (const char *e = env_of("KMERHIP_TEXT_ACC_MB")) k.text_acc_mb = strtoull(e, nullptr, 10);
    if (const char *e = env_of("KMERHIP_COPY_THREADS")) k.copy_threads = atoi(e);
    if (const char *e = env_of("KMERHIP_ESTIMATE")) k.estimate = e[0] != '0';
    if (const char *e = env_of("KMERHIP_POW2_TABLE")) k.pow2_table = e[0] == '1';
#if KH_TESTING
    if (const char *e = env_of("KMERHIP_PAYLOAD")) k.payload = atoi(e);
    if (const char *e = env_of("KMERHIP_TABLE_REGIONS")) k.table_regions = strtoull(e, nullptr, 10);
    if (const char *e = env_of("KMERHIP_REGION_NT")) k.region_nt = atoi(e);
    if (const char *e = env_of("KMERHIP_P2_FORCE_WIDE")) k.p2_force_wide = e[0] == '1';
    if (const char *e = env_of("KMERHIP_GENERIC_K")) k.generic_k = e[0] != '0';
    if (const char *e = env_of("KMERHIP_P1_BINS")) k.p1_legacy = e[0] == '0';
    if (const char *e = env_of("KMERHIP_P2_LINES")) k.p2_lines = e[0] != '0';
    if (const char *e = env_of("KMERHIP_L2_ARENA")) k.l2_arena = e[0] != '0';
    if (const char *e = env_of("KMERHIP_L2_OVF_CAP")) k.l2_ovf_cap = strtoull(e, nullptr, 10);
    if (const char *e = env_of("KMERHIP_L2_SKEW_X")) k.l2_skew_x = atoi(e);
    if (const char *e = env_of("KMERHIP_L2_HEAVY_ROOM")) k.l2_heavy_room = strtoull(e, nullptr, 10);
    if (const char *e = env_of("KMERHIP_NARROW")) k.narrow = e[0] != '0';
    if (const char *e = env_of("KMERHIP_L2_NARROW")) k.l2_narrow = e[0] != '0';
    k.l2_no_room_wide = env_of("KMERHIP_L2_NO_ROOM_WIDE") != nullptr;
    if (const char *e = env_of("KMERHIP_HOT_CUT")) k.hot_cut = (e[0] == '0' && !e[1]) ? ~0ull : strtoull(e, nullptr, 10);
    if (const char *e = env_of("KMERHIP_OVF_AGG")) k.ovf_agg = atoi(e);
    if (const char *e = env_of("KMERHIP_SURVIVAL")) k.survival = atof(e);
    if (const char *e = env_of("KMERHIP_TABLE_ROOM_MB")) k.table_room_mb = strtoull(e, nullptr, 10);
    k.heads_always = env_of("KMERHIP_HEADS_ALWAYS") != nullptr;
    k.stop_after_p1 = env_of("KMERHIP_STOP_AFTER_P1") != nullptr;
    k.stop_after_p2 = env_of("KMERHIP_STOP_AFTER_P2") != nullptr;
#endif
}

int enter(kh_ctx *c, bool flush_pending, bool need_table, bool keep_window, bool narrow_ok) {
    if (!c) return KH_ERR_BAD_ARG;
#if KH_TESTING
    read_knobs(c->knobs);  // (tests flip the switches between calls on one context; the product library reads them once, at kh_create)
#endif
    if (c->poisoned) return fail(c, KH_ERR_STATE, "context is poisoned by an earlier error");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->win_open && !keep_window) {
        int rc = close_fresh_window(c);
        if (rc != KH_OK) return rc;
    }
    if (flush_pending && c->acc_len) {
        int rc = flush_acc(c, false);
        if (rc != KH_OK) return rc;
    }
    if (flush_pending && c->txt_unscanned.on) {  // (KH_FLAG_DEFER_TEXT_SCAN: the last text's scan -- and its verdict -- is still to come)
        int rc = scan_unscanned(c);
        if (rc != KH_OK) return rc;
    }
    if (flush_pending && c->txt_acc_len) {
        int rc = flush_text(c);
        if (rc != KH_OK) return rc;
    }
    if (c->narrow && !narrow_ok) {
        int rc = ensure_wide(c);
        if (rc != KH_OK) return rc;
    }
    return (need_table && !c->narrow) ? clear_if_dirty(c) : KH_OK;
}

int alloc_table(kh_ctx *c, u64 cap, Slot **out) {
    Slot *t = nullptr;
    hipError_t e = hipMalloc((void **)&t, cap * sizeof(Slot));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, KH_ERR_OOM, "hipMalloc(table)", e);
    }
    hipLaunchKernelGGL(kh::table_init_kernel, dim3(grid_for(cap)), dim3(kh::BLOCK), 0, c->stream, t, cap);
    HIP_TRY(c, hipGetLastError());
    *out = t;
    return KH_OK;
}

// The 16-byte table exists only once something needs it (round 4).  A context whose batches all go through the partitioned
// path with 32-bit payloads keeps its counts in the 8-byte image (ntab) from the first pass to kh_finish / kh_histogram /
// kh_lookup / the packed exports: 16 bytes per slot -- 34 GB at the headline's size, 43 GB at configs[3]'s -- that are then
// never allocated, never initialised, and are room for one batch's partition buffers instead of two batches'.
// Allocated here it is uninitialised (table_dirty): whoever needs it cleared, clears it (clear_if_dirty); a widening or a
// fresh region pass writes every slot anyway.
int need_table(kh_ctx *c) {
    if (c->table) return KH_OK;
    if (void *p = borrow(c, c->cap * sizeof(Slot))) {  // (a merge's shard table, inside the idle partition buffers: see ctx.hip.h)
        c->table = (Slot *)p;
        c->table_borrowed = true;
        c->table_dirty = true;
        return KH_OK;
    }
    hipError_t e = hipMalloc((void **)&c->table, c->cap * sizeof(Slot));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c->table = nullptr;
        return fail(c, KH_ERR_OOM, "hipMalloc(table)", e);
    }
    c->table_dirty = true;
    return KH_OK;
}
void drop_table(kh_ctx *c) {
    if (c->table && !c->table_borrowed) (void)hipFree(c->table);  // (synchronises the device)
    c->table = nullptr;
    c->table_borrowed = false;
}
// ---- the idle partition buffers as a merge's scratch (ctx.hip.h, borrow_on) ----
u64 borrow_room(const kh_ctx *c) {
    if (!c->borrow_on) return 0;
    const u64 a = c->keysA && c->key_cap > c->borrow_off[0] ? c->key_cap - c->borrow_off[0] : 0;
    const u64 b = c->keysB && c->keyb_cap > c->borrow_off[1] ? c->keyb_cap - c->borrow_off[1] : 0;
    return std::max(a, b) & ~4095ull;
}
void *borrow(kh_ctx *c, u64 bytes) {
    if (!c->borrow_on) return nullptr;
    const u64 need = (std::max<u64>(bytes, 16) + 4095) & ~4095ull;
    uint8_t *const base[2] = {c->keysA, c->keysB};
    const u64 cap[2] = {c->key_cap, c->keyb_cap};
    // the smaller fit first: the large buffer stays whole for the large request (the table)
    int order[2] = {0, 1};
    if (cap[0] - std::min(cap[0], c->borrow_off[0]) > cap[1] - std::min(cap[1], c->borrow_off[1])) std::swap(order[0], order[1]);
    for (int i : order) {
        if (!base[i] || c->borrow_off[i] + need > cap[i]) continue;
        void *p = base[i] + c->borrow_off[i];
        c->borrow_off[i] += need;
        return p;
    }
    return nullptr;
}
int end_borrow(kh_ctx *c) {
    if (c->table_borrowed && c->table) {  // (rare: a table built by a merge is still wanted while a new count starts, or the buffers go)
        Slot *nt = nullptr;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (hipMalloc((void **)&nt, c->cap * sizeof(Slot)) != hipSuccess) {
            (void)hipGetLastError();
            return fail(c, KH_ERR_OOM, "hipMalloc(table leaving the partition buffers)");
        }
        HIP_TRY(c, hipMemcpyAsync(nt, c->table, c->cap * sizeof(Slot), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->table = nt;
    }
    c->table_borrowed = false;
    c->borrow_on = false;
    c->borrow_off[0] = c->borrow_off[1] = 0;
    return KH_OK;
}
// an EMPTY table of another size: nothing to move, nothing to allocate yet
void resize_empty_table(kh_ctx *c, u64 newcap) {
    drop_table(c);
    if (c->ntab) {
        (void)hipFree(c->ntab);
        c->ntab = nullptr;
        c->ntab_cap = 0;
    }
    c->cap = newcap;
    c->table_dirty = false;
    c->rheads_valid = false;
}

int clear_if_dirty(kh_ctx *c) {
    const int nrc = need_table(c);
    if (nrc != KH_OK) return nrc;
    if (!c->table_dirty) return KH_OK;
    hipLaunchKernelGGL(kh::table_init_kernel, dim3(grid_for(c->cap)), dim3(kh::BLOCK), 0, c->stream, c->table, c->cap);
    HIP_TRY(c, hipGetLastError());
    c->table_dirty = false;
    return KH_OK;
}

// A FRESH merge in pieces (kh_set_region_window) was interrupted, or needs the table as a whole (growth):
// the regions of the pieces not merged yet hold whatever the lazily reset table held -- make them empty.
int close_fresh_window(kh_ctx *c) {
    if (!c->win_open) return KH_OK;
    c->win_open = false;
    if (!c->win_dirty) return KH_OK;
    const u64 per_piece = c->cap / c->win_open_n;  // slots; a piece is a contiguous range of target regions
    for (uint32_t p = 0; p < c->win_open_n; ++p) {
        if (c->win_mask & (1ull << p)) continue;
        if (c->narrow)  // (a shard built as the 8-byte image, merge.hip: 0 = a free slot)
            HIP_TRY(c, hipMemsetAsync(c->ntab + (u64)p * per_piece, 0, per_piece * sizeof(u64), c->stream));
        else
            hipLaunchKernelGGL(kh::table_init_kernel, dim3(grid_for(per_piece)), dim3(kh::BLOCK), 0, c->stream,
                               c->table + (u64)p * per_piece, per_piece);
    }
    HIP_TRY(c, hipGetLastError());
    return KH_OK;
}

// The 8-byte image -> the 16-byte table (every slot of it is rewritten: a lazily reset wide table needs no clearing first).
int ensure_wide(kh_ctx *c) {
    if (!c->narrow) return KH_OK;
    {
        const int nrc = need_table(c);
        if (nrc != KH_OK) return nrc;
    }
    hipLaunchKernelGGL(kh::ntable_widen_kernel, dim3(grid_for(c->cap)), dim3(kh::BLOCK), 0, c->stream, (const u64 *)c->ntab, c->cap,
                       c->narrow_g, c->table);
    HIP_TRY(c, hipGetLastError());
    c->narrow = false;
    c->table_dirty = false;
    if (c->trace) fprintf(stderr, "[kmerhip] 8-byte table image widened to %llu 16-byte slots\n", c->cap);
    return KH_OK;
}

// Blocks until the stream is idle and refreshes the exact counters.
int sync_counters(kh_ctx *c) {
    HIP_TRY(c, hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->distinct_known = c->h_ctr->distinct;
    c->pending_bound = 0;
    if (c->h_ctr->failed) return fail(c, KH_ERR_TABLE_FULL, "an upsert found no free slot");
    return KH_OK;
}

// The smallest table of at least `want` slots.  Up to 1024 regions: a power of two.  Beyond: 1024 x b2 regions
// (kernels.hip.h TableGeom) with b2 in steps of at most an eighth -- every multiple of 8 from 64, of 16 from 128, of 32 from
// 256, of 64 from 512 -- so that a table ends within 12.5 % of the load it was sized for, and so that b2 stays a multiple of
// every power-of-two world size up to 8 (64 from b2 = 512): the hash-range shards of the multi-GPU merge nest in such a table
// (merge_regions).  KMERHIP_POW2_TABLE=1: powers of two only, rounds 1-3's tables (A/B, tests).
bool g_pow2_tables = false;  // (process-wide: set from the knobs of the last context created -- a sizing policy, not state)
bool pow2_tables() { return g_pow2_tables; }
u64 round_cap(double want) {
    u64 cap = MIN_CAP;
    while ((double)cap < want && cap < 1024ull * kh::REGION_SLOTS) cap *= 2;
    if ((double)cap >= want) return cap;
    if (pow2_tables()) {
        while ((double)cap < want) cap *= 2;
        return cap;
    }
    const double per_b2 = 1024.0 * kh::REGION_SLOTS;
    u64 b2 = (u64)(want / per_b2);
    if ((double)b2 * per_b2 < want) ++b2;
    u64 step = 1;
    if (b2 > 64) {
        u64 top = 64;
        while (top * 2 < b2) top *= 2;  // top < b2 <= 2 top
        step = std::min<u64>(top / 8, 64);
    } else {  // 1 .. 64: powers of two (tables of <= 2^28 slots: the granularity matters little there)
        u64 q = 1;
        while (q < b2) q *= 2;
        b2 = q;
    }
    b2 = (b2 + step - 1) / step * step;
    return b2 * 1024ull * kh::REGION_SLOTS;
}
kh::RegionGeom geom_of_cap(u64 cap) { return kh::kh_geom_of_regions(cap / kh::REGION_SLOTS); }
bool cap_is_pow2(u64 cap) { return (cap & (cap - 1)) == 0; }

kh::TableGeom table_geom(const kh_ctx *c, Slot *table, u64 cap);

kh::TableGeom table_geom(const kh_ctx *c, Slot *table, u64 cap) {
    kh::TableGeom tg;
    tg.table = table;
    const kh::RegionGeom rg = geom_of_cap(cap);
    tg.p1_bits = rg.p1_bits;
    tg.b2 = rg.b2;
    tg.k = c->k;
    tg.shard_shift = c->shard_shift;
    tg.shard_index = c->shard_index;
    return tg;
}

int grow_to(kh_ctx *c, u64 newcap) {
    Slot *nt = nullptr;
    int rc = ensure_wide(c);  // (the rehash reads the 16-byte table)
    if (rc != KH_OK) return rc;
    if (c->ntab) {  // the 8-byte image is for another capacity from here on: its memory is room for the new table
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->ntab);
        c->ntab = nullptr;
        c->ntab_cap = 0;
    }
    if (!c->table) {  // an empty table that was never needed: it just has another size now
        resize_empty_table(c, newcap);
        c->grows++;
        return KH_OK;
    }
    rc = alloc_table(c, newcap, &nt);
    if (rc != KH_OK) return rc;
    if (c->table_dirty) {  // logically empty: nothing to carry over, and the new table is clean
        c->table_dirty = false;
    } else {
        hipLaunchKernelGGL(kh::table_rehash_kernel, dim3(grid_for(c->cap)), dim3(kh::BLOCK), 0, c->stream, c->table,
                           c->cap, table_geom(c, nt, newcap), c->d_ctr);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    drop_table(c);
    if (c->trace) fprintf(stderr, "[kmerhip] table grown %llu -> %llu slots\n", c->cap, newcap);
    c->rheads_valid = false;
    c->table = nt;
    c->cap = newcap;
    c->grows++;
    return KH_OK;
}

// Guarantees that a launch claiming at most `bound` new slots cannot push the table past
// LOAD_HARD.  May synchronise and grow.  `bound` may be reduced by the caller and retried.
int ensure_room(kh_ctx *c, u64 bound, bool allow_shrink_hint, bool *want_smaller) {
    if (want_smaller) *want_smaller = false;
    const double hard = LOAD_HARD * (double)c->cap;
    if ((double)(c->distinct_known + c->pending_bound + bound) <= hard) return KH_OK;
    if (c->pending_bound) {
        int rc = sync_counters(c);
        if (rc != KH_OK) return rc;
        if ((double)(c->distinct_known + bound) <= hard) return KH_OK;
    }
    if (allow_shrink_hint && want_smaller && (double)c->distinct_known <= LOAD_TARGET * (double)c->cap) {
        *want_smaller = true;  // plenty of real room: a smaller launch avoids a needless doubling
        return KH_OK;
    }
    u64 newcap = c->cap;
    while ((double)(c->distinct_known + bound) > LOAD_HARD * (double)newcap ||
           (double)c->distinct_known > LOAD_TARGET * (double)newcap)
        newcap *= 2;
    return grow_to(c, newcap);
}

// The per-region scratch arrays of the partitioned path and of the shard merge.  ONE capacity for all of them: round 4 let the
// merge raise region_cap after growing only the three arrays it uses itself, and a later partitioned batch into the same context
// -- count little, merge into a larger geometry, kh_reset, count much -- then skipped the allocation of the others.
int ensure_region_scratch(kh_ctx *c, u64 nregions) {
    if (c->region_cap >= nregions && c->bstart) return KH_OK;
    int rc;
    const u64 have = c->region_cap;
    u64 z = c->bstart ? have + 1 : 0;
    if ((rc = ensure_buf(c, &c->bstart, &z, nregions + 1, "hipMalloc(bstart)")) != KH_OK) return rc;
#define KH_REGION_ARRAY(field)                                                                    \
    z = c->field ? have : 0;                                                                      \
    if ((rc = ensure_buf(c, &c->field, &z, nregions, "hipMalloc(" #field ")")) != KH_OK) return rc
    KH_REGION_ARRAY(rfail);
    KH_REGION_ARRAY(rnew);
    KH_REGION_ARRAY(rheads);
    KH_REGION_ARRAY(rreal);
    KH_REGION_ARRAY(bend);
    KH_REGION_ARRAY(hot_list);
    KH_REGION_ARRAY(radd);
    KH_REGION_ARRAY(rtouch);
#undef KH_REGION_ARRAY
    HIP_TRY(c, hipMemsetAsync(c->rtouch, 0, nregions, c->stream));
    c->region_cap = nregions;
    c->rheads_valid = false;  // (the array that held them is gone)
    return KH_OK;
}

int drain_events(kh_ctx *c) {
    for (auto &e : c->stage_events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            if (e.stage == ST_TEXT) {
                c->text_ms += ms;
            } else {
                c->stage_ms[e.stage] += ms;
                c->kernel_ms += ms;
            }
        }
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    c->stage_events.clear();
    for (auto &p : c->h2d_events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) c->h2d_ms += ms;
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    c->h2d_events.clear();
    return KH_OK;
}

int release_part_buffers(kh_ctx *c) {
    if (!c->keysA && !c->keysB) return KH_OK;
    int brc = end_borrow(c);
    if (brc != KH_OK) return brc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->keysA) (void)hipFree(c->keysA);
    if (c->keysB) (void)hipFree(c->keysB);
    c->keysA = c->keysB = nullptr;
    c->key_cap = c->keyb_cap = 0;
    c->prev_part_budget = c->part_budget;  // (what the context had: it gets it back if the memory is still there, batch.hip ensure_part_budget)
    c->part_budget = 0;  // (decided again at the next partitioned range)
    return KH_OK;
}

GeomChoice make_geom(const kh_ctx *c, u64 cap) {
    GeomChoice gc;
    const int hbits = 2 * (int)c->k - (int)c->shard_shift;  // significant bits of the placement hash
    uint32_t p1, b2;
    if (cap_is_pow2(cap)) {
        uint32_t rbits = 0;
        while (((u64)kh::REGION_SLOTS << (rbits + 1)) <= cap) ++rbits;  // 64-bit: a 2^32-slot table is legal
        uint32_t p2 = std::min<uint32_t>(rbits, kh::MAX_P2_BITS);
        p1 = rbits - p2;
        const int need = hbits - 32;  // level-1 bits needed for 32-bit payloads
        if (need > (int)p1 && need <= (int)kh::MAX_P1_BITS && need <= (int)rbits) {
            p1 = (uint32_t)need;
            p2 = rbits - p1;
        }
        // With more than 2^10 regions level 1 gets its full 1024 partitions, whatever the payload: the bins kernels are built
        // around one lane per partition, level 2 then has <= 512 buckets per partition up to 2^19 regions (what the arena
        // kernel and the unit-writing scatter take), and the written-out windows (window.hip.h) fix the digit at 10 bits.
        // (Round 2 did this for 64-bit payloads only: k = 19 at the headline size got 512 partitions x 1024 buckets, i.e. the
        // exact level 2 with the unaligned scatter -- 122 ms per step against k = 21's 75.)
        if (rbits > kh::MAX_P1_BITS && c->shard_shift == 0 && c->k >= kh::MAX_P1_BITS) {
            p1 = kh::MAX_P1_BITS;
            p2 = rbits - p1;
        }
        // (an 8-byte payload is the hash below the level-1 digit, shifted up by it -- part_common.hip.h Pay<u64> -- and all ones marks
        //  the padding of its segments: with at least one level-1 bit a payload's low bit is zero and cannot be taken for it)
        if (p1 == 0 && rbits >= 1 && (hbits > 32 || c->pay_mode == 64)) {
            p1 = 1;
            p2 = rbits - 1;
        }
        gc.ok = p1 <= kh::MAX_P1_BITS && p2 <= kh::MAX_P2_BITS;
        b2 = p2 <= 20 ? 1u << p2 : 0u;
    } else {  // 1024 x b2 regions, b2 not a power of two: the partition passes use the table's own geometry
        const kh::RegionGeom rg = geom_of_cap(cap);
        p1 = rg.p1_bits;
        b2 = rg.b2;
        gc.ok = p1 == kh::MAX_P1_BITS && b2 <= kh::MAX_B2 && c->shard_shift == 0;
    }
    gc.g.p1_bits = p1;
    gc.g.b2 = b2 ? b2 : 1u;
    gc.g.b2_magic = kh::part_magic_of(gc.g.b2);
    gc.g.p2_bits = kh::part_p2_bits_of(gc.g.b2);
    gc.g.k = c->k;
    gc.g.shard_shift = c->shard_shift;
    gc.g.shard_index = c->shard_index;
    gc.use32 = (hbits - (int)p1) <= 32 && c->pay_mode != 64;
    gc.g.defer = (gc.use32 && kh::p1_fast_ok(gc.g)) ? 1u : 0u;  // (level 1's FAST kernels leave the last Feistel round to level 2)
    return gc;
}

// exclusive scan of `in` (n u32 entries) into `out` (n+1 u64 entries) on the context's stream
int device_scan(kh_ctx *c, const uint32_t *in, u64 n, u64 *out) {
    const u64 nb = (n + kh::SCAN_CHUNK - 1) / kh::SCAN_CHUNK;
    int rc = ensure_buf(c, &c->scan_partial, &c->scan_cap, nb + 2, "hipMalloc(scan)");
    if (rc != KH_OK) return rc;
    hipLaunchKernelGGL(kh::scan_partials_kernel, dim3((unsigned)nb), dim3(kh::SCAN_NT), 0, c->stream, in, n, c->scan_partial);
    hipLaunchKernelGGL(kh::scan_spine_kernel, dim3(1), dim3(1024), 0, c->stream, c->scan_partial, nb);
    hipLaunchKernelGGL(kh::scan_apply_kernel, dim3((unsigned)nb), dim3(kh::SCAN_NT), 0, c->stream, in, n,
                       (const u64 *)c->scan_partial, out);
    HIP_TRY(c, hipGetLastError());
    return KH_OK;
}
int head_count_bits(const kh_ctx *c, u64 regions) {
    const int hb = kh::kh_below_bits(c->k, 0, kh::kh_geom_of_regions(regions));  // hash bits below the region index of an UNSHARDED table
    return (hb >= 1 && hb <= 28) ? 32 - hb : -1;  // at least 4 count bits
}

// region rebuild launch, by payload type
}  // namespace khi
using namespace khi;

// =============================================================================================
// lifecycle
// =============================================================================================
extern "C" int kh_abi_version(void) { return KMERHIP_ABI_VERSION; }

extern "C" int kh_create(kh_ctx **out, const kh_config *cfg) {
    if (!out || !cfg) return KH_ERR_BAD_ARG;
    *out = nullptr;
    if (cfg->struct_size != sizeof(kh_config)) return KH_ERR_BAD_ARG;
    if (cfg->k < 1 || cfg->k > 32) return KH_ERR_BAD_K;  // KmerLength::new, src/kmer.rs:100-110
    if (cfg->min_quality < -1 || cfg->min_quality > 255) return KH_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return KH_ERR_NO_DEVICE;
    }
    int dev = cfg->device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return KH_ERR_NO_DEVICE;
    if (dev >= ndev) return KH_ERR_BAD_ARG;

    kh_ctx *c = new (std::nothrow) kh_ctx();
    if (!c) return KH_ERR_OOM;
    c->device = dev;
    c->k = cfg->k;
    c->minq = cfg->min_quality;
    c->flags = cfg->flags;
    read_knobs(c->knobs);
    g_pow2_tables = c->knobs.pow2_table;
    if (!g_copy_threads) g_copy_threads = c->knobs.copy_threads;
    c->trace = (cfg->flags & KH_FLAG_TRACE) || c->knobs.trace;
    c->expect_bytes = (u64)cfg->input_mib << 20;
    c->hinted = cfg->capacity_hint != 0;
    c->hint_keys = cfg->capacity_hint;
    c->path_mode = (cfg->flags & KH_FLAG_FORCE_DIRECT) ? 1 : (cfg->flags & KH_FLAG_FORCE_PARTITION) ? 2 : 0;
    if (c->knobs.path) c->path_mode = c->knobs.path;
    c->pay_mode = c->knobs.payload;
    c->estimate_on = c->knobs.estimate;

    int rc = KH_OK;
    do {
        if (hipSetDevice(dev) != hipSuccess) { rc = KH_ERR_NO_DEVICE; break; }
        if (cfg->stream || (cfg->flags & KH_FLAG_CALLER_STREAM)) {
            c->stream = (hipStream_t)cfg->stream;  // NULL with KH_FLAG_CALLER_STREAM: the legacy default stream
        } else {
            if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rc = KH_ERR_HIP; break; }
            c->own_stream = true;
        }
        if (hipMalloc((void **)&c->d_ctr, sizeof(Counters)) != hipSuccess) { rc = KH_ERR_OOM; break; }
        if (hipHostMalloc((void **)&c->h_ctr, sizeof(Counters), hipHostMallocDefault) != hipSuccess) { rc = KH_ERR_OOM; break; }
        if (hipMemsetAsync(c->d_ctr, 0, sizeof(Counters), c->stream) != hipSuccess) { rc = KH_ERR_HIP; break; }
        u64 cap = cfg->capacity_hint ? round_cap((double)cfg->capacity_hint / HINT_LOAD) : DEFAULT_CAP;
        if (const u64 nr = c->knobs.table_regions)  // (test build: a table of exactly this many regions, e.g. 1024 x 40)
            if (kh::kh_regions_valid(nr) && nr * kh::REGION_SLOTS >= MIN_CAP) cap = nr * kh::REGION_SLOTS;
        c->cap = cap;  // (the table itself is allocated when something first needs it: need_table)
        if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = KH_ERR_HIP; break; }
    } while (0);
    if (rc != KH_OK) {
        kh_destroy(c);
        return rc;
    }
    *out = c;
    return KH_OK;
}

extern "C" void kh_destroy(kh_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    comm_release(c);
    drain_events(c);
    if (c->cstream) (void)hipStreamSynchronize(c->cstream);
    for (int i = 0; i < 2; ++i) {
        if (c->h_stage[i]) (void)hipHostFree(c->h_stage[i]);
        if (c->acc[i]) (void)hipFree(c->acc[i]);
        if (c->stage_done[i]) (void)hipEventDestroy(c->stage_done[i]);
        if (c->acc_free[i]) (void)hipEventDestroy(c->acc_free[i]);
    }
    if (c->cstream) (void)hipStreamDestroy(c->cstream);
    void *scratch[] = {c->keysA, c->keysB, c->blocks, c->moff, c->nch, c->info, c->H2, c->O2,
                       c->bstart, c->bend, c->hot_list, c->ptotal, c->pcap, c->heavy, c->ovf, c->ovf_list, c->rfail, c->rnew, c->rreal, c->rheads, c->radd, c->rdig, c->rtouch, c->scan_partial, c->est_set, c->merge_off, c->chunk_part, c->fill8, c->plist,
                       c->pcount, c->pstart, c->pool_next, c->txt_raw2[0], c->txt_raw2[1], c->txt_acc[0], c->txt_acc[1], c->txt_accq[0], c->txt_accq[1], c->txt_scan_partial, c->txt_ls, c->txt_st,
                       c->txt_tnl, c->txt_tbase, c->txt_tkeep, c->txt_tout, c->txt_err};
    for (void *q : scratch)
        if (q) (void)hipFree(q);
    drop_table(c);
    if (c->ntab) (void)hipFree(c->ntab);
    if (c->d_ctr) (void)hipFree(c->d_ctr);
    if (c->h_ctr) (void)hipHostFree(c->h_ctr);
    if (c->h_txt) (void)hipHostFree(c->h_txt);
    for (int i = 0; i < 2; ++i) {
        if (c->txt_acc_done[i]) (void)hipEventDestroy(c->txt_acc_done[i]);
        if (c->txt_copied[i]) (void)hipEventDestroy(c->txt_copied[i]);
        if (c->txt_scanned[i]) (void)hipEventDestroy(c->txt_scanned[i]);
    }
    if (c->sstream) {
        (void)hipStreamSynchronize(c->sstream);
        (void)hipStreamDestroy(c->sstream);
    }
    if (c->cstream2) {
        (void)hipStreamSynchronize(c->cstream2);
        (void)hipStreamDestroy(c->cstream2);
    }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int kh_reset(kh_ctx *c) {
    if (c && c->win_open) {  // the unwritten pieces are simply part of the lazy reset
        c->win_open = false;
        if (c->win_dirty) c->table_dirty = true;
    }
    int rc = enter(c, false, false, false, true);  // (touches no slot: the reset is lazy, and a table nobody has needed yet stays unallocated)
    if (rc != KH_OK) return rc;
    if (c->cstream) HIP_TRY(c, hipStreamSynchronize(c->cstream));
    c->acc_len = c->acc_carry = 0;  // pushes not yet counted are forgotten with everything else
    c->txt_acc_len = 0;
    if (c->sstream) HIP_TRY(c, hipStreamSynchronize(c->sstream));
    c->txt_unscanned.on = false;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    drain_events(c);
    // Lazy: no table_init here (5.5 ms for a 34 GB table).  A partitioned batch into an empty table
    // rewrites every region; any other use clears first (clear_if_dirty).
    if (!c->table_empty) c->table_dirty = true;
    if (c->table_borrowed) drop_table(c);  // (a merge's table inside the partition buffers: forgotten with its content)
    c->borrow_on = false;
    c->borrow_off[0] = c->borrow_off[1] = 0;
    c->narrow = false;         // (both images are stale now; the next fresh pass chooses again)
    c->narrow_banned = false;
    c->new_rate = -1.0;
    c->est_keys = 0;
    c->rheads_valid = false;
    HIP_TRY(c, hipMemsetAsync(c->d_ctr, 0, sizeof(Counters), c->stream));
    memset(c->h_ctr, 0, sizeof(Counters));  // (the host copy too: its k-mer total decides what may go into the 8-byte image)
    c->distinct_known = c->pending_bound = 0;
    c->bases_pushed = 0;
    c->launches = 0;
    c->kernel_ms = c->h2d_ms = c->text_ms = 0.0;
    for (double &m : c->stage_ms) m = 0.0;
    c->table_empty = true;
    c->part_batches = 0;
    c->shard_shift = c->shard_index = 0;  // KmerMap::new() again: an unsharded, empty table
    return KH_OK;
}

extern "C" int kh_finish(kh_ctx *c, kh_stats *st) {
    int rc = enter(c, true, true, false, true);
    if (rc != KH_OK) return rc;
    rc = sync_counters(c);
    if (rc != KH_OK) return rc;
    if (c->cstream) HIP_TRY(c, hipStreamSynchronize(c->cstream));
    drain_events(c);
    if (st) {
        st->bases = c->bases_pushed;
        st->kmers = c->h_ctr->kmers;
        st->distinct = c->h_ctr->distinct;
        st->table_slots = c->cap;
        st->grows = c->grows;
        st->launches = c->launches;
        st->count_kernel_ms = c->kernel_ms;
        st->h2d_ms = c->h2d_ms;
        st->part_batches = c->part_batches;
        for (int i = 0; i < KH_NUM_STAGES; ++i) st->stage_ms[i] = i < ST_N ? c->stage_ms[i] : 0.0;
        st->text_scan_ms = c->text_ms;
        st->slot_bytes = c->narrow ? 8 : 16;
    }
    if (c->trace)
        fprintf(stderr, "[kmerhip] bases=%llu kmers=%llu distinct=%llu slots=%llu load=%.3f launches=%llu kernel=%.3f ms h2d=%.3f ms | direct=%.2f p1c=%.2f p1s=%.2f p2c=%.2f p2s=%.2f region=%.2f misc=%.2f grow=%.2f\n",
                (u64)c->bases_pushed, c->h_ctr->kmers, c->h_ctr->distinct, c->cap,
                (double)c->h_ctr->distinct / (double)c->cap, (u64)c->launches, c->kernel_ms, c->h2d_ms,
                c->stage_ms[0], c->stage_ms[1], c->stage_ms[2], c->stage_ms[3], c->stage_ms[4], c->stage_ms[5],
                c->stage_ms[6], c->stage_ms[7]);
    return KH_OK;
}

// =============================================================================================
// output
// =============================================================================================

namespace khi {

int zero_cursors(kh_ctx *c) {
    HIP_TRY(c, hipMemsetAsync(&c->d_ctr->cursor, 0, 2 * sizeof(u64), c->stream));  // cursor + big
    return KH_OK;
}

int read_cursor(kh_ctx *c, u64 *cursor, u64 *big) {
    HIP_TRY(c, hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (cursor) *cursor = c->h_ctr->cursor;
    if (big) *big = c->h_ctr->big;
    return KH_OK;
}

}  // namespace khi

extern "C" int kh_result_size(kh_ctx *c, uint64_t min_count, uint64_t *n) {
    int rc = enter(c, true, true, false, true);
    if (rc != KH_OK) return rc;
    if (!n) return fail(c, KH_ERR_BAD_ARG, "n is NULL");
    rc = zero_cursors(c);
    if (rc != KH_OK) return rc;
    if (c->narrow)
        hipLaunchKernelGGL(kh::ntable_count_kernel, dim3(grid_for(c->cap)), dim3(kh::BLOCK), 0, c->stream, (const u64 *)c->ntab,
                           c->cap, (u64)min_count, c->d_ctr);
    else
    hipLaunchKernelGGL(kh::table_count_kernel, dim3(grid_for(c->cap)), dim3(kh::BLOCK), 0, c->stream, c->table,
                       c->cap, (u64)min_count, c->d_ctr);
    HIP_TRY(c, hipGetLastError());
    u64 cur = 0;
    rc = read_cursor(c, &cur, nullptr);
    if (rc != KH_OK) return rc;
    *n = cur;
    return KH_OK;
}

extern "C" int kh_result_copy_device(kh_ctx *c, uint64_t *d_keys, uint64_t *d_counts, uint64_t cap,
                                     uint64_t min_count, uint64_t *n) {
    int rc = enter(c, true, true, false, true);
    if (rc != KH_OK) return rc;
    if (!n || (cap && (!d_keys || !d_counts))) return fail(c, KH_ERR_BAD_ARG, "NULL output");
    rc = zero_cursors(c);
    if (rc != KH_OK) return rc;
    if (c->narrow)  // (keys come back through the inverse hash, slot by slot)
        hipLaunchKernelGGL(kh::ntable_compact_kernel, dim3(grid_for(c->cap)), dim3(kh::BLOCK), 0, c->stream, (const u64 *)c->ntab,
                           c->cap, c->narrow_g, (u64)min_count, (u64 *)d_keys, (u64 *)d_counts, (u64)cap, c->d_ctr);
    else
    hipLaunchKernelGGL(kh::table_compact_kernel, dim3(grid_for(c->cap)), dim3(kh::BLOCK), 0, c->stream, c->table,
                       c->cap, (u64)min_count, (u64 *)d_keys, (u64 *)d_counts, (u64)cap, c->d_ctr);
    HIP_TRY(c, hipGetLastError());
    u64 cur = 0;
    rc = read_cursor(c, &cur, nullptr);
    if (rc != KH_OK) return rc;
    *n = cur < cap ? cur : cap;
    if (cur > cap) return fail(c, KH_ERR_RANGE, "output arrays too small");
    return KH_OK;
}

extern "C" int kh_result_copy(kh_ctx *c, uint64_t *keys, uint64_t *counts, uint64_t cap, uint64_t min_count,
                              uint64_t *n) {
    int rc = enter(c, true, true, false, true);
    if (rc != KH_OK) return rc;
    if (!n || (cap && (!keys || !counts))) return fail(c, KH_ERR_BAD_ARG, "NULL output");
    uint64_t need = 0;
    rc = kh_result_size(c, min_count, &need);
    if (rc != KH_OK) return rc;
    if (need > cap) {
        *n = 0;
        return fail(c, KH_ERR_RANGE, "output arrays too small");
    }
    *n = 0;
    if (need == 0) return KH_OK;
    // The compacted pairs need two device arrays on their way out.  The partition buffers are idle here (every batch is
    // counted: enter() flushed) and, after any sizeable count, far larger than the result: use them instead of two fresh
    // multi-gigabyte allocations (mapping and unmapping 17 GB cost more than the compaction itself).
    // (Round 5: unless a merge has BORROWED parts of them -- its shard table may live there: then only what is not lent out,
    //  through the same bump allocator, handed back below.)
    uint64_t *dk = nullptr, *dc = nullptr;
    const u64 loan0 = c->borrow_off[0], loan1 = c->borrow_off[1];
    bool scratch = !c->borrow_on && c->keysA && c->keysB && c->key_cap >= need * sizeof(u64) && c->keyb_cap >= need * sizeof(u64);
    if (scratch) {
        dk = reinterpret_cast<uint64_t *>(c->keysA);
        dc = reinterpret_cast<uint64_t *>(c->keysB);
    } else if (c->borrow_on && (dk = (uint64_t *)borrow(c, need * sizeof(u64))) != nullptr && (dc = (uint64_t *)borrow(c, need * sizeof(u64))) != nullptr) {
        scratch = true;
    } else if ((c->borrow_off[0] = loan0, c->borrow_off[1] = loan1, dk = dc = nullptr, false) ||
               hipMalloc((void **)&dk, need * sizeof(u64)) != hipSuccess || hipMalloc((void **)&dc, need * sizeof(u64)) != hipSuccess) {
        (void)hipGetLastError();
        if (dk) (void)hipFree(dk);
        return fail(c, KH_ERR_OOM, "hipMalloc(result)");
    }
    uint64_t got = 0;
    rc = kh_result_copy_device(c, dk, dc, need, min_count, &got);
    if (rc == KH_OK) rc = d2h_staged(c, keys, dk, got * sizeof(u64));
    if (rc == KH_OK) rc = d2h_staged(c, counts, dc, got * sizeof(u64));
    if (rc == KH_OK) *n = got;
    if (!scratch) {
        (void)hipFree(dk);
        (void)hipFree(dc);
    }
    c->borrow_off[0] = loan0;  // (a loan taken for this call alone goes back)
    c->borrow_off[1] = loan1;
    return rc;
}

extern "C" int kh_histogram(kh_ctx *c, uint64_t min_count, uint64_t *count, uint64_t *freq, uint64_t cap,
                            uint64_t *n) {
    int rc = enter(c, true, true, false, true);
    if (rc != KH_OK) return rc;
    if (!n || (cap && (!count || !freq))) return fail(c, KH_ERR_BAD_ARG, "NULL output");
    u64 *d_dense = nullptr, *d_big = nullptr;
    u64 big_cap = 1ull << 16;
    std::vector<u64> dense(kh::HIST_DENSE), big;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (hipMalloc((void **)&d_dense, kh::HIST_DENSE * sizeof(u64)) != hipSuccess ||
            hipMalloc((void **)&d_big, big_cap * sizeof(u64)) != hipSuccess) {
            (void)hipGetLastError();
            if (d_dense) (void)hipFree(d_dense);
            return fail(c, KH_ERR_OOM, "hipMalloc(histogram)");
        }
        rc = zero_cursors(c);
        if (rc == KH_OK && hipMemsetAsync(d_dense, 0, kh::HIST_DENSE * sizeof(u64), c->stream) != hipSuccess)
            rc = fail(c, KH_ERR_HIP, "hipMemsetAsync(histogram)");
        u64 nbig = 0;
        if (rc == KH_OK) {
            if (c->narrow)
                hipLaunchKernelGGL(kh::ntable_hist_kernel, dim3(grid_for(c->cap)), dim3(kh::BLOCK), 0, c->stream, (const u64 *)c->ntab,
                                   c->cap, (u64)min_count, d_dense, d_big, big_cap, c->d_ctr);
            else
            hipLaunchKernelGGL(kh::table_hist_kernel, dim3(grid_for(c->cap)), dim3(kh::BLOCK), 0, c->stream, c->table,
                               c->cap, (u64)min_count, d_dense, d_big, big_cap, c->d_ctr);
            if (hipGetLastError() != hipSuccess) rc = fail(c, KH_ERR_HIP, "table_hist_kernel");
        }
        if (rc == KH_OK) rc = read_cursor(c, nullptr, &nbig);
        if (rc == KH_OK && nbig <= big_cap) {
            big.resize(nbig);
            if (hipMemcpy(dense.data(), d_dense, kh::HIST_DENSE * sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess ||
                (nbig && hipMemcpy(big.data(), d_big, nbig * sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess))
                rc = fail(c, KH_ERR_HIP, "hipMemcpy(histogram)");
        }
        (void)hipFree(d_dense);
        (void)hipFree(d_big);
        d_dense = d_big = nullptr;
        if (rc != KH_OK) return rc;
        if (nbig <= big_cap) break;
        big_cap = nbig;  // second pass with an exactly sized list
    }
    // BTreeMap<u64,u64> order: ascending by count (src/histogram.rs:33,88-94)
    std::map<u64, u64> tail;
    for (u64 v : big) tail[v]++;
    u64 out = 0;
    for (u64 i = 0; i < kh::HIST_DENSE; ++i)
        if (dense[i]) {
            if (out < cap) { count[out] = i; freq[out] = dense[i]; }
            ++out;
        }
    for (auto &kv : tail) {
        if (out < cap) { count[out] = kv.first; freq[out] = kv.second; }
        ++out;
    }
    *n = out < cap ? out : cap;
    if (out > cap) return fail(c, KH_ERR_RANGE, "histogram arrays too small");
    return KH_OK;
}

extern "C" int kh_lookup(kh_ctx *c, const uint64_t *keys, uint64_t n, uint64_t *counts) {
    int rc = enter(c, true, true, false, true);
    if (rc != KH_OK) return rc;
    if (n == 0) return KH_OK;
    if (!keys || !counts) return fail(c, KH_ERR_BAD_ARG, "NULL argument");
    u64 *dk = nullptr, *dc = nullptr;
    if (hipMalloc((void **)&dk, n * sizeof(u64)) != hipSuccess || hipMalloc((void **)&dc, n * sizeof(u64)) != hipSuccess) {
        (void)hipGetLastError();
        if (dk) (void)hipFree(dk);
        return fail(c, KH_ERR_OOM, "hipMalloc(lookup)");
    }
    hipError_t e = hipMemcpyAsync(dk, keys, n * sizeof(u64), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        if (c->narrow)
            hipLaunchKernelGGL(kh::ntable_lookup_kernel, dim3(grid_for(n)), dim3(kh::BLOCK), 0, c->stream, (const u64 *)c->ntab,
                               c->narrow_g, dk, (u64)n, dc);
        else
        hipLaunchKernelGGL(kh::table_lookup_kernel, dim3(grid_for(n)), dim3(kh::BLOCK), 0, c->stream,
                           table_geom(c, c->table, c->cap), dk, (u64)n, dc);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(counts, dc, n * sizeof(u64), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dk);
    (void)hipFree(dc);
    if (e != hipSuccess) return fail(c, KH_ERR_HIP, "kh_lookup", e);
    return KH_OK;
}

// =============================================================================================
// host memory the device can reach directly
// =============================================================================================
extern "C" int kh_host_alloc(void **out, uint64_t bytes) {
    if (!out) return KH_ERR_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return KH_ERR_NO_DEVICE;
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        return KH_ERR_OOM;
    }
    *out = p;
    return KH_OK;
}

extern "C" int kh_host_free(void *p) {
    if (!p) return KH_OK;
    if (hipHostFree(p) != hipSuccess) {
        (void)hipGetLastError();
        return KH_ERR_BAD_ARG;
    }
    return KH_OK;
}

extern "C" int kh_host_register(void *p, uint64_t bytes) {
    if (!p || !bytes) return KH_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return KH_ERR_NO_DEVICE;
    }
    if (hipHostRegister(p, bytes, hipHostRegisterPortable) != hipSuccess) {
        (void)hipGetLastError();
        return KH_ERR_OOM;
    }
    return KH_OK;
}

extern "C" int kh_host_unregister(void *p) {
    if (!p) return KH_ERR_BAD_ARG;
    if (hipHostUnregister(p) != hipSuccess) {
        (void)hipGetLastError();
        return KH_ERR_BAD_ARG;
    }
    return KH_OK;
}

// =============================================================================================
// pure helpers
// =============================================================================================
extern "C" int kh_pack(const uint8_t *bases, uint32_t k, uint64_t *packed, uint32_t *err_pos) {
    if (!bases || !packed) return KH_ERR_BAD_ARG;
    if (k < 1 || k > 32) return KH_ERR_BAD_K;
    uint64_t acc = 0;
    for (uint32_t i = 0; i < k; ++i) {
        if (!kh_base_valid(bases[i])) {  // InvalidBaseError{position}, src/kmer.rs:277-280
            if (err_pos) *err_pos = i;
            return KH_ERR_BAD_ARG;
        }
        acc = (acc << 2) | kh_base_code(bases[i]);
    }
    *packed = acc;
    return KH_OK;
}

extern "C" int kh_unpack(uint64_t packed, uint32_t k, uint8_t *out) {
    if (!out) return KH_ERR_BAD_ARG;
    if (k < 1 || k > 32) return KH_ERR_BAD_K;
    for (uint32_t i = 0; i < k; ++i) out[i] = (uint8_t)"ACGT"[(packed >> (2 * (k - 1 - i))) & 3u];
    return KH_OK;
}

extern "C" int kh_canonical(uint64_t packed, uint32_t k, uint64_t *canonical, int *is_rc) {
    if (!canonical) return KH_ERR_BAD_ARG;
    if (k < 1 || k > 32) return KH_ERR_BAD_K;
    packed &= kh_kmask(k);
    const uint64_t rc = kh_revcomp(packed, k);
    *canonical = packed < rc ? packed : rc;
    if (is_rc) *is_rc = rc < packed;  // strictly smaller only: palindrome keeps the original (kmer.rs:365)
    return KH_OK;
}

// =============================================================================================
// diagnostics
// =============================================================================================
extern "C" const char *kh_strerror(int s) {
    switch (s) {
    case KH_OK: return "ok";
    case KH_ERR_BAD_K: return "k-mer length is out of range: must be between 1 and 32";  // src/error.rs:87
    case KH_ERR_BAD_ARG: return "invalid argument";
    case KH_ERR_NO_DEVICE: return "no usable HIP device";
    case KH_ERR_OOM: return "out of memory";
    case KH_ERR_TABLE_FULL: return "hash table full";
    case KH_ERR_HIP: return "HIP runtime error";
    case KH_ERR_STATE: return "invalid context state";
    case KH_ERR_RANGE: return "output array too small";
    case KH_ERR_FORMAT: return "text layout not accepted by the device record scanner";
    case KH_ERR_RCCL: return "RCCL error";
    case KH_ERR_PEER: return "another rank of the collective failed";
    default: return "unknown error";
    }
}

extern "C" const char *kh_last_error(const kh_ctx *c) { return c ? c->last_error.c_str() : ""; }

// =============================================================================================
// synthetic reads
// =============================================================================================
extern "C" int kh_synth_reads_device(int device, void *stream, uint64_t seed, uint64_t genome_len, uint32_t read_len,
                                     uint64_t first_read, uint64_t n_reads, uint8_t *d_bases, uint8_t *d_qual) {
    if (!d_bases || read_len == 0 || genome_len < read_len) return KH_ERR_BAD_ARG;
    if ((((uintptr_t)d_bases) & 15) || (d_qual && (((uintptr_t)d_qual) & 15))) return KH_ERR_BAD_ARG;
    if (n_reads == 0) return KH_OK;
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return KH_ERR_NO_DEVICE;
    const u64 nchunks = (n_reads * ((u64)read_len + 1) + 15) / 16;
    hipLaunchKernelGGL(kh::synth_reads_kernel, dim3(grid_for(nchunks)), dim3(kh::BLOCK), 0, (hipStream_t)stream,
                       (u64)seed, (u64)genome_len, read_len, (u64)first_read, (u64)n_reads, d_bases, d_qual);
    return hipGetLastError() == hipSuccess ? KH_OK : KH_ERR_HIP;
}

