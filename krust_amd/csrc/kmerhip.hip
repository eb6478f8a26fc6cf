// kmerhip.hip -- C-ABI implementation (include/kmerhip.h) over the gfx950 kernels.
//
// Host-side orchestration of the device path that replaces KmerMap::build /
// build_with_quality / into_hashmap (reference src/run.rs:494-582).  No CPU fallback exists:
// every counting entry point needs a HIP device and fails with KH_ERR_NO_DEVICE otherwise.
#include "../../include/kmerhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "kernels.hip.h"

using kh::Counters;
using kh::Slot;
using kh::u64;

namespace {

constexpr double LOAD_HARD = 0.80;    // never let distinct exceed this fraction of capacity
constexpr double LOAD_TARGET = 0.50;  // load right after a growth
constexpr u64 MIN_CAP = 1ull << 16;
constexpr u64 DEFAULT_CAP = 1ull << 20;
constexpr u64 SUB_TILES = 1ull << 16;      // tiles per count launch (2^28 positions)
constexpr u64 SUB_TILES_MIN = 1ull << 10;  // smallest launch when squeezing under LOAD_HARD
constexpr u64 STAGE_BYTES = 64ull << 20;   // host staging chunk for kh_push
constexpr u64 HALO = 32;                   // >= k-1 bytes re-sent in front of every staged chunk
constexpr int GRID_CAP = 256 * 8;          // 256 CUs x 8 resident workgroups of 256 threads

}  // namespace

struct kh_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t k = 0;
    int32_t minq = -1;
    uint32_t flags = 0;
    bool trace = false;

    Slot *table = nullptr;
    u64 cap = 0;
    Counters *d_ctr = nullptr;
    Counters *h_ctr = nullptr;  // pinned

    u64 distinct_known = 0;  // exact as of the last counter read-back
    u64 pending_bound = 0;   // upper bound on claims by launches since then
    u64 bases_pushed = 0;
    u64 grows = 0;
    u64 launches = 0;
    double kernel_ms = 0.0;
    double h2d_ms = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;

    uint8_t *h_stage[2] = {nullptr, nullptr};  // pinned: bases then qual, each HALO+STAGE_BYTES
    uint8_t *d_stage[2] = {nullptr, nullptr};
    hipEvent_t stage_done[2] = {nullptr, nullptr};
    bool stage_used[2] = {false, false};

    bool poisoned = false;
    std::string last_error;
};

namespace {

int fail(kh_ctx *c, int code, const char *what, hipError_t e = hipSuccess) {
    if (c) {
        c->last_error = what;
        if (e != hipSuccess) {
            c->last_error += ": ";
            c->last_error += hipGetErrorString(e);
        }
        if (code == KH_ERR_HIP || code == KH_ERR_TABLE_FULL || code == KH_ERR_OOM) c->poisoned = true;
    }
    return code;
}

#define HIP_TRY(c, call)                                              \
    do {                                                              \
        hipError_t e_ = (call);                                       \
        if (e_ != hipSuccess) return fail((c), KH_ERR_HIP, #call, e_); \
    } while (0)

int grid_for(u64 items) {
    u64 b = (items + kh::BLOCK - 1) / kh::BLOCK;
    if (b < 1) b = 1;
    if (b > (u64)GRID_CAP) b = GRID_CAP;
    return (int)b;
}

int enter(kh_ctx *c) {
    if (!c) return KH_ERR_BAD_ARG;
    if (c->poisoned) return fail(c, KH_ERR_STATE, "context is poisoned by an earlier error");
    HIP_TRY(c, hipSetDevice(c->device));
    return KH_OK;
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

// Blocks until the stream is idle and refreshes the exact counters.
int sync_counters(kh_ctx *c) {
    HIP_TRY(c, hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->distinct_known = c->h_ctr->distinct;
    c->pending_bound = 0;
    if (c->h_ctr->failed) return fail(c, KH_ERR_TABLE_FULL, "an upsert found no free slot");
    return KH_OK;
}

u64 round_cap(double want) {
    u64 cap = (u64)want;
    if (cap < MIN_CAP) cap = MIN_CAP;
    return (cap + 4095) & ~4095ull;
}

int grow_to(kh_ctx *c, u64 newcap) {
    Slot *nt = nullptr;
    int rc = alloc_table(c, newcap, &nt);
    if (rc != KH_OK) return rc;
    hipLaunchKernelGGL(kh::table_rehash_kernel, dim3(grid_for(c->cap)), dim3(kh::BLOCK), 0, c->stream, c->table,
                       c->cap, nt, newcap, c->d_ctr);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipFree(c->table));
    if (c->trace) fprintf(stderr, "[kmerhip] table grown %llu -> %llu slots\n", c->cap, newcap);
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
                       qbase, qaligned, vbeg, vend, wlo, tile0, ntiles, tpb, c->k, thr, c->table, c->cap, c->d_ctr);
}

// Count all windows of the device buffer [d_bases, d_bases+n) that end at offset >= wlo_off.
int count_device_range(kh_ctx *c, const uint8_t *d_bases, const uint8_t *d_qual, u64 n, u64 wlo_off) {
    if (n == 0) return KH_OK;
    const uintptr_t addr = (uintptr_t)d_bases;
    const u64 lead = addr & 15;
    const uint8_t *abase = d_bases - lead;
    const u64 vbeg = lead, vend = lead + n, wlo = lead + wlo_off;
    const bool use_qual = (d_qual != nullptr) && (c->minq >= 0);
    const uint8_t *qbase = nullptr;
    int qaligned = 0;
    if (use_qual) {
        qbase = d_qual - lead;  // same virtual coordinates as the bases
        qaligned = (((uintptr_t)qbase) & 15) == 0;
    }
    const u64 first_tile = wlo / kh::TILE;
    const u64 end_tile = (vend + kh::TILE - 1) / kh::TILE;
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
        hipEvent_t e0, e1;
        HIP_TRY(c, hipEventCreate(&e0));
        HIP_TRY(c, hipEventCreate(&e1));
        HIP_TRY(c, hipEventRecord(e0, c->stream));
        if (use_qual) launch_count<true>(c, abase, qbase, qaligned, vbeg, vend, wlo, t, nt);
        else launch_count<false>(c, abase, nullptr, 0, vbeg, vend, wlo, t, nt);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(e1, c->stream));
        c->events.emplace_back(e0, e1);
        c->launches++;
        c->pending_bound += nt * kh::TILE;
        t += nt;
    }
    return KH_OK;
}

int drain_events(kh_ctx *c) {
    for (auto &p : c->events) {
        float ms = 0.f;
        hipError_t e = hipEventElapsedTime(&ms, p.first, p.second);
        if (e == hipSuccess) c->kernel_ms += ms;
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    c->events.clear();
    return KH_OK;
}

int ensure_stage(kh_ctx *c, bool with_qual) {
    const u64 per = HALO + STAGE_BYTES;
    const u64 bytes = per * 2;  // bases + qual halves
    (void)with_qual;
    for (int i = 0; i < 2; ++i) {
        if (!c->h_stage[i]) {
            hipError_t e = hipHostMalloc((void **)&c->h_stage[i], bytes, hipHostMallocDefault);
            if (e != hipSuccess) return fail(c, KH_ERR_OOM, "hipHostMalloc(stage)", e);
            e = hipMalloc((void **)&c->d_stage[i], bytes);
            if (e != hipSuccess) return fail(c, KH_ERR_OOM, "hipMalloc(stage)", e);
            HIP_TRY(c, hipEventCreateWithFlags(&c->stage_done[i], hipEventDisableTiming));
        }
    }
    return KH_OK;
}

}  // namespace

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
    const char *tr = getenv("KMERHIP_TRACE");
    c->trace = (cfg->flags & KH_FLAG_TRACE) || (tr && tr[0] && tr[0] != '0');

    int rc = KH_OK;
    do {
        if (hipSetDevice(dev) != hipSuccess) { rc = KH_ERR_NO_DEVICE; break; }
        if (cfg->stream) {
            c->stream = (hipStream_t)cfg->stream;
        } else {
            if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rc = KH_ERR_HIP; break; }
            c->own_stream = true;
        }
        if (hipMalloc((void **)&c->d_ctr, sizeof(Counters)) != hipSuccess) { rc = KH_ERR_OOM; break; }
        if (hipHostMalloc((void **)&c->h_ctr, sizeof(Counters), hipHostMallocDefault) != hipSuccess) { rc = KH_ERR_OOM; break; }
        if (hipMemsetAsync(c->d_ctr, 0, sizeof(Counters), c->stream) != hipSuccess) { rc = KH_ERR_HIP; break; }
        u64 cap = cfg->capacity_hint ? round_cap((double)cfg->capacity_hint / 0.6) : DEFAULT_CAP;
        rc = alloc_table(c, cap, &c->table);
        if (rc != KH_OK) break;
        c->cap = cap;
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
    drain_events(c);
    for (int i = 0; i < 2; ++i) {
        if (c->h_stage[i]) (void)hipHostFree(c->h_stage[i]);
        if (c->d_stage[i]) (void)hipFree(c->d_stage[i]);
        if (c->stage_done[i]) (void)hipEventDestroy(c->stage_done[i]);
    }
    if (c->table) (void)hipFree(c->table);
    if (c->d_ctr) (void)hipFree(c->d_ctr);
    if (c->h_ctr) (void)hipHostFree(c->h_ctr);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int kh_reset(kh_ctx *c) {
    int rc = enter(c);
    if (rc != KH_OK) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    drain_events(c);
    hipLaunchKernelGGL(kh::table_init_kernel, dim3(grid_for(c->cap)), dim3(kh::BLOCK), 0, c->stream, c->table, c->cap);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemsetAsync(c->d_ctr, 0, sizeof(Counters), c->stream));
    c->distinct_known = c->pending_bound = 0;
    c->bases_pushed = 0;
    c->launches = 0;
    c->kernel_ms = c->h2d_ms = 0.0;
    return KH_OK;
}

// =============================================================================================
// input
// =============================================================================================
extern "C" int kh_push_device(kh_ctx *c, const uint8_t *d_bases, const uint8_t *d_qual, uint64_t n) {
    int rc = enter(c);
    if (rc != KH_OK) return rc;
    if (n && !d_bases) return fail(c, KH_ERR_BAD_ARG, "d_bases is NULL");
    rc = count_device_range(c, d_bases, d_qual, n, 0);
    if (rc == KH_OK) c->bases_pushed += n;
    return rc;
}

extern "C" int kh_push(kh_ctx *c, const uint8_t *bases, const uint8_t *qual, uint64_t n) {
    int rc = enter(c);
    if (rc != KH_OK) return rc;
    if (n && !bases) return fail(c, KH_ERR_BAD_ARG, "bases is NULL");
    if (n == 0) return KH_OK;
    const bool with_qual = (qual != nullptr) && (c->minq >= 0);
    rc = ensure_stage(c, with_qual);
    if (rc != KH_OK) return rc;
    const u64 per = HALO + STAGE_BYTES;
    int p = 0;
    for (u64 off = 0; off < n; off += STAGE_BYTES, p ^= 1) {
        const u64 len = std::min(STAGE_BYTES, n - off);
        const u64 halo = off ? HALO : 0;  // re-send the k-1 look-back in front of every later chunk
        if (c->stage_used[p]) HIP_TRY(c, hipEventSynchronize(c->stage_done[p]));
        hipEvent_t t0, t1;
        HIP_TRY(c, hipEventCreate(&t0));
        HIP_TRY(c, hipEventCreate(&t1));
        memcpy(c->h_stage[p] + (HALO - halo), bases + off - halo, halo + len);
        if (with_qual) memcpy(c->h_stage[p] + per + (HALO - halo), qual + off - halo, halo + len);
        HIP_TRY(c, hipEventRecord(t0, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_stage[p] + (HALO - halo), c->h_stage[p] + (HALO - halo), halo + len,
                                  hipMemcpyHostToDevice, c->stream));
        if (with_qual)
            HIP_TRY(c, hipMemcpyAsync(c->d_stage[p] + per + (HALO - halo), c->h_stage[p] + per + (HALO - halo),
                                      halo + len, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipEventRecord(t1, c->stream));
        HIP_TRY(c, hipEventRecord(c->stage_done[p], c->stream));
        c->stage_used[p] = true;
        rc = count_device_range(c, c->d_stage[p] + (HALO - halo), with_qual ? c->d_stage[p] + per + (HALO - halo) : nullptr,
                                halo + len, halo);
        if (rc != KH_OK) return rc;
        // the device staging buffer is reused two chunks later on the same stream (ordered);
        // the pinned one is guarded by stage_done.  H2D time is accounted lazily.
        HIP_TRY(c, hipEventSynchronize(t1));
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t0, t1) == hipSuccess) c->h2d_ms += ms;
        (void)hipEventDestroy(t0);
        (void)hipEventDestroy(t1);
    }
    c->bases_pushed += n;
    return KH_OK;
}

extern "C" int kh_finish(kh_ctx *c, kh_stats *st) {
    int rc = enter(c);
    if (rc != KH_OK) return rc;
    rc = sync_counters(c);
    if (rc != KH_OK) return rc;
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
    }
    if (c->trace)
        fprintf(stderr, "[kmerhip] bases=%llu kmers=%llu distinct=%llu slots=%llu load=%.3f launches=%llu kernel=%.3f ms h2d=%.3f ms\n",
                (u64)c->bases_pushed, c->h_ctr->kmers, c->h_ctr->distinct, c->cap,
                (double)c->h_ctr->distinct / (double)c->cap, (u64)c->launches, c->kernel_ms, c->h2d_ms);
    return KH_OK;
}

// =============================================================================================
// output
// =============================================================================================
namespace {

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

}  // namespace

extern "C" int kh_result_size(kh_ctx *c, uint64_t min_count, uint64_t *n) {
    int rc = enter(c);
    if (rc != KH_OK) return rc;
    if (!n) return fail(c, KH_ERR_BAD_ARG, "n is NULL");
    rc = zero_cursors(c);
    if (rc != KH_OK) return rc;
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
    int rc = enter(c);
    if (rc != KH_OK) return rc;
    if (!n || (cap && (!d_keys || !d_counts))) return fail(c, KH_ERR_BAD_ARG, "NULL output");
    rc = zero_cursors(c);
    if (rc != KH_OK) return rc;
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
    int rc = enter(c);
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
    uint64_t *dk = nullptr, *dc = nullptr;
    if (hipMalloc((void **)&dk, need * sizeof(u64)) != hipSuccess || hipMalloc((void **)&dc, need * sizeof(u64)) != hipSuccess) {
        (void)hipGetLastError();
        if (dk) (void)hipFree(dk);
        return fail(c, KH_ERR_OOM, "hipMalloc(result)");
    }
    uint64_t got = 0;
    rc = kh_result_copy_device(c, dk, dc, need, min_count, &got);
    if (rc == KH_OK) {
        hipError_t e1 = hipMemcpy(keys, dk, got * sizeof(u64), hipMemcpyDeviceToHost);
        hipError_t e2 = hipMemcpy(counts, dc, got * sizeof(u64), hipMemcpyDeviceToHost);
        if (e1 != hipSuccess || e2 != hipSuccess) rc = fail(c, KH_ERR_HIP, "hipMemcpy(result)", e1 != hipSuccess ? e1 : e2);
        else *n = got;
    }
    (void)hipFree(dk);
    (void)hipFree(dc);
    return rc;
}

extern "C" int kh_histogram(kh_ctx *c, uint64_t min_count, uint64_t *count, uint64_t *freq, uint64_t cap,
                            uint64_t *n) {
    int rc = enter(c);
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
    int rc = enter(c);
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
        hipLaunchKernelGGL(kh::table_lookup_kernel, dim3(grid_for(n)), dim3(kh::BLOCK), 0, c->stream, c->table, c->cap,
                           dk, (u64)n, dc);
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
// multi-GPU merge support
// =============================================================================================
extern "C" uint32_t kh_owner(uint64_t key, uint32_t nparts) { return nparts ? kh_owner_of(key, nparts) : 0; }

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
                           c->cap, nparts, d_parts);
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
                               c->table, c->cap, nparts, d_parts, (u64 *)d_keys, (u64 *)d_counts, (u64)cap);
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
        hipLaunchKernelGGL(kh::table_merge_pairs_kernel, dim3(grid_for(m)), dim3(kh::BLOCK), 0, c->stream, c->table,
                           c->cap, (const u64 *)d_keys + off, (const u64 *)d_counts + off, m, c->d_ctr);
        HIP_TRY(c, hipGetLastError());
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
