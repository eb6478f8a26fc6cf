// kmerhip.hip -- C-ABI implementation (include/kmerhip.h) over the gfx950 kernels.
//
// Host-side orchestration of the device path that replaces KmerMap::build /
// build_with_quality / into_hashmap (reference src/run.rs:494-582).  No CPU fallback exists:
// every counting entry point needs a HIP device and fails with KH_ERR_NO_DEVICE otherwise.
#include "../../include/kmerhip.h"

#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "kernels.hip.h"
#include "partition.hip.h"
#include "level1_api.h"
#include "shard.hip.h"
#include "rawparse.hip.h"

using kh::Counters;
using kh::Slot;
using kh::u64;

namespace {

constexpr double LOAD_HARD = 0.80;    // never let distinct exceed this fraction of capacity
constexpr double LOAD_TARGET = 0.50;  // load right after a growth
constexpr double HINT_LOAD = 0.65;    // a capacity hint of n keys gets the smallest table that holds them at this load
constexpr u64 MIN_CAP = 8ull * kh::REGION_SLOTS;
constexpr u64 DEFAULT_CAP = 1ull << 20;
constexpr u64 SUB_TILES = 1ull << 16;      // tiles per count launch (2^28 positions)
constexpr u64 SUB_TILES_MIN = 1ull << 10;  // smallest launch when squeezing under LOAD_HARD
constexpr u64 STAGE_BYTES = 64ull << 20;   // host staging chunk for kh_push
constexpr u64 ACC_MAX = 8ull << 30;        // device accumulation buffer of kh_push (x2, x2 with qualities): an upper limit --
                                           // acc_limit() also keeps the buffers within a quarter of the free memory.  (Round 2: 2 GiB,
                                           // i.e. 8 partitioned batches per S100M, each non-fresh region pass re-reading and re-writing
                                           // the whole 34 GB table: 177 ms of kernels against 74 resident.  8 GiB: two batches.)
constexpr u64 ACC_MIN = 1ull << 20;
constexpr u64 HALO = 32;                   // >= k-1 bytes re-sent in front of every staged chunk
constexpr int GRID_CAP = 256 * 8;          // 256 CUs x 8 resident workgroups of 256 threads
#ifndef KH_ARENA_UNITB
#define KH_ARENA_UNITB 128  // bytes per flushed unit of the arena level 2, 4-byte payloads (64: A/B builds)
#endif
#ifndef KH_PART_G1
#define KH_PART_G1 512
#endif
constexpr int PART_G1 = KH_PART_G1;               // level-1 workgroups (fixed: count and scatter must agree)
constexpr u64 PART_MIN_WINDOWS = 1ull << 22;   // below this the partition passes cannot pay off
constexpr double LOAD_PART = 0.70;         // grow before the next partitioned batch above this load
enum { ST_DIRECT = 0, ST_P1_COUNT, ST_P1_SCATTER, ST_P2_COUNT, ST_P2_SCATTER, ST_REGION, ST_MISC, ST_GROW, ST_N, ST_TEXT = ST_N };

}  // namespace

namespace {
struct Comm;  // exchange.hip.h: RCCL communicator (or the process-local hub) of this rank

// ---- environment knobs (round 4: ONE place) -----------------------------------------------------------------------------
// Read ONCE, at kh_create, into the context.  Two kinds:
//   * tunables of the product library: how much memory, how many threads, how long to wait, what to print, which insert path.
//     None of them can change a count.
//   * switches of the TEST build (-DKH_TESTING=1: krust_amd/lib/libkmerhip_testing.so, what tests/ load): force a kernel
//     variant, a table geometry, a fallback, an injected failure.  They exist so that every path can be driven against the
//     oracle; the product library does not compile them in -- there is no environment variable that makes it take an
//     ablation path or fail a merge.  (The test build also re-reads them at every call: tests flip them between batches.)
#ifndef KH_TESTING
#define KH_TESTING 0
#endif
struct Knobs {
    // product
    bool trace = false;              // KMERHIP_TRACE=1
    int path = 0;                    // KMERHIP_PATH=direct|partition: 1 | 2 (0: chosen per push)
    double part_budget_gb = 0;       // KMERHIP_PART_BUDGET_GB
    u64 acc_max_mb = 0;              // KMERHIP_ACC_MAX_MB
    u64 text_acc_mb = 0;             // KMERHIP_TEXT_ACC_MB
    int copy_threads = 0;            // KMERHIP_COPY_THREADS
    bool estimate = true;            // KMERHIP_ESTIMATE=0: size tables from the hint / the worst case, never from the level-1 sample
    bool pow2_table = false;         // KMERHIP_POW2_TABLE=1: tables of 2^n regions only (rounds 1-3's)
    // test build only
    int payload = 0;                 // KMERHIP_PAYLOAD=64
    u64 table_regions = 0;           // KMERHIP_TABLE_REGIONS
    int region_nt = 0;               // KMERHIP_REGION_NT
    bool p2_force_wide = false;      // KMERHIP_P2_FORCE_WIDE=1
    bool generic_k = false;          // KMERHIP_GENERIC_K=1
    bool p1_legacy = false;          // KMERHIP_P1_BINS=0
    bool p2_lines = true;            // KMERHIP_P2_LINES=0
    bool l2_arena = true;            // KMERHIP_L2_ARENA=0
    u64 l2_ovf_cap = ~0ull;          // KMERHIP_L2_OVF_CAP
    int l2_skew_x = -1;              // KMERHIP_L2_SKEW_X (-1: default 2)
    u64 l2_heavy_room = ~0ull;       // KMERHIP_L2_HEAVY_ROOM
    bool narrow = true;              // KMERHIP_NARROW=0
    u64 hot_cut = 0;                 // KMERHIP_HOT_CUT (0: default; ~0: no bucket is hot)
    int ovf_agg = -1;                // KMERHIP_OVF_AGG
    double survival = 0;             // KMERHIP_SURVIVAL
    bool stop_after_p1 = false, stop_after_p2 = false;  // ablation builds (KH_ABL*)
};
inline const char *env_of(const char *name) {
    const char *e = getenv(name);
    return (e && *e) ? e : nullptr;
}
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
    if (const char *e = env_of("KMERHIP_HOT_CUT")) k.hot_cut = (e[0] == '0' && !e[1]) ? ~0ull : strtoull(e, nullptr, 10);
    if (const char *e = env_of("KMERHIP_OVF_AGG")) k.ovf_agg = atoi(e);
    if (const char *e = env_of("KMERHIP_SURVIVAL")) k.survival = atof(e);
    k.stop_after_p1 = env_of("KMERHIP_STOP_AFTER_P1") != nullptr;
    k.stop_after_p2 = env_of("KMERHIP_STOP_AFTER_P2") != nullptr;
#endif
}
}

struct kh_ctx {
    Knobs knobs;
    int device = 0;
    Comm *comm = nullptr;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t k = 0;
    int32_t minq = -1;
    uint32_t flags = 0;
    bool trace = false;

    Slot *table = nullptr;
    u64 cap = 0;
    // The 8-byte image of the table (partition.hip.h, region_count_kernel32<.., NARROW>): count << 32 | 32-bit payload per
    // slot.  While `narrow` is set IT holds the counts and the 16-byte table is stale; ensure_wide() converts.  A fresh
    // partitioned pass with 32-bit payloads writes it, later such passes update it, kh_finish / kh_result_* / kh_histogram /
    // kh_lookup read it as it is; everything else (the direct path, growth, exports, merges) goes through enter(), which
    // widens first.
    u64 *ntab = nullptr;
    u64 ntab_cap = 0;
    bool narrow = false;
    bool narrow_banned = false;   // a count left 32 bits once: this table stays 16-byte until kh_reset
    kh::PartGeom narrow_g;        // the geometry the image's payloads are relative to
    Counters *d_ctr = nullptr;
    Counters *h_ctr = nullptr;  // pinned

    u64 distinct_known = 0;  // exact as of the last counter read-back
    u64 pending_bound = 0;   // upper bound on claims by launches since then
    u64 bases_pushed = 0;
    u64 grows = 0;
    u64 launches = 0;
    double kernel_ms = 0.0;
    double h2d_ms = 0.0;

    // ---- kh_push: pinned staging -> device accumulation buffers -> one count per filled buffer ----
    hipStream_t cstream = nullptr;             // copy stream (H2D overlaps counting on `stream`)
    uint8_t *h_stage[2] = {nullptr, nullptr};  // pinned: bases then qual, each STAGE_BYTES
    hipEvent_t stage_done[2] = {nullptr, nullptr};
    bool stage_used[2] = {false, false};
    int stage_next = 0;
    uint8_t *acc[2] = {nullptr, nullptr};      // device: [HALO | bases acc_cap | pad][HALO | qual acc_cap | pad]
    u64 acc_cap = 0;                           // bytes of bases one accumulation buffer holds
    int acc_cur = 0;
    u64 acc_len = 0;                           // bytes accumulated in acc[acc_cur] (after the HALO head)
    u64 acc_carry = 0;                         // HALO bytes at the head are the tail of the previous buffer
    bool acc_qual = false;
    bool acc_has_qual = false;                 // the buffers were allocated with their quality halves
    hipEvent_t acc_free[2] = {nullptr, nullptr};
    bool acc_busy[2] = {false, false};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> h2d_events;

    // ---- partitioned path ----
    bool table_empty = true;   // no insert since creation / reset: regions need not be read back
    // kh_set_region_window: the next region-ordered exports / merges cover piece win_piece of win_n of
    // every owner's region range.  A FRESH merge done in pieces leaves the regions of the pieces not
    // yet merged unwritten (stale if the table was lazily reset): win_open / win_mask / win_dirty
    // track that until the last piece, or until anything else touches the table (close_fresh_window).
    uint32_t win_piece = 0, win_n = 1;
    bool win_open = false, win_dirty = false;
    uint32_t win_open_n = 0;
    u64 win_mask = 0;
    bool table_dirty = false;  // kh_reset is lazy: the slots hold stale data that the next operation either
                               // overwrites wholesale (a FRESH region pass) or clears first (everything else)
    bool hinted = false;       // caller gave a capacity hint
    u64 hint_keys = 0;         // ... of this many distinct k-mers
    double new_rate = -1.0;    // new keys per window of the last partitioned range (-1: none yet): sizes an unhinted table
    int path_mode = 0;         // 0 auto, 1 force direct, 2 force partitioned
    int pay_mode = 0;          // 0 auto, 64 = always 64-bit payloads (env KMERHIP_PAYLOAD=64, for A/B)
    uint32_t shard_shift = 0;  // table holds shard `shard_index` of 2^shard_shift (kh_set_shard)
    uint32_t shard_index = 0;
    u64 *merge_off = nullptr;  // scans of the senders' region counts (kh_merge_regions_device)
    u64 merge_off_cap = 0;
    u64 part_budget = 0;       // bytes for the two key buffers (0 = decide at first use)
    uint8_t *keysA = nullptr, *keysB = nullptr;  // partition ping-pong buffers
    u64 key_cap = 0, keyb_cap = 0;  // bytes of keysA / keysB
    kh::Part2Block *blocks = nullptr;
    u64 blocks_cap = 0;
    u64 *moff = nullptr;
    uint32_t *nch = nullptr;
    u64 *info = nullptr;
    uint32_t *H2 = nullptr;
    u64 *O2 = nullptr;
    u64 h2_cap = 0;
    u64 *bstart = nullptr;
    uint8_t *rfail = nullptr;
    uint32_t *rnew = nullptr;
    u64 *rreal = nullptr;            // k-mers per bucket (its size minus the unit-padding sentinels)
    uint32_t *rheads = nullptr;      // exchange heads per region, left by a FRESH region pass
    bool rheads_valid = false;       // ... and still describing the table (nothing else touched it since)
    bool rheads_wide = false;
    uint32_t rheads_cb = 0;
    u64 *bend = nullptr;             // arena path: end of every region's data (the exact path uses bstart + 1)
    uint32_t *hot_list = nullptr;    // [regions] buckets left to hot_buckets_kernel (partition.hip.h)
    u64 *ptotal = nullptr;           // [MAX_P1] payloads per level-1 partition
    uint32_t *pcap = nullptr;        // [MAX_P1] arena capacity of that partition's buckets
    uint8_t *heavy = nullptr;        // [MAX_P1] the partition is too heavy for one workgroup: the exact kernels take it
    u64 *ovf = nullptr;              // [4] overflow list: entries handed out, "list full" flag; heavy partitions, payloads in them
    kh::OvfEntry *ovf_list = nullptr;
    u64 ovf_cap = 0;
    u64 ovf_pending = 0;             // entries of the overflow list still to be inserted (this batch)
    uint16_t *chunk_part = nullptr;  // chunk pool metadata (32-bit payload path)
    uint8_t *fill8 = nullptr;
    uint32_t *plist = nullptr;
    u64 pool_cap = 0;                // chunks the metadata arrays hold
    uint32_t *pcount = nullptr;      // [MAX_P1] chunks per partition, then cursors
    u64 *pstart = nullptr;           // [MAX_P1 + 1]
    u64 *pool_next = nullptr;
    u64 region_cap = 0;
    u64 *scan_partial = nullptr;
    u64 scan_cap = 0;
    u64 *est_set = nullptr;          // scratch of distinct_sample_kernel (partition.hip.h): the set a few level-1 partitions are counted in
    u64 est_set_cap = 0;
    u64 est_keys = 0;                // distinct keys the current fresh range is expected to bring (from that sample; 0 = no estimate)
    bool sized_by_sample = false;    // the table's size comes from such a sample (stats / trace)
    bool estimate_on = true;         // KMERHIP_ESTIMATE=0: never (rounds 1-3's sizing: the hint, or the worst case)
    u64 part_batches = 0;
    double stage_ms[ST_N] = {0};
    struct StageEv { int stage; hipEvent_t a, b; };
    std::vector<StageEv> stage_events;

    // ---- kh_push_text: device-side record scanning ----
    uint8_t *txt_raw2[2] = {nullptr, nullptr};  u64 txt_raw2_cap[2] = {0, 0};  // host text lands here (two: KH_FLAG_DEFER_TEXT_SCAN copies one while the other is scanned)
    int txt_raw_next = 0;
    hipStream_t sstream = nullptr;         // KH_FLAG_DEFER_TEXT_SCAN: the stream the scans run on, beside the copy stream
    hipEvent_t txt_copied[2] = {nullptr, nullptr};
    hipEvent_t txt_scanned[2] = {nullptr, nullptr};  // the scan kernels that read raw buffer r are done (recorded on the scan stream)
    bool txt_scanned_on[2] = {false, false};
    hipStream_t cstream2 = nullptr;        // a second copy stream: a large pinned text travels as two halves on two DMA engines
    struct { bool on = false; int r = 0; u64 n = 0; int format = 0; } txt_unscanned;  // a text on the device whose scan is still to come
    uint8_t *txt_acc[2] = {nullptr, nullptr};   u64 txt_acc_cap[2] = {0, 0};    // flat bases of the texts pushed, accumulated for the count kernels
    uint8_t *txt_accq[2] = {nullptr, nullptr};  u64 txt_accq_cap[2] = {0, 0};   // ... and their qualities
    int txt_cur = 0;                       // the buffer the scans append to
    u64 txt_acc_len = 0;                   // bytes accumulated there and not counted yet (a multiple of 16)
    bool txt_acc_qual = false;             // ... with qualities
    hipEvent_t txt_acc_done[2] = {nullptr, nullptr};  // the count of that buffer's last content (on `stream`)
    bool txt_acc_busy[2] = {false, false};
    hipStream_t txt_scan_stream = nullptr; // the stream the accumulated scans ran on
    u64 *txt_scan_partial = nullptr;  u64 txt_scan_cap = 0;  // scan scratch of the text stream
    u64 expect_bytes = 0;                  // kh_config::input_mib: what the caller expects to push in total (0 = unknown)
    u64 *txt_ls = nullptr;        u64 txt_ls_cap = 0;    // line starts
    uint8_t *txt_hdr = nullptr;   u64 txt_hdr_cap = 0;   // FASTA: line is a header
    uint32_t *txt_tnl = nullptr;  u64 txt_tnl_cap = 0;   // per-tile newline counts
    u64 *txt_tbase = nullptr;     u64 txt_tbase_cap = 0;
    uint32_t *txt_tkeep = nullptr; u64 txt_tkeep_cap = 0;
    u64 *txt_tout = nullptr;      u64 txt_tout_cap = 0;
    uint32_t *txt_err = nullptr;  u64 txt_err_cap = 0;
    struct TxtHost { u64 total; u64 end_mark; uint32_t err; uint8_t first, last; } *h_txt = nullptr;  // pinned
    double text_ms = 0.0;

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

int flush_acc(kh_ctx *c, bool carry);
int clear_if_dirty(kh_ctx *c);
bool is_pinned_host(const void *p);
void comm_release(kh_ctx *c);

// Every entry point starts here.  Host pushes are accumulated on the device and counted lazily;
// anything that looks at the table first counts what is pending.  need_table = false: the caller
// decides itself whether a lazily reset table must be cleared (the input entry points: a
// partitioned batch into an empty table overwrites every region anyway).
int close_fresh_window(kh_ctx *c);

int ensure_wide(kh_ctx *c);

int flush_text(kh_ctx *c);
int scan_unscanned(kh_ctx *c);
int need_table(kh_ctx *c);
int enter(kh_ctx *c, bool flush_pending = true, bool need_table = true, bool keep_window = false, bool narrow_ok = false) {
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
    hipError_t e = hipMalloc((void **)&c->table, c->cap * sizeof(Slot));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c->table = nullptr;
        return fail(c, KH_ERR_OOM, "hipMalloc(table)", e);
    }
    c->table_dirty = true;
    return KH_OK;
}
// an EMPTY table of another size: nothing to move, nothing to allocate yet
void resize_empty_table(kh_ctx *c, u64 newcap) {
    if (c->table) (void)hipFree(c->table);  // (synchronises the device)
    c->table = nullptr;
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
    HIP_TRY(c, hipFree(c->table));
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

// ---- stage timing: HIP events on the launch stream, resolved lazily ---------------------------
struct StageTimer {
    kh_ctx *c;
    int stage;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st_stream;
    StageTimer(kh_ctx *ctx, int st, hipStream_t s = nullptr) : c(ctx), stage(st), st_stream(s ? s : ctx->stream) {
        if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) (void)hipEventRecord(a, st_stream);
    }
    void stop() {
        if (a && b) {
            (void)hipEventRecord(b, st_stream);
            c->stage_events.push_back({stage, a, b});
            a = b = nullptr;
        }
    }
    ~StageTimer() { stop(); }
};

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

// ---- device scratch management for the partitioned path --------------------------------------
double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename T>
int ensure_buf(kh_ctx *c, T **ptr, u64 *cap, u64 need, const char *what) {
    if (*cap >= need && *ptr) return KH_OK;
    const double t0 = c->trace ? wall_ms() : 0.0;
    struct Tr {
        kh_ctx *c; double t0; const char *what; u64 bytes;
        ~Tr() { if (c->trace && wall_ms() - t0 > 5.0) fprintf(stderr, "[kmerhip] %s: %.1f MB took %.1f ms\n", what, (double)bytes / 1e6, wall_ms() - t0); }
    } tr{c, t0, what, need * sizeof(T)};
    if (*ptr) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        (void)hipFree(*ptr);
        *ptr = nullptr;
        *cap = 0;
    }
    hipError_t e = hipMalloc((void **)ptr, need * sizeof(T));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, KH_ERR_OOM, what, e);
    }
    *cap = need;
    return KH_OK;
}

// Two-level split of the region index.  use32 = the 32-bit payload format applies (the hash bits
// left after level 1 fit in 32).  ok = false when the table is too large for two levels.
struct GeomChoice {
    kh::PartGeom g;
    bool use32;
    bool ok;
};

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
int head_count_bits(const kh_ctx *c, u64 regions) {
    const int hb = kh::kh_below_bits(c->k, 0, kh::kh_geom_of_regions(regions));  // hash bits below the region index of an UNSHARDED table
    return (hb >= 1 && hb <= 28) ? 32 - hb : -1;  // at least 4 count bits
}

// region rebuild launch, by payload type
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
template <>
void launch_region<u64>(kh_ctx *c, const kh::PartGeom &g, u64 nregions, u64 hot, const u64 *bend, bool, u64 skip, u64 expect) {
    const kh::TableGeom tg = table_geom(c, c->table, c->cap);
#define KH_REGION64(FRESH, NT) \
    hipLaunchKernelGGL((kh::region_count_kernel64<FRESH, NT>), dim3((unsigned)nregions), dim3(NT), 0, c->stream, tg, \
                       (const u64 *)c->keysB, bend, (const u64 *)c->bstart, c->rfail, c->rnew, hot, (uint32_t)c->table_dirty, c->rreal, skip)
    if (c->table_empty && region_small_groups(c, expect, nregions)) KH_REGION64(true, 512);
    else if (c->table_empty) KH_REGION64(true, kh::REGION_NT);
    else KH_REGION64(false, kh::REGION_NT);
#undef KH_REGION64
}
template <>
void launch_region<uint32_t>(kh_ctx *c, const kh::PartGeom &g, u64 nregions, u64 hot, const u64 *bend, bool narrow, u64 skip, u64 expect) {
    const kh::TableGeom tg = table_geom(c, c->table, c->cap);
    const int cb = (c->table_empty && !c->shard_shift) ? head_count_bits(c, nregions) : -1;
    c->rheads_cb = cb > 0 ? (uint32_t)cb : 0u;
    // (a narrow FRESH pass must write every region of the image whatever the table held: dirty = 1)
    const bool pow2 = g.p2_bits != 0xFFFFFFFFu;  // (the power-of-two instances take digit and start by shifts: rounds 1-3's code)
#define KH_REGION32_P(FRESH, NARROW, NT, DIRTY, CB, RH, P2) \
    hipLaunchKernelGGL((kh::region_count_kernel32<FRESH, NARROW, NT, P2>), dim3((unsigned)nregions), dim3(NT), 0, c->stream, tg, g, \
                       (const uint32_t *)c->keysB, bend, (const u64 *)c->bstart, c->rfail, c->rnew, hot, DIRTY, CB, RH, c->d_ctr, c->rreal, c->ntab, skip)
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

template <typename PT>
int partition_batch(kh_ctx *c, const RangeArgs &ra, GeomChoice &gc, u64 tile0, u64 ntiles, bool size_from_sample, double range_scale) {
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
                    const u64 room = ((u64)fr + (c->table ? c->cap * sizeof(Slot) : 0) + (c->ntab ? c->ntab_cap * sizeof(u64) : 0)) / 3;
                    while (newcap > c->cap && newcap * sizeof(Slot) > room) newcap = round_cap((double)newcap * 0.8);
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
    if (c->region_cap < nregions) {
        u64 z = c->bstart ? c->region_cap + 1 : 0;
        if ((rc = ensure_buf(c, &c->bstart, &z, nregions + 1, "hipMalloc(bstart)")) != KH_OK) return rc;
        z = c->rfail ? c->region_cap : 0;
        if ((rc = ensure_buf(c, &c->rfail, &z, nregions, "hipMalloc(rfail)")) != KH_OK) return rc;
        z = c->rnew ? c->region_cap : 0;
        if ((rc = ensure_buf(c, &c->rnew, &z, nregions, "hipMalloc(rnew)")) != KH_OK) return rc;
        z = c->rheads ? c->region_cap : 0;
        if ((rc = ensure_buf(c, &c->rheads, &z, nregions, "hipMalloc(rheads)")) != KH_OK) return rc;
        z = c->rreal ? c->region_cap : 0;
        if ((rc = ensure_buf(c, &c->rreal, &z, nregions, "hipMalloc(rreal)")) != KH_OK) return rc;
        z = c->bend ? c->region_cap : 0;
        if ((rc = ensure_buf(c, &c->bend, &z, nregions, "hipMalloc(bend)")) != KH_OK) return rc;
        z = c->hot_list ? c->region_cap : 0;
        if ((rc = ensure_buf(c, &c->hot_list, &z, nregions, "hipMalloc(hot_list)")) != KH_OK) return rc;
        c->region_cap = nregions;
    }
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
    const u64 b_bytes = std::max((n_pay + pad_ub) * (u64)sizeof(PT), arena ? (heavy_base + heavy_room + heavy_pad + 64) * (u64)sizeof(PT) : 0);
    if (c->keyb_cap < b_bytes) {
        u64 z = c->keysB ? c->keyb_cap : 0;
        if ((rc = ensure_buf(c, &c->keysB, &z, b_bytes, "hipMalloc(keysB)")) != KH_OK) return rc;
        c->keyb_cap = b_bytes;
    }
    PT *bufA = reinterpret_cast<PT *>(c->keysA), *bufB = reinterpret_cast<PT *>(c->keysB);
    if (arena && c->ovf_cap < ovf_need) {
        u64 z = c->ovf_list ? c->ovf_cap : 0;
        if ((rc = ensure_buf(c, &c->ovf_list, &z, ovf_need, "hipMalloc(ovf_list)")) != KH_OK) return rc;
        c->ovf_cap = ovf_need;
    }
    {
        StageTimer t(c, ST_MISC);
        if (size_from_sample)  // (b2 is final now: the plan's matrix offsets again)
            hipLaunchKernelGGL(kh::part2_plan_chunked_kernel, dim3(1), dim3(1024), 0, c->stream, (const u64 *)c->pstart, g,
                               c->blocks, max_blocks, c->moff, c->nch, c->info, c->pcount, force_wide, (const uint8_t *)nullptr);
        HIP_TRY(c, hipMemsetAsync(c->H2, 0, n2 * sizeof(uint32_t), c->stream));
    }
    const u64 *bend = c->bstart + 1;  // end of region r's data: the next region's start (exact path) or c->bend[r] (arenas)
    bool arena_done = false, heavy_exact = false;
    const u64 ovf_test_cap = c->knobs.l2_ovf_cap;
    const u64 ovf_lim = std::min(c->ovf_cap, ovf_test_cap);
    if (arena) {
        {
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
            if (g.b2 > 768) {  // 769 .. 1024 buckets per partition: the 128 KiB of bins shared out among them, 64-byte units, four buckets per lane group
                if (pow2) KH_ARENA(64, 1024, true);
                else KH_ARENA(64, 1024, false);
            } else if (g.b2 > 512) {  // 513 .. 768: three buckets per lane group
#ifndef KH_ARENA_UNITB_768
#define KH_ARENA_UNITB_768 64  // (128: whole lines while a bin holds >= 48 payloads, i.e. up to 682 buckets -- A/B builds)
#endif
                if (KH_ARENA_UNITB_768 == 128 && sizeof(PT) == 4 && g.b2 <= 682) KH_ARENA(128, 768, false);
                else KH_ARENA(64, 768, false);
            } else if (pow2) {
                KH_ARENA(UB512, 512, true);
            } else {
                KH_ARENA(UB512, 512, false);
            }
#undef KH_ARENA
        }
        u64 hov[4] = {0, 0, 0, 0};
        HIP_TRY(c, hipMemcpyAsync(hov, c->ovf, sizeof(hov), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
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
    if (heavy_exact) {
        StageTimer t(c, ST_MISC);
        hipLaunchKernelGGL(kh::part2_plan_chunked_kernel, dim3(1), dim3(1024), 0, c->stream, (const u64 *)c->pstart, g,
                           c->blocks, max_blocks, c->moff, c->nch, c->info, c->pcount, force_wide, (const uint8_t *)c->heavy);
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
    {
        StageTimer t(c, ST_MISC);
        // (three same-address atomics per WAVE at ~10 ns each: a block per 256 regions -- 2048 blocks -- took 0.2 ms for 7 MB)
        hipLaunchKernelGGL(kh::region_reduce_kernel, dim3((unsigned)std::min<u64>(128, (nregions + kh::BLOCK - 1) / kh::BLOCK)), dim3(kh::BLOCK), 0, c->stream,
                           (const u64 *)c->bstart, (const uint8_t *)c->rfail, (const uint32_t *)c->rnew, (const u64 *)c->rreal, (u64)nregions, c->d_ctr);
        // (a long list is mostly copies -- bursts of a tandem repeat's payloads, a repeat family's: summed in LDS first; 4-byte
        //  payloads, and on the 8-byte image only where no count can leave 32 bits: the table's k-mers so far plus this batch's
        //  windows stay below 2^32.  KMERHIP_OVF_AGG=0: never; =1: for lists of any length -- tests)
        const int agg_env = c->knobs.ovf_agg;
        const bool ovf_agg = sizeof(PT) == 4 && agg_env != 0 && (agg_env == 1 || c->ovf_pending >= (1u << 16)) &&
                             (!nar || c->h_ctr->kmers + n_all < 0xFFFFFFFFull);
        if (c->ovf_pending && ovf_agg) {
            const unsigned grid = (unsigned)std::min<u64>(2048, (c->ovf_pending + 8191) / 8192);
            if (nar)
                hipLaunchKernelGGL((kh::ovf_agg_insert_kernel<true>), dim3(grid), dim3(kh::BLOCK), 0, c->stream, table_geom(c, c->table, c->cap), g,
                                   (const kh::OvfEntry *)c->ovf_list, (const u64 *)c->ovf, ovf_lim, c->d_ctr, c->ntab);
            else
                hipLaunchKernelGGL((kh::ovf_agg_insert_kernel<false>), dim3(grid), dim3(kh::BLOCK), 0, c->stream, table_geom(c, c->table, c->cap), g,
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
    c->table_empty = false;
    c->table_dirty = false;  // the FRESH region pass wrote every region
    c->launches++;
    c->part_batches++;

    // exact bookkeeping after every batch (batches are hundreds of ms; one sync is noise)
    rc = sync_counters(c);
    if (rc != KH_OK) return rc;
    // the per-region exchange-head counts of a FRESH 32-bit pass describe the whole table until
    // anything else touches it
    c->rheads_valid = was_empty && sizeof(PT) == 4 && c->rheads_cb != 0 && c->h_ctr->part_failed == 0 && c->ovf_pending == 0;
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
    // (up to 0.78 of what is free: the 8-byte table image -- 8 bytes per slot, allocated after level 2 -- and the small arrays
    //  take the rest.  Round 3 stopped at 160 GiB / 0.75: configs[3]'s 125 M reads then ran as two batches, the second one a
    //  pass over a filled table that re-reads and re-writes all of it: 36 ms of region pass where one fresh pass takes 24)
    // A rank of a multi-GPU merge (a communicator is attached) leaves room for what kh_merge_across allocates while the
    // partition buffers are still there: send and receive buffers (16 B per local key) and the shard's 16-byte table -- about
    // 49 B per local key, 64 GB at configs[3]'s size -- hence 0.55 there: configs[3]'s share then runs as two batches.
    u64 budget = 224ull << 30;
    const double share = c->comm ? 0.55 : 0.78;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) budget = std::min<u64>(budget, (u64)((double)(fr + c->key_cap + c->keyb_cap) * share));
    else (void)hipGetLastError();
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
    ensure_part_budget(c);
    const u64 first_tile = ra.wlo / kh::PART_TILE;
    const u64 end_tile = (ra.vend + kh::PART_TILE - 1) / kh::PART_TILE;
    const u64 distinct_before = c->distinct_known;
    for (u64 t = first_tile; t < end_tile;) {
        const GeomChoice gc = make_geom(c, c->cap);  // re-evaluated per batch: the table may have grown
        if (!gc.ok) return direct_range(c, ra, t * (kh::PART_TILE / kh::TILE), (ra.vend + kh::TILE - 1) / kh::TILE);
        // bytes per key over the two buffers and the overflow list: pool (1.04 x payload) + arenas (1.25 x + 1) + 1
        const u64 per_key = gc.use32 ? 11 : 20;
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
        int rc = gcb.use32 ? partition_batch<uint32_t>(c, ra, gcb, t, nt, from_sample, range_scale) : partition_batch<u64>(c, ra, gcb, t, nt, from_sample, range_scale);
        if (rc == KH_RETRY_FULL_SIZE) {  // the sample misjudged these tiles: the rest of the range is sized for every window
            ra.survive = 1.0;
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
// CPUs this process may really use: the visible ones capped by the cgroup quota (the GPU box shows 256 and grants 16)
unsigned usable_cpus() {
    unsigned t = std::thread::hardware_concurrency();
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        unsigned long long period = 0;
        if (fscanf(f, "%31s %llu", q, &period) == 2 && strcmp(q, "max") != 0 && period) {
            const unsigned long long quota = strtoull(q, nullptr, 10);
            if (quota) t = std::min<unsigned>(t, (unsigned)std::max<unsigned long long>(1, quota / period));
        }
        fclose(f);
    }
    return t < 1 ? 1u : t;
}

int g_copy_threads = 0;  // KMERHIP_COPY_THREADS of the first context created (the staging threads are a property of the process)
void staged_memcpy(void *dst, const void *src, size_t n) {
    static const unsigned hw = [] {
        unsigned t = usable_cpus();
        if (t > 6) t = 6;  // (measured on the box, 15 GB pushes / 17 GB results: 6 threads 30 / 24 GB/s, 12 threads 20 / 14 GB/s)
        if (g_copy_threads > 0) t = (unsigned)g_copy_threads;
        return t < 1 ? 1u : t;
    }();
    const size_t min_part = 4u << 20;
    unsigned parts = (unsigned)std::min<size_t>(hw, n / min_part);
    if (parts <= 1) {
        memcpy(dst, src, n);
        return;
    }
    const size_t per = ((n + parts - 1) / parts + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    size_t done_by_threads_from = n;  // [this, n) is copied by helper threads, [0, this) by the caller
    try {  // (no exception may cross the C ABI: if a thread cannot be started the caller copies that part)
        th.reserve(parts - 1);
        for (unsigned i = parts - 1; i >= 1; --i) {
            const size_t off = (size_t)i * per;
            if (off >= n) continue;
            const size_t len = std::min(per, n - off);
            th.emplace_back([=] { memcpy((char *)dst + off, (const char *)src + off, len); });
            done_by_threads_from = off;
        }
    } catch (...) {
    }
    memcpy(dst, src, done_by_threads_from);
    for (auto &t : th) t.join();
}

int ensure_stage(kh_ctx *c) {
    if (!c->cstream) HIP_TRY(c, hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        if (!c->h_stage[i]) {
            hipError_t e = hipHostMalloc((void **)&c->h_stage[i], 2 * STAGE_BYTES, hipHostMallocDefault);
            if (e != hipSuccess) return fail(c, KH_ERR_OOM, "hipHostMalloc(stage)", e);
            HIP_TRY(c, hipEventCreateWithFlags(&c->stage_done[i], hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&c->acc_free[i], hipEventDisableTiming));
        }
    }
    return KH_OK;
}

// Device -> pageable host memory through the two pinned staging buffers: the D2H of chunk i+1 runs
// while chunk i is copied out (by several threads: first-touch page faults of a fresh destination
// array cost more than the copy itself).
int d2h_staged(kh_ctx *c, void *dst, const void *d_src, u64 bytes) {
    int rc = ensure_stage(c);
    if (rc != KH_OK) return rc;
    if (is_pinned_host(dst)) {  // a registered destination takes the DMA itself: no bounce, no first-touch faults
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // d_src was produced on the compute stream
        HIP_TRY(c, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->cstream));
        HIP_TRY(c, hipStreamSynchronize(c->cstream));
        return KH_OK;
    }
    // A fresh destination array is all first-touch page faults (they, not the copy, were most of the time of
    // kh_result_copy): ask for transparent huge pages on its page-aligned interior -- a hint, errors are ignored.
    if (bytes >= (64ull << 20)) {
        const uintptr_t lo = ((uintptr_t)dst + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
        const uintptr_t hi = ((uintptr_t)dst + bytes) & ~(uintptr_t)((2u << 20) - 1);
        if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // d_src was produced on the compute stream
    HIP_TRY(c, hipStreamSynchronize(c->cstream));  // the staging buffers are free
    const u64 CH = 2 * STAGE_BYTES;
    const u64 nch = (bytes + CH - 1) / CH;
    auto issue = [&](u64 i) -> hipError_t {
        const u64 off = i * CH, len = std::min(CH, bytes - off);
        hipError_t e = hipMemcpyAsync(c->h_stage[i & 1], (const char *)d_src + off, len, hipMemcpyDeviceToHost, c->cstream);
        if (e == hipSuccess) e = hipEventRecord(c->stage_done[i & 1], c->cstream);
        return e;
    };
    if (nch) HIP_TRY(c, issue(0));
    for (u64 i = 0; i < nch; ++i) {
        HIP_TRY(c, hipEventSynchronize(c->stage_done[i & 1]));
        if (i + 1 < nch) HIP_TRY(c, issue(i + 1));
        const u64 off = i * CH, len = std::min(CH, bytes - off);
        staged_memcpy((char *)dst + off, c->h_stage[i & 1], len);
    }
    c->stage_used[0] = c->stage_used[1] = false;  // nothing in flight on the staging buffers any more
    return KH_OK;
}

u64 acc_stride(u64 cap) { return HALO + cap + 64; }  // one of the two halves (bases / qual) of a buffer

// Largest accumulation buffer this device affords: a power of two, the two buffers with their quality halves within a
// quarter of what is free now (plus what the current buffers hold), never above ACC_MAX.
u64 acc_limit(const kh_ctx *c) {
    size_t fr = 0, tot = 0;
    u64 lim = ACC_MAX;
    if (c->knobs.acc_max_mb) lim = std::max<u64>(ACC_MIN, c->knobs.acc_max_mb << 20);  // (small buffers exercise the seams)
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
        // what a batch of `lim` bases takes besides the table: two accumulation buffers with their quality halves (4 x)
        // and the partition buffers and overflow list of its ~lim windows (11 B per window with 4-byte payloads, 20 with
        // 8-byte ones).  What this context already holds of those counts as available: it is what they would be made of.
        const u64 held = (c->acc_cap ? (c->acc_has_qual ? 4 : 2) * acc_stride(c->acc_cap) : 0) + c->key_cap + c->keyb_cap;
        const u64 avail = (u64)fr + held;
        while (lim > ACC_MIN && 24 * lim > avail - avail / 8) lim /= 2;
    } else {
        (void)hipGetLastError();
    }
    return lim;
}

// Host memory the device can DMA from / into directly: hipHostMalloc'ed (kh_host_alloc) or hipHostRegister'ed
// (kh_host_register, or the caller's own).  Pageable memory goes through the pinned staging chunks instead.
bool is_pinned_host(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // (older runtimes: "invalid value" for memory they do not know)
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// (Re)allocates the two accumulation buffers for `cap` bytes of bases each (and as many quality bytes if with_qual).
// Only when empty.
int alloc_acc(kh_ctx *c, u64 cap, bool with_qual) {
    HIP_TRY(c, hipStreamSynchronize(c->cstream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->acc_has_qual = with_qual;
    for (int i = 0; i < 2; ++i) {
        if (c->acc[i]) (void)hipFree(c->acc[i]);
        c->acc[i] = nullptr;
        c->acc_busy[i] = false;
        hipError_t e = hipMalloc((void **)&c->acc[i], (with_qual ? 2 : 1) * acc_stride(cap));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            c->acc_cap = 0;
            return fail(c, KH_ERR_OOM, "hipMalloc(accumulation buffer)", e);
        }
    }
    c->acc_cap = cap;
    c->acc_cur = 0;
    return KH_OK;
}

// Counts what the current accumulation buffer holds and switches to the other one.  carry: the
// flush falls inside a push, so the last HALO bytes are re-presented at the head of the next buffer
// (windows that straddle the seam are counted there, once).
int flush_acc(kh_ctx *c, bool carry) {
    if (c->acc_len == 0 && !carry) return KH_OK;
    const int cur = c->acc_cur, nxt = cur ^ 1;
    const u64 stride = acc_stride(c->acc_cap);
    hipEvent_t ready;
    HIP_TRY(c, hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(ready, c->cstream));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, ready, 0));
    (void)hipEventDestroy(ready);
    const u64 head = HALO - c->acc_carry;
    const u64 len = c->acc_carry + c->acc_len;
    const u64 acc_len = c->acc_len;
    c->acc_len = 0;  // (count_device_range re-enters nothing, but keep the state consistent on errors)
    int rc = count_device_range(c, c->acc[cur] + head, c->acc_qual ? c->acc[cur] + stride + head : nullptr, len, c->acc_carry);
    if (rc != KH_OK) return rc;
    if (carry) {  // tail -> head of the next buffer, on the compute stream (ordered after the count)
        if (c->acc_busy[nxt]) HIP_TRY(c, hipEventSynchronize(c->acc_free[nxt]));
        HIP_TRY(c, hipMemcpyAsync(c->acc[nxt], c->acc[cur] + HALO + acc_len - HALO, HALO, hipMemcpyDeviceToDevice, c->stream));
        if (c->acc_qual)
            HIP_TRY(c, hipMemcpyAsync(c->acc[nxt] + stride, c->acc[cur] + stride + HALO + acc_len - HALO, HALO,
                                      hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(c, hipEventRecord(c->acc_free[cur], c->stream));
    c->acc_busy[cur] = true;
    if (c->acc_busy[nxt]) {  // the copy stream may not overwrite a buffer that is still being counted
        HIP_TRY(c, hipStreamWaitEvent(c->cstream, c->acc_free[nxt], 0));
        c->acc_busy[nxt] = false;
    }
    c->acc_cur = nxt;
    c->acc_carry = carry ? HALO : 0;
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
                       c->bstart, c->bend, c->hot_list, c->ptotal, c->pcap, c->heavy, c->ovf, c->ovf_list, c->rfail, c->rnew, c->rreal, c->rheads, c->scan_partial, c->est_set, c->merge_off, c->chunk_part, c->fill8, c->plist,
                       c->pcount, c->pstart, c->pool_next, c->txt_raw2[0], c->txt_raw2[1], c->txt_acc[0], c->txt_acc[1], c->txt_accq[0], c->txt_accq[1], c->txt_scan_partial, c->txt_ls, c->txt_hdr,
                       c->txt_tnl, c->txt_tbase, c->txt_tkeep, c->txt_tout, c->txt_err};
    for (void *q : scratch)
        if (q) (void)hipFree(q);
    if (c->table) (void)hipFree(c->table);
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

// =============================================================================================
// input
// =============================================================================================
extern "C" int kh_push_device(kh_ctx *c, const uint8_t *d_bases, const uint8_t *d_qual, uint64_t n) {
    int rc = enter(c, true, false, false, true);
    if (rc != KH_OK) return rc;
    if (n && !d_bases) return fail(c, KH_ERR_BAD_ARG, "d_bases is NULL");
    if (c->shard_shift) return fail(c, KH_ERR_STATE, "a shard table only accepts kh_merge_*; kh_reset makes it a full table again");
    rc = count_device_range(c, d_bases, d_qual, n, 0);
    if (rc == KH_OK) c->bases_pushed += n;
    return rc;
}

extern "C" int kh_push(kh_ctx *c, const uint8_t *bases, const uint8_t *qual, uint64_t n) {
    int rc = enter(c, false, false, false, true);
    if (rc != KH_OK) return rc;
    if (n && !bases) return fail(c, KH_ERR_BAD_ARG, "bases is NULL");
    if (c->shard_shift) return fail(c, KH_ERR_STATE, "a shard table only accepts kh_merge_*; kh_reset makes it a full table again");
    if (n == 0) return KH_OK;
    const bool with_qual = (qual != nullptr) && (c->minq >= 0);
    rc = ensure_stage(c);
    if (rc != KH_OK) return rc;
    if (c->acc_len && c->acc_qual != with_qual) {  // a buffer is counted with or without qualities, not both
        rc = flush_acc(c, false);
        if (rc != KH_OK) return rc;
    }
    // size the accumulation buffers for this push (grow-only, 1 MiB .. acc_limit)
    u64 want = ACC_MIN;
    const u64 lim = acc_limit(c);
    while (want < n + 1 && want < lim) want *= 2;
    if (want > c->acc_cap || (with_qual && !c->acc_has_qual)) {
        rc = flush_acc(c, false);
        if (rc == KH_OK) rc = alloc_acc(c, std::max(want, c->acc_cap), with_qual || c->acc_has_qual);
        if (rc != KH_OK) return rc;
    }
    c->acc_qual = with_qual;
    const u64 stride = acc_stride(c->acc_cap);
    // Pinned / registered source (kh_host_alloc, kh_host_register): the copy engine reads the caller's memory itself --
    // no staging memcpy (which, not PCIe, bounded kh_push from pageable memory: ~20-30 against 57 GB/s).
    const bool direct = is_pinned_host(bases) && (!with_qual || is_pinned_host(qual));
    if (direct) {
        hipEvent_t t0, t1;
        HIP_TRY(c, hipEventCreate(&t0));
        HIP_TRY(c, hipEventCreate(&t1));
        HIP_TRY(c, hipEventRecord(t0, c->cstream));
        for (u64 off = 0; off < n;) {
            if (c->acc_len + 1 >= c->acc_cap) {
                rc = flush_acc(c, off != 0);
                if (rc != KH_OK) return rc;
            }
            const u64 len = std::min(n - off, c->acc_cap - c->acc_len - 1);
            uint8_t *dst = c->acc[c->acc_cur] + HALO + c->acc_len;
            HIP_TRY(c, hipMemcpyAsync(dst, bases + off, len, hipMemcpyHostToDevice, c->cstream));
            if (with_qual) HIP_TRY(c, hipMemcpyAsync(dst + stride, qual + off, len, hipMemcpyHostToDevice, c->cstream));
            c->acc_len += len;
            off += len;
        }
        HIP_TRY(c, hipMemsetAsync(c->acc[c->acc_cur] + HALO + c->acc_len, '\n', 1, c->cstream));
        if (with_qual) HIP_TRY(c, hipMemsetAsync(c->acc[c->acc_cur] + stride + HALO + c->acc_len, '\n', 1, c->cstream));
        c->acc_len += 1;
        HIP_TRY(c, hipEventRecord(t1, c->cstream));
        c->h2d_events.emplace_back(t0, t1);
        HIP_TRY(c, hipStreamSynchronize(c->cstream));  // the caller may reuse its buffers when this returns
        c->bases_pushed += n;
        return KH_OK;
    }
    for (u64 off = 0; off < n;) {
        // (an accumulation buffer may be SMALLER than a staging chunk -- KMERHIP_ACC_MAX_MB, or little free device memory:
        //  acc_limit() -- so a chunk is cut to the room that is left, +1 for the separator appended after the push)
        const u64 want = std::min(STAGE_BYTES, n - off);
        if (c->acc_len && c->acc_len + want + 1 > c->acc_cap) {
            rc = flush_acc(c, off != 0);          // inside a push the seam needs the k-1 look-back
            if (rc != KH_OK) return rc;
        }
        const u64 len = std::min(want, c->acc_cap - c->acc_len - 1);
        const int p = c->stage_next;
        c->stage_next ^= 1;
        if (c->stage_used[p]) HIP_TRY(c, hipEventSynchronize(c->stage_done[p]));
        staged_memcpy(c->h_stage[p], bases + off, len);
        if (with_qual) staged_memcpy(c->h_stage[p] + STAGE_BYTES, qual + off, len);
        hipEvent_t t0, t1;
        HIP_TRY(c, hipEventCreate(&t0));
        HIP_TRY(c, hipEventCreate(&t1));
        HIP_TRY(c, hipEventRecord(t0, c->cstream));
        uint8_t *dst = c->acc[c->acc_cur] + HALO + c->acc_len;
        HIP_TRY(c, hipMemcpyAsync(dst, c->h_stage[p], len, hipMemcpyHostToDevice, c->cstream));
        if (with_qual) HIP_TRY(c, hipMemcpyAsync(dst + stride, c->h_stage[p] + STAGE_BYTES, len, hipMemcpyHostToDevice, c->cstream));
        HIP_TRY(c, hipEventRecord(t1, c->cstream));
        HIP_TRY(c, hipEventRecord(c->stage_done[p], c->cstream));
        c->stage_used[p] = true;
        c->h2d_events.emplace_back(t0, t1);
        c->acc_len += len;
        off += len;
    }
    // k-mers never span pushes: a separator byte follows the last record of every push
    HIP_TRY(c, hipMemsetAsync(c->acc[c->acc_cur] + HALO + c->acc_len, '\n', 1, c->cstream));
    if (with_qual) HIP_TRY(c, hipMemsetAsync(c->acc[c->acc_cur] + stride + HALO + c->acc_len, '\n', 1, c->cstream));
    c->acc_len += 1;
    c->bases_pushed += n;
    return KH_OK;
}

// ---- raw text: records are found on the device (rawparse.hip.h) ------------------------------
// Round 4: scanned text ACCUMULATES on the device -- the flat bases (and qualities) of push after push, appended in one of
// two buffers of up to an eighth of the free memory -- and is counted when a buffer is full or something looks at the table
// (flush_text).  A file streamed through kh_push_text in 256 MiB chunks used to be 120 counting batches into a growing
// table (device atomics for most of them, 17 G k-mers/s); now it is one or a few partitioned batches at the rate of the
// resident benchmark, the first of them FRESH and its table sized from the level-1 sample (partition_batch).
// The scan kernels run on the COPY stream, right behind the text's own transfer: copies and scans of later texts go on
// while an accumulated buffer is being counted on the context's stream (the other buffer takes them).
namespace {

int text_fail(kh_ctx *c, const char *why) { return fail(c, KH_ERR_FORMAT, why); }

// bytes of text one accumulation buffer may hold: per text byte there are two buffers' worth of flat bases (+ qualities)
// and the partition buffers of its ~0.45 surviving windows (11 B each) -- an eighth (a tenth) of what is free
u64 text_acc_limit(const kh_ctx *c, bool with_qual) {
    u64 lim = 40ull << 30;
    if (c->knobs.text_acc_mb) return std::max<u64>(1ull << 20, c->knobs.text_acc_mb << 20);  // (small buffers exercise the switch-over)
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
        u64 held = c->key_cap + c->keyb_cap;
        for (int i = 0; i < 2; ++i) held += c->txt_acc_cap[i] + c->txt_accq_cap[i];
        lim = std::min<u64>(lim, ((u64)fr + held) / (with_qual ? 10 : 8));
    } else {
        (void)hipGetLastError();
    }
    return std::max<u64>(lim & ~((1ull << 20) - 1), 64ull << 20);
}

// exclusive scan on the text stream (its own scratch: the context's stream may be scanning for a counting batch meanwhile)
int text_device_scan(kh_ctx *c, hipStream_t s, const uint32_t *in, u64 n, u64 *out) {
    const u64 nb = (n + kh::SCAN_CHUNK - 1) / kh::SCAN_CHUNK;
    if (c->txt_scan_cap < nb + 2) {
        HIP_TRY(c, hipStreamSynchronize(s));
        if (c->txt_scan_partial) (void)hipFree(c->txt_scan_partial);
        c->txt_scan_partial = nullptr;
        c->txt_scan_cap = 0;
        if (hipMalloc((void **)&c->txt_scan_partial, (nb + 2) * 2 * sizeof(u64)) != hipSuccess) {
            (void)hipGetLastError();
            return fail(c, KH_ERR_OOM, "hipMalloc(text scan)");
        }
        c->txt_scan_cap = (nb + 2) * 2;
    }
    hipLaunchKernelGGL(kh::scan_partials_kernel, dim3((unsigned)nb), dim3(kh::SCAN_NT), 0, s, in, n, c->txt_scan_partial);
    hipLaunchKernelGGL(kh::scan_spine_kernel, dim3(1), dim3(1024), 0, s, c->txt_scan_partial, nb);
    hipLaunchKernelGGL(kh::scan_apply_kernel, dim3((unsigned)nb), dim3(kh::SCAN_NT), 0, s, in, n, (const u64 *)c->txt_scan_partial, out);
    HIP_TRY(c, hipGetLastError());
    return KH_OK;
}

// d_text: 16-byte aligned device text holding whole records; s: the stream its bytes arrive on (the scan runs there).
// Appends the flat form to the current accumulation buffer.
int scan_text(kh_ctx *c, const uint8_t *d_text, u64 n, int format, hipStream_t s, bool counted_at_once = false) {
    const bool fastq = format == KH_TEXT_FASTQ;
    const bool with_qual = fastq && c->minq >= 0;
    const u64 ntiles = (n + kh::RAW_TILE - 1) / kh::RAW_TILE;
    const u64 need = (n + 15) / 16 * 16 + 64;  // (FASTQ: as many bytes as the text; FASTA: at most)
    int rc;
    if (!c->h_txt) {
        hipError_t e = hipHostMalloc((void **)&c->h_txt, sizeof(*c->h_txt), hipHostMallocDefault);
        if (e != hipSuccess) return fail(c, KH_ERR_OOM, "hipHostMalloc(text scan)", e);
    }
    // room in the current buffer -- else what it holds is counted and the other buffer takes over
    if (c->txt_acc_len && (c->txt_acc_qual != with_qual || c->txt_acc_len + need > c->txt_acc_cap[c->txt_cur])) {
        if ((rc = flush_text(c)) != KH_OK) return rc;
    }
    if (c->txt_acc_len == 0) {
        // a fresh accumulation: the buffer that exists and is idle, rather than a new allocation (after a reset the other
        // buffer would be "next": tens of GB allocated for nothing -- and a process that allocates while another one's
        // memory is still being reclaimed waits for that: 4 s of a bench step, measured)
        for (int i = 0; i < 2; ++i)
            if (c->txt_acc_busy[i] && hipEventQuery(c->txt_acc_done[i]) == hipSuccess) c->txt_acc_busy[i] = false;
        (void)hipGetLastError();
        const int o = c->txt_cur ^ 1;
        const bool cur_ok = !c->txt_acc_busy[c->txt_cur] && c->txt_acc_cap[c->txt_cur] >= need && (!with_qual || c->txt_accq_cap[c->txt_cur] >= need);
        const bool oth_ok = !c->txt_acc_busy[o] && c->txt_acc_cap[o] >= need && (!with_qual || c->txt_accq_cap[o] >= need);
        if (!cur_ok && oth_ok) c->txt_cur = o;
        else if (cur_ok && oth_ok && c->txt_acc_cap[o] > c->txt_acc_cap[c->txt_cur]) c->txt_cur = o;
    }
    const int cur = c->txt_cur;
    if (c->txt_acc_busy[cur]) {  // its last content is still being counted on the context's stream
        HIP_TRY(c, hipStreamWaitEvent(s, c->txt_acc_done[cur], 0));
        c->txt_acc_busy[cur] = false;
    }
    if (c->txt_acc_cap[cur] < c->txt_acc_len + need || (with_qual && c->txt_accq_cap[cur] < c->txt_acc_len + need)) {
        // (only ever grown when empty: its content cannot be moved.  A first text of n bytes gets room for 128 like it, within the limit)
        const u64 lim = text_acc_limit(c, with_qual);
        // How much: what the caller says it will push (kh_config::input_mib); else, for a text of 32 MiB or more -- a chunk of a
        // file being streamed -- the whole limit (one big batch instead of several), for a small one 128 like it.  A resident
        // text (kh_push_text_device) is counted at once: exactly its size.
        u64 want = c->expect_bytes ? c->expect_bytes + (c->expect_bytes >> 6) + need : (n >= (32ull << 20) ? lim : 128 * need);
        if (counted_at_once) want = need;
        want = std::max<u64>(need, std::min<u64>(lim, std::max<u64>(want, counted_at_once ? 0 : c->txt_acc_cap[cur ^ 1])));
        want = std::max<u64>(want, c->txt_acc_cap[cur]);
        if (c->txt_acc_cap[cur] < want) {
            const double ta = wall_ms();
            HIP_TRY(c, hipStreamSynchronize(s));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            const double tb = wall_ms();
            if (c->txt_acc[cur]) (void)hipFree(c->txt_acc[cur]);
            c->txt_acc[cur] = nullptr;
            c->txt_acc_cap[cur] = 0;
            hipError_t e = hipMalloc((void **)&c->txt_acc[cur], want);
            if (e != hipSuccess && want > need) {  // (no room for the generous size: what this text needs, then)
                (void)hipGetLastError();
                want = need;
                e = hipMalloc((void **)&c->txt_acc[cur], want);
            }
            if (e != hipSuccess) {
                (void)hipGetLastError();
                return fail(c, KH_ERR_OOM, "hipMalloc(text bases)", e);
            }
            c->txt_acc_cap[cur] = want;
            if (c->trace) fprintf(stderr, "[kmerhip] text accumulation buffer %d: %.1f GB (sync %.1f ms, alloc %.1f ms)\n", cur, (double)want / 1e9, tb - ta, wall_ms() - tb);
        }
        if (with_qual && c->txt_accq_cap[cur] < c->txt_acc_cap[cur]) {
            HIP_TRY(c, hipStreamSynchronize(s));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (c->txt_accq[cur]) (void)hipFree(c->txt_accq[cur]);
            c->txt_accq[cur] = nullptr;
            c->txt_accq_cap[cur] = 0;
            hipError_t e = hipMalloc((void **)&c->txt_accq[cur], c->txt_acc_cap[cur]);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                return fail(c, KH_ERR_OOM, "hipMalloc(text qualities)", e);
            }
            c->txt_accq_cap[cur] = c->txt_acc_cap[cur];
        }
    }
    uint8_t *const out = c->txt_acc[cur] + c->txt_acc_len;  // (16-byte aligned: lengths are kept multiples of 16)
    uint8_t *const outq = with_qual ? c->txt_accq[cur] + c->txt_acc_len : nullptr;
    // (scratch of the scan: sized per text, reallocated only when a larger text comes -- on the text stream)
    auto tbuf = [&](auto **ptr, u64 *cap, u64 want, const char *what) -> int {
        if (*cap >= want && *ptr) return KH_OK;
        HIP_TRY(c, hipStreamSynchronize(s));
        if (*ptr) (void)hipFree(*ptr);
        *ptr = nullptr;
        *cap = 0;
        if (hipMalloc((void **)ptr, want * sizeof(**ptr)) != hipSuccess) {
            (void)hipGetLastError();
            return fail(c, KH_ERR_OOM, what);
        }
        *cap = want;
        return KH_OK;
    };
    if ((rc = tbuf(&c->txt_tnl, &c->txt_tnl_cap, ntiles, "hipMalloc(text tiles)")) != KH_OK) return rc;
    if ((rc = tbuf(&c->txt_tbase, &c->txt_tbase_cap, ntiles + 1, "hipMalloc(text tiles)")) != KH_OK) return rc;
    if ((rc = tbuf(&c->txt_err, &c->txt_err_cap, (u64)4, "hipMalloc(text err)")) != KH_OK) return rc;
    const unsigned grid = (unsigned)std::min<u64>(ntiles, (u64)GRID_CAP);
    u64 out_len = 0;
    {
        StageTimer tm(c, ST_TEXT, s);
        hipLaunchKernelGGL(kh::raw_nl_count_kernel, dim3(grid), dim3(kh::BLOCK), 0, s, d_text, n, ntiles, c->txt_tnl);
        if ((rc = text_device_scan(c, s, c->txt_tnl, ntiles, c->txt_tbase)) != KH_OK) return rc;
        HIP_TRY(c, hipMemsetAsync(c->txt_err, 0, sizeof(uint32_t), s));
        HIP_TRY(c, hipMemcpyAsync(&c->h_txt->total, c->txt_tbase + ntiles, sizeof(u64), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(&c->h_txt->first, d_text, 1, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(&c->h_txt->last, d_text + n - 1, 1, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        const bool open_end = c->h_txt->last != '\n';           // no final newline: the text end closes the line
        const u64 nlines = c->h_txt->total + (open_end ? 1 : 0);
        if (c->h_txt->first != (fastq ? '@' : '>')) return text_fail(c, fastq ? "text does not start with '@'" : "text does not start with '>'");
        if (fastq && (nlines & 3)) return text_fail(c, "FASTQ line count is not a multiple of 4");
        if ((rc = tbuf(&c->txt_ls, &c->txt_ls_cap, nlines + 2, "hipMalloc(line starts)")) != KH_OK) return rc;
        HIP_TRY(c, hipMemsetAsync(c->txt_ls, 0, sizeof(u64), s));
        hipLaunchKernelGGL(kh::raw_line_starts_kernel, dim3(grid), dim3(kh::BLOCK), 0, s, d_text, n, ntiles,
                           (const u64 *)c->txt_tbase, c->txt_ls);
        if (open_end) {
            c->h_txt->end_mark = n + 1;
            HIP_TRY(c, hipMemcpyAsync(c->txt_ls + nlines, &c->h_txt->end_mark, sizeof(u64), hipMemcpyHostToDevice, s));
        }
        if (fastq) {
            const u64 nrec = nlines / 4;
            hipLaunchKernelGGL(kh::fastq_validate_kernel, dim3(grid_for(nrec)), dim3(kh::BLOCK), 0, s, d_text,
                               (const u64 *)c->txt_ls, nrec, c->txt_err);
            HIP_TRY(c, hipMemcpyAsync(&c->h_txt->err, c->txt_err, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            if (c->h_txt->err) return text_fail(c, "not 4-line FASTQ ('@' / '+' markers or |seq| != |qual|)");
            // (only a validated layout is marked: the quality gather reads |seq| bytes from the quality line's start)
            if (with_qual)
                hipLaunchKernelGGL(kh::fastq_mark_kernel<true>, dim3(grid), dim3(kh::BLOCK), 0, s, d_text, n, ntiles,
                                   (const u64 *)c->txt_tbase, (const u64 *)c->txt_ls, out, outq);
            else
                hipLaunchKernelGGL(kh::fastq_mark_kernel<false>, dim3(grid), dim3(kh::BLOCK), 0, s, d_text, n, ntiles,
                                   (const u64 *)c->txt_tbase, (const u64 *)c->txt_ls, out, (uint8_t *)nullptr);
            out_len = n;
        } else {
            if ((rc = tbuf(&c->txt_hdr, &c->txt_hdr_cap, nlines + 2, "hipMalloc(header flags)")) != KH_OK) return rc;
            if ((rc = tbuf(&c->txt_tkeep, &c->txt_tkeep_cap, ntiles, "hipMalloc(text tiles)")) != KH_OK) return rc;
            if ((rc = tbuf(&c->txt_tout, &c->txt_tout_cap, ntiles + 1, "hipMalloc(text tiles)")) != KH_OK) return rc;
            hipLaunchKernelGGL(kh::fasta_headers_kernel, dim3(grid_for(nlines + 1)), dim3(kh::BLOCK), 0, s, d_text, n,
                               (const u64 *)c->txt_ls, nlines + 1, c->txt_hdr);
            hipLaunchKernelGGL(kh::fasta_compact_kernel<0>, dim3(grid), dim3(kh::BLOCK), 0, s, d_text, n, ntiles,
                               (const u64 *)c->txt_tbase, (const uint8_t *)c->txt_hdr, c->txt_tkeep, (const u64 *)nullptr,
                               (uint8_t *)nullptr, c->txt_err);
            if ((rc = text_device_scan(c, s, c->txt_tkeep, ntiles, c->txt_tout)) != KH_OK) return rc;
            HIP_TRY(c, hipMemcpyAsync(&c->h_txt->total, c->txt_tout + ntiles, sizeof(u64), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipMemcpyAsync(&c->h_txt->err, c->txt_err, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            if (c->h_txt->err) return text_fail(c, "blank before a line end, or a CR not followed by LF, inside a FASTA record");
            out_len = c->h_txt->total;
            hipLaunchKernelGGL(kh::fasta_compact_kernel<1>, dim3(grid), dim3(kh::BLOCK), 0, s, d_text, n, ntiles,
                               (const u64 *)c->txt_tbase, (const uint8_t *)c->txt_hdr, (uint32_t *)nullptr,
                               (const u64 *)c->txt_tout, out, (uint32_t *)nullptr);
        }
        HIP_TRY(c, hipGetLastError());
        if (out_len) {  // a separator behind the text, and on to the next multiple of 16
            const u64 end = (c->txt_acc_len + out_len + 1 + 15) & ~15ull;
            HIP_TRY(c, hipMemsetAsync(out + out_len, '\n', end - (c->txt_acc_len + out_len), s));
            if (with_qual) HIP_TRY(c, hipMemsetAsync(outq + out_len, '\n', end - (c->txt_acc_len + out_len), s));
            c->txt_acc_len = end;
            c->txt_acc_qual = with_qual;
            c->txt_scan_stream = s;
        }
    }
    return KH_OK;
}

}  // namespace
namespace {
// counts what the text pushes have accumulated; the other buffer takes what comes next
int flush_text(kh_ctx *c) {
    const u64 n = c->txt_acc_len;
    if (!n) return KH_OK;
    const int cur = c->txt_cur;
    c->txt_acc_len = 0;
    c->txt_cur ^= 1;
    if (c->txt_scan_stream && c->txt_scan_stream != c->stream) {  // the scans that filled the buffer ran on the copy stream
        hipEvent_t ready;
        HIP_TRY(c, hipEventCreateWithFlags(&ready, hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(ready, c->txt_scan_stream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, ready, 0));
        (void)hipEventDestroy(ready);
    }
    const double t0 = wall_ms();
    const int rc = count_device_range(c, c->txt_acc[cur], c->txt_acc_qual ? c->txt_accq[cur] : nullptr, n, 0);
    if (c->trace) fprintf(stderr, "[kmerhip] %.2f GB of accumulated text counted (host side of it: %.1f ms)\n", (double)n / 1e9, wall_ms() - t0);
    if (!c->txt_acc_done[cur]) HIP_TRY(c, hipEventCreateWithFlags(&c->txt_acc_done[cur], hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->txt_acc_done[cur], c->stream));
    c->txt_acc_busy[cur] = true;
    return rc;
}

int text_args(kh_ctx *c, const uint8_t *text, u64 n, int format) {
    if (n && !text) return fail(c, KH_ERR_BAD_ARG, "text is NULL");
    if (format != KH_TEXT_FASTA && format != KH_TEXT_FASTQ) return fail(c, KH_ERR_BAD_ARG, "format must be KH_TEXT_FASTA or KH_TEXT_FASTQ");
    if (c->shard_shift) return fail(c, KH_ERR_STATE, "a shard table only accepts kh_merge_*; kh_reset makes it a full table again");
    return KH_OK;
}

}  // namespace

extern "C" int kh_push_text_device(kh_ctx *c, const uint8_t *d_text, uint64_t n, int format) {
    int rc = enter(c, true, false, false, true);
    if (rc != KH_OK) return rc;
    if ((rc = text_args(c, d_text, n, format)) != KH_OK) return rc;
    if (n == 0) return KH_OK;
    if ((uintptr_t)d_text & 15) return fail(c, KH_ERR_BAD_ARG, "d_text must be 16-byte aligned");
    rc = scan_text(c, d_text, n, format, c->stream, true);
    if (rc == KH_OK) rc = flush_text(c);  // (resident text: counted right away, as kh_push_device counts resident bases)
    if (rc == KH_OK) c->bases_pushed += n;
    return rc;
}

namespace {
// KH_FLAG_DEFER_TEXT_SCAN: the text copied by the previous kh_push_text is scanned now (on the scan stream, behind its copy)
int scan_unscanned(kh_ctx *c) {
    if (!c->txt_unscanned.on) return KH_OK;
    c->txt_unscanned.on = false;
    const int r = c->txt_unscanned.r;
    HIP_TRY(c, hipStreamWaitEvent(c->sstream, c->txt_copied[r], 0));
    const int rc = scan_text(c, c->txt_raw2[r], c->txt_unscanned.n, c->txt_unscanned.format, c->sstream);
    // (the scan's last kernels -- the ones that read the raw text into the accumulation buffer -- are still in flight: the
    //  next copy INTO this raw buffer, on the copy stream, has to wait for them)
    if (!c->txt_scanned[r]) HIP_TRY(c, hipEventCreateWithFlags(&c->txt_scanned[r], hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->txt_scanned[r], c->sstream));
    c->txt_scanned_on[r] = true;
    return rc;
}
}  // namespace

extern "C" int kh_push_text(kh_ctx *c, const uint8_t *text, uint64_t n, int format) {
    // (what earlier calls have accumulated stays where it is: it is counted when its buffer is full, or by whatever looks
    //  at the table next)
    int rc = enter(c, false, false, false, true);
    if (rc != KH_OK) return rc;
    if ((rc = text_args(c, text, n, format)) != KH_OK) return rc;
    if (n == 0) return KH_OK;
    if ((rc = ensure_stage(c)) != KH_OK) return rc;
    if (c->acc_len && (rc = flush_acc(c, false)) != KH_OK) return rc;  // (kh_push's own accumulation: counted first, so that its buffers stay bounded)
    const bool defer = (c->flags & KH_FLAG_DEFER_TEXT_SCAN) != 0;
    if (defer && !c->sstream) HIP_TRY(c, hipStreamCreateWithFlags(&c->sstream, hipStreamNonBlocking));
    // the raw buffer: always [0] when the scan follows the copy on one stream; alternating when the previous text is scanned
    // beside this one's copy
    const int r = defer ? c->txt_raw_next : 0;
    if (defer) c->txt_raw_next ^= 1;
    if (c->txt_raw2_cap[r] < n + 64) {
        u64 want = std::max<u64>(1ull << 20, c->txt_raw2_cap[r]);
        while (want < n + 64) want *= 2;
        HIP_TRY(c, hipStreamSynchronize(c->cstream));  // (the last text's scan read the old buffer)
        if (c->sstream) HIP_TRY(c, hipStreamSynchronize(c->sstream));
        if (c->txt_raw2[r]) (void)hipFree(c->txt_raw2[r]);
        c->txt_raw2[r] = nullptr;
        c->txt_raw2_cap[r] = 0;
        if (hipMalloc((void **)&c->txt_raw2[r], want) != hipSuccess) {
            (void)hipGetLastError();
            return fail(c, KH_ERR_OOM, "hipMalloc(text)");
        }
        c->txt_raw2_cap[r] = want;
    }
    uint8_t *const raw = c->txt_raw2[r];
    if (defer && c->txt_scanned_on[r]) {  // the text this buffer held before is (perhaps) still being read by its scan
        HIP_TRY(c, hipStreamWaitEvent(c->cstream, c->txt_scanned[r], 0));
        c->txt_scanned_on[r] = false;
    }
    // the text -> the device, on the copy stream (behind the previous text's scan where that read the same buffer)
    if (is_pinned_host(text)) {  // pinned / registered text: DMA straight from the caller's memory, no staging memcpy
        // (one DMA engine moves ~42 GB/s from pinned memory, the link takes 57: a large text travels as two halves on two
        //  streams; the copy stream then waits for the second half)
        const u64 half = n >= (64ull << 20) ? ((n / 2) & ~4095ull) : n;
        hipEvent_t t0, t1;
        HIP_TRY(c, hipEventCreate(&t0));
        HIP_TRY(c, hipEventCreate(&t1));
        HIP_TRY(c, hipEventRecord(t0, c->cstream));
        if (half < n) {
            if (!c->cstream2) HIP_TRY(c, hipStreamCreateWithFlags(&c->cstream2, hipStreamNonBlocking));
            hipEvent_t go, done2;
            HIP_TRY(c, hipEventCreateWithFlags(&go, hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&done2, hipEventDisableTiming));
            HIP_TRY(c, hipEventRecord(go, c->cstream));            // (the second stream starts where the copy stream stands: the raw buffer is free)
            HIP_TRY(c, hipStreamWaitEvent(c->cstream2, go, 0));
            HIP_TRY(c, hipMemcpyAsync(raw + half, text + half, n - half, hipMemcpyHostToDevice, c->cstream2));
            HIP_TRY(c, hipEventRecord(done2, c->cstream2));
            HIP_TRY(c, hipMemcpyAsync(raw, text, half, hipMemcpyHostToDevice, c->cstream));
            HIP_TRY(c, hipStreamWaitEvent(c->cstream, done2, 0));
            (void)hipEventDestroy(go);
            (void)hipEventDestroy(done2);
        } else {
            HIP_TRY(c, hipMemcpyAsync(raw, text, n, hipMemcpyHostToDevice, c->cstream));
        }
        HIP_TRY(c, hipEventRecord(t1, c->cstream));
        c->h2d_events.emplace_back(t0, t1);
    } else
    for (u64 off = 0; off < n; off += 2 * STAGE_BYTES) {
        const u64 len = std::min(2 * STAGE_BYTES, n - off);
        const int p = c->stage_next;
        c->stage_next ^= 1;
        if (c->stage_used[p]) HIP_TRY(c, hipEventSynchronize(c->stage_done[p]));
        staged_memcpy(c->h_stage[p], text + off, len);
        hipEvent_t t0, t1;
        HIP_TRY(c, hipEventCreate(&t0));
        HIP_TRY(c, hipEventCreate(&t1));
        HIP_TRY(c, hipEventRecord(t0, c->cstream));
        HIP_TRY(c, hipMemcpyAsync(raw + off, c->h_stage[p], len, hipMemcpyHostToDevice, c->cstream));
        HIP_TRY(c, hipEventRecord(t1, c->cstream));
        HIP_TRY(c, hipEventRecord(c->stage_done[p], c->cstream));
        c->stage_used[p] = true;
        c->h2d_events.emplace_back(t0, t1);
    }
    if (!defer) {
        // The scan -- the part that can refuse the text -- runs right behind the copy, on the same stream, and is over when this
        // call returns (its first host read-back waits for the copy too: the caller may reuse its buffer).
        rc = scan_text(c, raw, n, format, c->cstream);
        if (rc != KH_OK) (void)hipStreamSynchronize(c->cstream);  // (whatever happened: the caller gets its buffer back)
        if (rc == KH_OK) c->bases_pushed += n;
        return rc;
    }
    // KH_FLAG_DEFER_TEXT_SCAN: while this text travels, the PREVIOUS one is scanned on the scan stream (kernels and host
    // round trips beside the DMA); this one's scan -- and a refusal of it -- is the next call's business (or kh_finish's)
    if (!c->txt_copied[r]) HIP_TRY(c, hipEventCreateWithFlags(&c->txt_copied[r], hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->txt_copied[r], c->cstream));
    rc = scan_unscanned(c);
    (void)hipEventSynchronize(c->txt_copied[r]);  // the caller may reuse its buffer
    if (rc != KH_OK) return rc;  // the previous text was refused (or its scan failed): this one is dropped with it -- the caller starts over
    c->txt_unscanned.on = true;
    c->txt_unscanned.r = r;
    c->txt_unscanned.n = n;
    c->txt_unscanned.format = format;
    c->bases_pushed += n;
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
    uint64_t *dk = nullptr, *dc = nullptr;
    const bool scratch = c->keysA && c->keysB && c->key_cap >= need * sizeof(u64) && c->keyb_cap >= need * sizeof(u64);
    if (scratch) {
        dk = reinterpret_cast<uint64_t *>(c->keysA);
        dc = reinterpret_cast<uint64_t *>(c->keysB);
    } else if (hipMalloc((void **)&dk, need * sizeof(u64)) != hipSuccess || hipMalloc((void **)&dc, need * sizeof(u64)) != hipSuccess) {
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
    if (c->ntab) {  // (a shard table is never kept as the 8-byte image: its 8 bytes per slot are room for the shard's 16-byte table)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->ntab);
        c->ntab = nullptr;
        c->ntab_cap = 0;
    }
    return KH_OK;
}

namespace {
enum { XF_WIDE = 0, XF_PACKED64 = 1, XF_HEADS32 = 2 };  // exchange unit formats (shard.hip.h)


// fmt XF_PACKED64: one u64 per pair into d_keys (d_counts unused); XF_HEADS32: u32 heads into d_keys
// what the export kernels read: the 16-byte table, or its 8-byte image while that holds the counts
kh::SlotSrc slot_src(const kh_ctx *c) {
    kh::SlotSrc s;
    s.table = c->table;
    s.ntab = c->narrow ? c->ntab : nullptr;
    s.geo = c->narrow ? kh::RegionGeom{c->narrow_g.p1_bits, c->narrow_g.b2} : geom_of_cap(c->cap);
    return s;
}

int export_regions(kh_ctx *c, int fmt, uint32_t nparts, void *d_keys, uint64_t *d_counts, uint64_t cap,
                   uint32_t *d_region_counts, uint64_t region_cap, uint64_t *part_counts, uint64_t *table_regions) {
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
    bool counted_by_region_pass = false;
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
                           slot_src(c), (uint32_t)cb, d_region_counts, &c->d_ctr->big);
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
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (uint32_t p = 0; p < nparts; ++p) part_counts[p] = bounds[p + 1] - bounds[p];
    const u64 total = bounds[nparts];
    if (total > cap) return fail(c, KH_ERR_RANGE, "export arrays too small");
    if (total && (!d_keys || (!packed && !d_counts))) return fail(c, KH_ERR_BAD_ARG, "NULL output");
    if (total && fmt == XF_HEADS32) {
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
}  // namespace

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
                               slot_src(c), (uint32_t)cb, d_region_counts, &c->d_ctr->big);
            HIP_TRY(c, hipGetLastError());
            u64 wide = 0;
            rc = read_cursor(c, nullptr, &wide);
            if (rc != KH_OK) return rc;
            if (wide) return fail(c, KH_ERR_RANGE, "a count is too large for 32-bit heads");
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
    return export_regions(c, XF_HEADS32, nparts, d_heads, nullptr, cap, d_region_counts, region_cap, part_counts, table_regions);
}

namespace {
int merge_regions(kh_ctx *c, int fmt, uint32_t nsenders, uint64_t sender_regions, const void *const *d_keys,
                  const uint64_t *const *d_counts, const uint32_t *const *d_region_counts) {
    // a FRESH merge rewrites every region of a lazily reset table; in pieces (kh_set_region_window), the
    // pieces still to come stay unwritten until then (win_open)
    const bool windowed = c && c->win_n > 1;
    int rc = enter(c, true, false, windowed && c->win_open && c->win_open_n == c->win_n && !(c->win_mask & (1ull << c->win_piece)));
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
    if ((rc = need_table(c)) != KH_OK) return rc;  // (uninitialised if new: a fresh merge writes every region -- `dirty` below)
    const kh::TableGeom tg = table_geom(c, c->table, c->cap);
    const u64 nregions = c->cap / kh::REGION_SLOTS;
    if (c->region_cap < nregions) {
        u64 zz = c->bstart ? c->region_cap + 1 : 0;
        if ((rc = ensure_buf(c, &c->bstart, &zz, nregions + 1, "hipMalloc(bstart)")) != KH_OK) return rc;
        zz = c->rfail ? c->region_cap : 0;
        if ((rc = ensure_buf(c, &c->rfail, &zz, nregions, "hipMalloc(rfail)")) != KH_OK) return rc;
        zz = c->rnew ? c->region_cap : 0;
        if ((rc = ensure_buf(c, &c->rnew, &zz, nregions, "hipMalloc(rnew)")) != KH_OK) return rc;
        c->region_cap = nregions;
    }
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
            c->win_dirty = c->table_dirty;
        }
        fresh = c->win_open && !(c->win_mask & (1ull << c->win_piece));  // (enter() closed a window this piece does not fit)
    }
    {
        StageTimer t(c, ST_REGION);
        const dim3 mg((unsigned)nwin), mb(1024);
        const uint8_t *none = nullptr;
        const uint32_t dirty = (uint32_t)(windowed ? (fresh && c->win_dirty) : c->table_dirty);
#define KH_MERGE_LAUNCH(FRESH, FMT) \
    hipLaunchKernelGGL((kh::shard_merge_kernel<FRESH, false, FMT>), mg, mb, 0, c->stream, tg, a, c->rfail, c->rnew, none, kh::RegionGeom{0u, 1u}, c->d_ctr, \
                       FRESH ? dirty : 0u, (uint32_t)region0)
        if (fresh) {
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
                           (const uint8_t *)c->rfail + region0, (const uint32_t *)c->rnew + region0, (u64)nwin, c->d_ctr);
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
                       table_geom(c, c->table, c->cap), a, c->rfail, c->rnew, (const uint8_t *)c->rfail, old_geo, c->d_ctr, 0u, \
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
}  // namespace

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

// =============================================================================================
// the exchange behind the ABI: kh_comm_* / kh_merge_across / kh_group_* (RCCL over xGMI)
// =============================================================================================
#include "exchange.hip.h"
