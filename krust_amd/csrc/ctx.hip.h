// ctx.hip.h -- what the translation units behind include/kmerhip.h share: the context, its constants and knobs, and the
// helpers one unit defines for the others.
//
//   kmerhip.hip   life cycle, table management (lazy allocation, growth, the 8-byte image), results, host memory, pure helpers
//   batch.hip     counting one device-resident range: the direct path and the partitioned batch (level 1 -> level 2 -> regions)
//   input.hip     kh_push* / kh_push_text*: staging, device accumulation, record scanning
//   merge.hip     exports and merges of one context (pairs, dense, region-ordered)
//   exchange.hip  kh_comm_* / kh_merge_across / kh_group_*: the exchange between contexts (RCCL over xGMI, or the local hub)
//   level1_*.hip  the level-1 kernels, one instance per k
//
// Everything here is hidden: the library is built with -fvisibility=hidden and linked with kmerhip.map, so it exports the kh_*
// entry points of include/kmerhip.h only (tests/test_abi.py holds `nm -D` to that list).
#pragma once
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

using kh::Counters;
using kh::Slot;
using kh::u64;

namespace khi __attribute__((visibility("hidden"))) {

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

struct Comm;  // exchange.hip: RCCL communicator (or the process-local hub) of this rank

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
    bool l2_narrow = true;           // KMERHIP_L2_NARROW=0: level 2 never narrows 8-byte payloads to the 4 bytes below the region index
    bool l2_no_room_wide = false;    // KMERHIP_L2_NO_ROOM_WIDE=1: as if a batch sized for a narrowing level 2 had no room for 8-byte output (drives KH_RETRY_WIDE)
    u64 hot_cut = 0;                 // KMERHIP_HOT_CUT (0: default; ~0: no bucket is hot)
    int ovf_agg = -1;                // KMERHIP_OVF_AGG
    double survival = 0;             // KMERHIP_SURVIVAL
    bool heads_always = false;       // KMERHIP_HEADS_ALWAYS=1: every fresh pass leaves the exchange-head counts behind, communicator or not
    u64 table_room_mb = 0;           // KMERHIP_TABLE_ROOM_MB: what the sample-sized table may take, as if the device had no more
    bool stop_after_p1 = false, stop_after_p2 = false;  // ablation builds (KH_ABL*)
};
inline const char *env_of(const char *name) {
    const char *e = getenv(name);
    return (e && *e) ? e : nullptr;
}
void read_knobs(Knobs &k);

}  // namespace khi

struct kh_ctx {
    khi::Knobs knobs;
    int device = 0;
    khi::Comm *comm = nullptr;
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
    uint8_t *h_stage[2] = {nullptr, nullptr};  // pinned: bases then qual, each stage_bytes
    u64 stage_bytes = 0;                       // 0 until pageable memory is pushed / fetched; then 1, 8 or 64 MiB (input.hip ensure_stage)
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
    u64 prev_part_budget = 0;  // ... as it was when release_part_buffers gave them back
    uint8_t *keysA = nullptr, *keysB = nullptr;  // partition ping-pong buffers
    u64 key_cap = 0, keyb_cap = 0;  // bytes of keysA / keysB
    // Round 5: between the end of a count and the next kh_reset the two partition buffers are idle -- up to 0.8 of the device --
    // while a merge needs tens of GB of scratch AND the shard's 16-byte table.  Round 4 gave the buffers back to the driver
    // (release_part_buffers) and took them again at the next count: 370 ms of hipFree / hipMalloc per count-and-merge step
    // at configs[3]'s size (bench.py --force-merge).  Now the merge BORROWS from them: a bump allocator over keysA / keysB
    // (kmerhip.hip borrow()), for the exchange's send / receive buffers and for the table itself (table_borrowed).  The loan
    // ends with kh_reset, or when anything is about to write the partition buffers (end_borrow: a borrowed table moves out).
    bool exports_seen = false;       // a heads export was asked of this context: its fresh passes leave the head counts behind (batch.hip want_heads)
    bool borrow_on = false;
    u64 borrow_off[2] = {0, 0};     // bytes lent out of keysA / keysB
    bool table_borrowed = false;    // `table` points into keysA / keysB: never hipFree'd
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
    uint8_t *rtouch = nullptr;       // [regions] all zero between uses: the regions an overflow list touched (merge.hip recount_touched_heads)
    u64 *radd = nullptr;             // [regions] shard_merge_kernel: sum of the counts it put into every target region (conservation)
    u64 *rdig = nullptr;             // shard_merge_narrow_kernel: [target][sender][3] arrival-digest partials; region_compact_*: [region][2]
    u64 rdig_cap = 0;
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
    u64 region_cap = 0;              // entries of EVERY [regions] array above: they grow together, ensure_region_scratch()
    u64 *scan_partial = nullptr;
    u64 scan_cap = 0;
    u64 *est_set = nullptr;          // scratch of distinct_sample_kernel (partition.hip.h): the set a few level-1 partitions are counted in
    u64 est_set_cap = 0;
    u64 est_keys = 0;                // distinct keys the current fresh range is expected to bring (from that sample; 0 = no estimate)
    bool sized_by_sample = false;    // the table's size comes from such a sample (stats / trace)
    bool narrow2_refused = false;    // a batch sized for a narrowing level 2 (15 B per window) could not narrow and had no room for 8-byte output: later ranges are sized for that
    bool estimate_on = true;         // KMERHIP_ESTIMATE=0: never (rounds 1-3's sizing: the hint, or the worst case)
    u64 part_batches = 0;
    double stage_ms[khi::ST_N] = {0};
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
    uint8_t *txt_st = nullptr;    u64 txt_st_cap = 0;    // FASTA: line state behind each 1 KiB unit (rawparse.hip.h)
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

namespace khi __attribute__((visibility("hidden"))) {

inline int fail(kh_ctx *c, int code, const char *what, hipError_t e = hipSuccess) {
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

inline int grid_for(u64 items) {
    u64 b = (items + kh::BLOCK - 1) / kh::BLOCK;
    if (b < 1) b = 1;
    if (b > (u64)GRID_CAP) b = GRID_CAP;
    return (int)b;
}

// ---- kmerhip.hip -----------------------------------------------------------------------------------------------------------
// Every entry point starts with enter(): host pushes are accumulated on the device and counted lazily; anything that looks at
// the table first counts what is pending.
int enter(kh_ctx *c, bool flush_pending = true, bool need_table = true, bool keep_window = false, bool narrow_ok = false);
int need_table(kh_ctx *c);
void resize_empty_table(kh_ctx *c, u64 newcap);
int clear_if_dirty(kh_ctx *c);
int close_fresh_window(kh_ctx *c);
int ensure_wide(kh_ctx *c);
int sync_counters(kh_ctx *c);
u64 round_cap(double want);
kh::RegionGeom geom_of_cap(u64 cap);
bool cap_is_pow2(u64 cap);
kh::TableGeom table_geom(const kh_ctx *c, Slot *table, u64 cap);
int grow_to(kh_ctx *c, u64 newcap);
int ensure_room(kh_ctx *c, u64 bound, bool allow_shrink_hint, bool *want_smaller);
int drain_events(kh_ctx *c);
int ensure_region_scratch(kh_ctx *c, u64 nregions);  // every [regions] scratch array of the context, grown together
int release_part_buffers(kh_ctx *c);  // the two partition buffers back to the device (they come back with the next partitioned batch)
void *borrow(kh_ctx *c, u64 bytes);   // `bytes` of the idle partition buffers (nullptr: not lent out, or no room); never freed
u64 borrow_room(const kh_ctx *c);     // the largest single request borrow() could serve now
int end_borrow(kh_ctx *c);            // the partition buffers are about to be written (or freed): a borrowed table moves to memory of its own
void drop_table(kh_ctx *c);           // the 16-byte table is no longer needed: freed, unless it was borrowed
int device_scan(kh_ctx *c, const uint32_t *in, u64 n, u64 *out);
int zero_cursors(kh_ctx *c);
int read_cursor(kh_ctx *c, u64 *cursor, u64 *big);
int head_count_bits(const kh_ctx *c, u64 regions);
extern bool g_pow2_tables;
extern int g_copy_threads;
// ---- batch.hip
int count_device_range(kh_ctx *c, const uint8_t *d_bases, const uint8_t *d_qual, u64 n, u64 wlo_off);
// ---- input.hip
int flush_acc(kh_ctx *c, bool carry);
int flush_text(kh_ctx *c);
int scan_unscanned(kh_ctx *c);
bool is_pinned_host(const void *p);
int d2h_staged(kh_ctx *c, void *dst, const void *d_src, u64 bytes);
// ---- merge.hip
enum { XF_WIDE = 0, XF_PACKED64 = 1, XF_HEADS32 = 2 };  // exchange unit formats (shard.hip.h)
int mark_touched_regions(kh_ctx *c, const void *ovf_list, const u64 *d_ovf, u64 ovf_lim);  // (batch.hip: before the overflow list's insert ...
int recount_touched_heads(kh_ctx *c, u64 nregions);                                           //  ... and after it)
// d_digest (optional; 3 x nparts u64 on the device): where the compaction can digest what it writes on the way (packed pairs / heads out
// of the 8-byte image), the sender's digests of exchange.hip are left there and *digest_done is set; otherwise the caller digests.
int export_regions(kh_ctx *c, int fmt, uint32_t nparts, void *d_keys, uint64_t *d_counts, uint64_t cap, uint32_t *d_region_counts,
                   uint64_t region_cap, uint64_t *part_counts, uint64_t *table_regions, u64 *d_digest = nullptr, bool *digest_done = nullptr);
// d_digest (optional; 3 x nsenders u64 on the device): where the merge kernel can digest what it reads on the way (the shard built as
// the 8-byte image), the arrival digests of exchange.hip are written there and *digest_done is set; otherwise the caller digests.
int merge_regions(kh_ctx *c, int fmt, uint32_t nsenders, uint64_t sender_regions, const void *const *d_keys,
                  const uint64_t *const *d_counts, const uint32_t *const *d_region_counts, u64 *d_digest = nullptr, bool *digest_done = nullptr);
// ---- exchange.hip
void comm_release(kh_ctx *c);

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

// ---- device scratch management for the partitioned path --------------------------------------
inline double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
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

GeomChoice make_geom(const kh_ctx *c, u64 cap);

}  // namespace khi
