/*
 * kmerhip.h -- C ABI of the MI355X-native canonical k-mer counter.
 *
 * This is the drop-in boundary for krust's counting hot path.  The reference
 * (Rust crate `kmerust` 0.3.1) has no FFI of its own; the seam these entry
 * points replace is the internal call
 *
 *     KmerMap::new().build(seqs, k)                     src/run.rs:494-503
 *     KmerMap::new().build_with_quality(seqs, k, minq)  src/run.rs:505-520
 *     KmerMap::into_hashmap(k)                          src/run.rs:573-582
 *
 * and its public packed twin
 *
 *     count_kmers_from_sequences(iter<Bytes>, KmerLength) -> HashMap<u64,u64>
 *                                                       src/streaming.rs:198-204
 *
 * INTEGRATION.md shows the Rust `extern "C"` binding a krust maintainer would
 * add at those call sites.  Plain pointers and sizes only; no C++ or torch
 * types cross this boundary; no exception crosses it.
 *
 * Semantics (bit-exact with the reference):
 *   - k-mer = k consecutive bytes all in ACGTacgt (src/kmer.rs:266-286), 2-bit
 *     packed A=0 C=1 G=2 T=3, first base most significant (kmer.rs:21-32,467-471)
 *   - canonical = lexicographic min of k-mer and reverse complement
 *     (kmer.rs:348-390) == integer min of the packed forms
 *   - optional quality mask: a window is skipped iff any of its k quality bytes
 *     is < min_quality.saturating_add(33)          (run.rs:538,543-548)
 *   - k-mers never span records; in a flat buffer records are separated by at
 *     least one byte outside ACGTacgt (e.g. '\n'), which no window may contain
 *   - count[key] += 1 per window, u64 counts          (run.rs:565-571)
 *
 * Threading: one producer thread per context (not re-entrant); independent
 * contexts may be used concurrently.  All functions return KH_OK (0) or a
 * negative kh_status; k-mers are never dropped silently.
 */
#ifndef KMERHIP_H
#define KMERHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: these declarations -- and nothing else -- are what it exports. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define KMERHIP_ABI_VERSION 2

typedef enum kh_status {
    KH_OK = 0,
    KH_ERR_BAD_K = -1,      /* KmerLengthError, src/kmer.rs:100-110, src/error.rs:86-95 */
    KH_ERR_BAD_ARG = -2,
    KH_ERR_NO_DEVICE = -3,  /* no HIP device / HIP runtime unusable */
    KH_ERR_OOM = -4,        /* device or pinned-host allocation failed */
    KH_ERR_TABLE_FULL = -5, /* table cannot grow further */
    KH_ERR_HIP = -6,        /* a HIP call failed; kh_last_error() has the text */
    KH_ERR_STATE = -7,      /* call not valid in the context's current state */
    KH_ERR_RANGE = -8,      /* caller-provided output array too small */
    KH_ERR_FORMAT = -9,     /* kh_push_text*: a record layout the device scanner does not take
                               (nothing was counted; parse on the host and use kh_push) */
    KH_ERR_RCCL = -10,      /* an RCCL call failed or timed out (kh_comm_init / kh_merge_across); kh_last_error() has the text */
    KH_ERR_PEER = -11       /* kh_merge_across: ANOTHER rank of the collective failed (its own status says why;
                               kh_last_error() names the rank); every rank leaves the merge together */
} kh_status;

typedef struct kh_ctx kh_ctx;

/* Counter configuration.  Mirrors what the reference passes across its seam:
 * k (run.rs:500), min_quality: Option<u8> (run.rs:509). */
typedef struct kh_config {
    uint32_t struct_size;    /* sizeof(kh_config), for ABI evolution */
    uint32_t k;              /* 1..32 */
    int32_t  min_quality;    /* -1 = None; 0..255 = Some(q) (Phred, +33 applied inside) */
    int32_t  device;         /* HIP device ordinal; -1 = current device */
    uint64_t capacity_hint;  /* expected number of DISTINCT canonical k-mers; 0 = grow on demand */
    void    *stream;         /* hipStream_t to launch on.  NULL = the context creates its own NON-BLOCKING stream
                                (not ordered with the legacy default stream: synchronise buffers handed to
                                kh_push_device yourself) -- unless KH_FLAG_CALLER_STREAM says that NULL means
                                the legacy default stream itself (what torch.cuda.current_stream() usually is) */
    uint32_t flags;          /* KH_FLAG_* */
    uint32_t input_mib;      /* expected total input of this context in MiB (the size of the file about to be pushed), 0 = unknown.
                                Never needed: it lets kh_push_text size its device-side accumulation for THIS input instead
                                of for an eighth of the free memory (a small file then costs a small allocation) */
} kh_config;

#define KH_FLAG_TRACE 1u           /* print phase timings to stderr (also env KMERHIP_TRACE=1) */
#define KH_FLAG_FORCE_DIRECT 2u    /* always use the direct (device-scope atomic) insert path */
#define KH_FLAG_FORCE_PARTITION 4u /* always use the partitioned (LDS region rebuild) path;
                                      default: chosen per push from batch and table size
                                      (env KMERHIP_PATH=direct|partition overrides) */
#define KH_FLAG_CALLER_STREAM 8u   /* launch on cfg->stream even when it is NULL (= the legacy default stream) */
#define KH_FLAG_DEFER_TEXT_SCAN 16u /* kh_push_text returns when its text is on the device; the record scan -- and with it a
                                      KH_ERR_FORMAT refusal -- happens during the NEXT call that enters the context (the next
                                      kh_push_text scans it beside its own transfer; kh_finish at the latest).  A refusal then
                                      concerns the PREVIOUS text (the current one is dropped with it): for callers that restart
                                      the whole input on KH_ERR_FORMAT, as the kmerust command line does */

/* indices into kh_stats.stage_ms */
#define KH_NUM_STAGES 8
#define KH_STAGE_DIRECT 0      /* count_direct_kernel (the device-atomic path) */
#define KH_STAGE_P1_COUNT 1    /* (no longer used: level 1 is a single pass; always 0) */
#define KH_STAGE_P1_SCATTER 2  /* level 1: part1_bins_kernel (4-byte payloads, k <= 21 at >= 1024 regions) /
                                  part1_bins64_kernel (8-byte payloads) */
#define KH_STAGE_P2_COUNT 3    /* level 2, exact path only: part2_count_kernel (0 = the batch took the arena path) */
#define KH_STAGE_P2_SCATTER 4  /* level 2: part2_arena_kernel, or part2_scatter_lines_kernel (+ part2_scatter_kernel) on the exact path */
#define KH_STAGE_REGION 5      /* region_count_kernel32 / region_count_kernel64 (+ shard_merge_kernel in a merge) */
#define KH_STAGE_MISC 6        /* scans, plans, chunk lists, bounds, memsets, region_reduce, ovf_insert */
#define KH_STAGE_GROW 7        /* table growth / rehash */

typedef struct kh_stats {
    uint64_t bases;          /* bytes pushed (incl. separators) */
    uint64_t kmers;          /* valid windows counted == sum of all counts */
    uint64_t distinct;       /* live table entries */
    uint64_t table_slots;    /* current table capacity (16-byte slots) */
    uint64_t grows;          /* number of table rehashes */
    uint64_t launches;       /* count-kernel launches */
    double   count_kernel_ms;/* sum of all counting-kernel durations (HIP events on the launch stream) */
    double   h2d_ms;         /* host->device staging time for kh_push */
    uint64_t part_batches;   /* batches that went through the partitioned path */
    double   stage_ms[KH_NUM_STAGES]; /* per-stage kernel time, see KH_STAGE_* */
    double   text_scan_ms;   /* kh_push_text*: record-scanning kernels (not part of count_kernel_ms) */
    uint64_t slot_bytes;     /* ABI 2: bytes per slot of the form the counts live in right now -- 16: {u64 key, u64 count};
                              * 8: the image a partitioned count, or a fresh merge of packed / heads units, leaves (count << 32 |
                              * 32 hash bits); what reads results takes either, everything else converts first */
} kh_stats;

/* ---- lifecycle ---------------------------------------------------------- */
int  kh_create(kh_ctx **out, const kh_config *cfg);
void kh_destroy(kh_ctx *ctx);
/* Forget all counts, keep the table allocation (KmerMap::new() again). */
int  kh_reset(kh_ctx *ctx);

/* ---- input: replaces KmerMap::build / build_with_quality ---------------- */
/* Host buffers.  `bases` is a flat byte buffer of n bytes holding any number
 * of records separated by >=1 non-ACGTacgt byte; `qual` is NULL or a parallel
 * buffer of n bytes (same offsets).  With qual==NULL or min_quality==-1 no
 * quality masking happens (run.rs:543: both must be Some).  A quality byte of
 * 0xFF can never mask (the threshold is min_quality.saturating_add(33) <= 255
 * and a base is masked iff its byte is BELOW it): it is the filler for records
 * without qualities (FASTA) pushed beside FASTQ ones in one flat buffer --
 * any smaller filler ('~' = 126) would drop those records' windows once
 * min_quality >= 94.  Returns after the buffers have been consumed (caller may
 * reuse them). */
int kh_push(kh_ctx *ctx, const uint8_t *bases, const uint8_t *qual, uint64_t n);
/* Same, for buffers already resident in this device's HBM (no copy). */
int kh_push_device(kh_ctx *ctx, const uint8_t *d_bases, const uint8_t *d_qual, uint64_t n);
/* Raw FASTA / FASTQ TEXT, scanned on the device: replaces the record readers in front of the
 * path (SeqReader, src/reader.rs:58-79; the needletail / rust-bio loops of src/streaming.rs:858-893)
 * for uncompressed input.  `text` holds WHOLE records: it starts at a record header and ends at a
 * record end (a missing final newline is fine).  FASTA: wrapped records are joined, k-mers never
 * span records.  FASTQ: 4-line records only; qualities are used iff the context has a min_quality.
 * Returns KH_ERR_FORMAT -- with nothing counted -- for anything else (wrapped FASTQ, blank lines
 * between FASTQ records, a missing '@' / '+' / '>' marker, |seq| != |qual|, blanks at line ends of
 * a FASTA record): the caller then parses that text itself and uses kh_push.
 * kh_push_text returns when `text` has been copied and scanned (the caller may reuse it; a refusal is always about
 * THIS call's text -- unless the context was created with KH_FLAG_DEFER_TEXT_SCAN).  The scanned, flat form of the texts
 * ACCUMULATES on the device (up to an eighth of the free memory, or kh_config.input_mib) and is counted when that is
 * full or when anything looks at the table (kh_finish, kh_result_*, kh_lookup, kh_push_device ...): a file streamed in
 * chunks is counted in one or a few large batches.  So an error of the counting itself (KH_ERR_OOM, KH_ERR_TABLE_FULL:
 * errors that poison the context) can surface in a later call, at the latest in kh_finish.
 * kh_push_text_device: d_text must be 16-byte aligned. */
#define KH_TEXT_FASTA 1
#define KH_TEXT_FASTQ 2
int kh_push_text(kh_ctx *ctx, const uint8_t *text, uint64_t n, int format);
int kh_push_text_device(kh_ctx *ctx, const uint8_t *d_text, uint64_t n, int format);
/* Wait for all pushed work; fill stats (may be NULL). */
int kh_finish(kh_ctx *ctx, kh_stats *stats);

/* ---- host memory the device reaches directly ---------------------------- */
/* kh_push / kh_push_text stage PAGEABLE caller memory through pinned chunks (a multi-threaded memcpy: ~20-30 GB/s, below
 * PCIe's ~57) and kh_result_copy bounces the other way (plus the first-touch page faults of a fresh array).  Buffers
 * from kh_host_alloc (pinned, any device of the process), or the caller's own memory after kh_host_register, skip both:
 * the copy engine reads / writes them itself.  Detected per call (hipPointerGetAttributes): nothing else changes, and
 * kh_push still returns only when the buffers may be reused.  The reference reads a whole file into Vec<Bytes> before
 * counting (src/reader.rs:58-79); a Rust host would read() into such a buffer instead. */
int kh_host_alloc(void **out, uint64_t bytes);
int kh_host_free(void *p);
int kh_host_register(void *p, uint64_t bytes);
int kh_host_unregister(void *p);

/* ---- output: replaces KmerMap::into_hashmap ----------------------------- */
/* Number of distinct canonical k-mers with count >= min_count
 * (min_count filter of output_counts, run.rs:447-450). */
int kh_result_size(kh_ctx *ctx, uint64_t min_count, uint64_t *n);
/* Copy (packed canonical key, count) pairs with count >= min_count into
 * caller arrays of capacity cap; order unspecified (HashMap iteration order is
 * unspecified in the reference too).  *n receives the number written. */
int kh_result_copy(kh_ctx *ctx, uint64_t *keys, uint64_t *counts, uint64_t cap,
                   uint64_t min_count, uint64_t *n);
/* Same into device arrays (for the multi-GPU exchange and device consumers). */
int kh_result_copy_device(kh_ctx *ctx, uint64_t *d_keys, uint64_t *d_counts, uint64_t cap,
                          uint64_t min_count, uint64_t *n);
/* Count-of-counts after the min_count filter, ascending by count
 * (compute_histogram, src/histogram.rs:88-94 as used by run.rs:471-481). */
int kh_histogram(kh_ctx *ctx, uint64_t min_count, uint64_t *count, uint64_t *freq,
                 uint64_t cap, uint64_t *n);
/* counts[i] = count of packed canonical key keys[i], 0 if absent
 * (`kmerust query`, src/main.rs:264-280). */
int kh_lookup(kh_ctx *ctx, const uint64_t *keys, uint64_t n, uint64_t *counts);

/* ---- multi-GPU merge (no reference counterpart) ----------- */
/* Owner shard of a packed canonical k-mer among nparts shards: a fast-range of the top bits of
 * the table hash, so a shard is a contiguous range of table regions (same function on host and
 * device). */
uint32_t kh_owner(uint64_t key, uint32_t k, uint32_t nparts);

/* Fast path for a power-of-two number of ranks whose tables have the same size:
 *   sender:   kh_export_regions_device -- live pairs in REGION order (hence grouped by owner) plus the
 *             live count of every region; part_counts[p] (host) = pairs owned by shard p.
 *   exchange: all-to-all of the pair segments and of each owner's slice of the region counts.
 *   receiver: kh_reset, kh_set_shard(rank, nranks), kh_merge_regions_device -- the shard table is
 *             rebuilt region by region in LDS from the senders' segments (no global atomics).
 * *table_regions receives the sender's region count (must be equal on all ranks).
 * A shard table holds only keys of its hash range: it accepts kh_merge_* but not kh_push*
 * (KH_ERR_STATE); kh_reset turns it back into a full table. */
int kh_set_shard(kh_ctx *ctx, uint32_t index, uint32_t count);
int kh_export_regions_device(kh_ctx *ctx, uint32_t nparts, uint64_t *d_keys, uint64_t *d_counts, uint64_t cap,
                             uint32_t *d_region_counts, uint64_t region_cap, uint64_t *part_counts,
                             uint64_t *table_regions);
/* d_keys[s], d_counts[s]: sender s's pairs for THIS shard (device pointers, host arrays of nsenders
 * entries); d_region_counts[s]: sender s's live counts of the sender_regions / shard_count regions of
 * this shard's range. */
int kh_merge_regions_device(kh_ctx *ctx, uint32_t nsenders, uint64_t sender_regions,
                            const uint64_t *const *d_keys, const uint64_t *const *d_counts,
                            const uint32_t *const *d_region_counts);

/* Packed form of the same two calls -- half the bytes on the links: ONE uint64 per pair,
 * count << 32 | the 32 bits of the table hash below the region index (the receiver knows every
 * segment's region, and the hash is a bijection, so the key comes back exactly).  Representable iff
 * 2k - log2(table_regions) <= 32 and every count < 2^32; otherwise the export returns KH_ERR_RANGE
 * and the caller uses the unpacked pair of calls (all ranks must take the same route). */
int kh_export_regions_packed_device(kh_ctx *ctx, uint32_t nparts, uint64_t *d_pairs, uint64_t cap,
                                    uint32_t *d_region_counts, uint64_t region_cap, uint64_t *part_counts,
                                    uint64_t *table_regions);
int kh_merge_regions_packed_device(kh_ctx *ctx, uint32_t nsenders, uint64_t sender_regions,
                                   const uint64_t *const *d_pairs, const uint32_t *const *d_region_counts);
/* Narrowest form -- a quarter of the bytes: 32-bit "heads" = [2k - log2(table_regions) hash bits |
 * addend - 1 in the remaining cb bits].  A pair whose count exceeds 2^cb travels as several heads of
 * the same key (the receiver adds them up); part_counts / region counts are in heads.  Applies iff
 * 1 <= 2k - log2(table_regions) <= 28 and no count exceeds 64 * 2^cb, else KH_ERR_RANGE.  `cap` is
 * in heads (allow ~2x the distinct count). */
int kh_export_regions_heads_device(kh_ctx *ctx, uint32_t nparts, uint32_t *d_heads, uint64_t cap,
                                   uint32_t *d_region_counts, uint64_t region_cap, uint64_t *part_counts,
                                   uint64_t *table_regions);
int kh_merge_regions_heads_device(kh_ctx *ctx, uint32_t nsenders, uint64_t sender_regions,
                                  const uint32_t *const *d_heads, const uint32_t *const *d_region_counts);

/* Exchange in pieces, so that export, all-to-all and merge of successive pieces overlap: after
 * kh_set_region_window(ctx, piece, npieces) (npieces a power of two <= 64) the three export calls above
 * cover only piece `piece` of every owner's region range -- the counts of the other regions come back as
 * zero and nothing of them is written, so the caller sees an ordinary export of a table that is empty
 * elsewhere -- and the three merge calls rebuild only the matching share of the shard's regions (the
 * senders' region-count arrays keep their full length, zero outside the piece).  Pieces may come in any
 * order; a merge into an empty shard table stays a "fresh" rebuild for every piece.  Anything else that
 * touches the table in between is allowed (the pieces still missing then count as empty).
 * (0, 1) restores whole-range calls.  KH_ERR_BAD_ARG if a range has fewer regions than pieces. */
int kh_set_region_window(kh_ctx *ctx, uint32_t piece, uint32_t npieces);
/* Exchange units held by every region of the whole table (ignores the window): heads (unit_bytes 4),
 * packed pairs (8) or pairs (16) -- the first phase of the matching export on its own, so that a
 * pipelined exchange can announce the sizes of all its pieces before the first one is compacted.
 * Blocks until d_region_counts is complete.  KH_ERR_RANGE as the matching export. */
int kh_region_unit_counts_device(kh_ctx *ctx, uint32_t unit_bytes, uint32_t *d_region_counts, uint64_t region_cap,
                                 uint64_t *table_regions);

/* Small k (2k <= 26): the whole key space is a dense array of 4^k counts, which -- unlike a hash
 * table -- IS element-wise reducible: ranks merge with one all-reduce(sum) (any number of ranks,
 * tables of any size).  kh_export_dense_device: d_dense[key] = count, 0 for absent keys.
 * kh_merge_dense_device: count[key] += d_dense[key] for the keys shard `owner` of `nparts` owns
 * (kh_owner); the table keeps its full geometry.  KH_ERR_RANGE for larger k. */
int kh_export_dense_device(kh_ctx *ctx, uint64_t *d_dense, uint64_t n_entries);
int kh_merge_dense_device(kh_ctx *ctx, const uint64_t *d_dense, uint64_t n_entries, uint32_t owner, uint32_t nparts);

/* Generic path (any number of shards, tables of any size): pairs grouped by owner, then
 * kh_merge_pairs_device re-inserts them with device atomics. */
/* kh_export_by_owner_device: compact all live (key,count) pairs grouped by owner shard into device arrays
 * of capacity cap; part_counts[p] (host array, nparts entries) receives the
 * number of pairs for shard p; pairs of shard p start at sum(part_counts[0..p)). */
int kh_export_by_owner_device(kh_ctx *ctx, uint32_t nparts, uint64_t *d_keys,
                              uint64_t *d_counts, uint64_t cap, uint64_t *part_counts);
/* count[key] += counts[i] for n device-resident / host-resident pairs. */
int kh_merge_pairs_device(kh_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_counts, uint64_t n);
int kh_merge_pairs(kh_ctx *ctx, const uint64_t *keys, const uint64_t *counts, uint64_t n);

/* ---- the exchange itself, behind this ABI (RCCL over xGMI) --------------- */
/* north_star: "reads shard naturally per GPU ... with a final RCCL reduce of per-GPU hash tables over xGMI".
 * The reference's only parallelism is rayon over records (src/run.rs:500-503); these calls are what stands
 * in for it at N > 1 GPUs, so that a host (Rust, C++) needs nothing but this library:
 *
 *   rank 0:        kh_comm_unique_id(&id);   ... the host hands the 128 bytes to every rank (any means) ...
 *   every rank r:  kh_create(&ctx, &cfg) on its own device;   kh_comm_init(ctx, nranks, r, &id);
 *                  kh_push*(ctx, its share of the reads) ...;     kh_merge_across(ctx, &info);
 *                  kh_result_* / kh_histogram / kh_lookup         -> the keys with kh_owner(key, k, nranks) == r
 *
 * kh_merge_across is a COLLECTIVE: every rank of the communicator calls it, once per merge.  It picks, by a
 * vote among the ranks, the narrowest exchange unit every table can represent (32-bit heads, packed u64,
 * 16-byte pairs; dense counts + one all-reduce for 2k <= 26), sends every owner its region segments with
 * ncclSend / ncclRecv groups on an own stream -- in KMERHIP_MERGE_PIECES (default 4) pieces, so that the
 * export of piece i + 1 and the LDS merge of piece i - 1 overlap the transfer of piece i -- and leaves this
 * context holding its hash-range shard (kh_set_shard state; kh_reset makes it a full table again).
 * Ranks need not arrive with tables of one size (each is sized from its own input): the merge starts with a gather
 * of the sizes, and a rank whose table is smaller than the largest re-lays it out to that size first.
 * One rank per context; ranks may be processes (one per GPU) or threads of one process (kh_group_*).
 *
 * Failure is collective too.  Every small all-gather of the sequence carries each rank's status, no transfer starts
 * before such a gather has come back clean, and one more closes the merge: a rank that fails on its own (out of
 * memory for a receive buffer, a kernel error) returns its status, EVERY other rank returns KH_ERR_PEER from the same
 * call -- nobody is left waiting.  That includes a rank whose context is unusable when it arrives (poisoned by an earlier
 * error, or the counting its last kh_push left pending fails now): it still joins the first gather and reports there.
 * Refused BEFORE the collective starts, on the calling rank alone: a dead communicator, and a table that is already a
 * shard (a finished merge leaves every rank one, so that refusal is collective by itself).  Every wait is bounded by KMERHIP_MERGE_TIMEOUT_S (default 300 s): on expiry, or on
 * an asynchronous RCCL error, the communicator is aborted (ncclCommAbort) and the call returns KH_ERR_RCCL; that
 * context's communicator is dead afterwards (kh_merge_across refuses it; make a new context and communicator).
 * After any failed merge the table's content is unspecified until kh_reset. */
typedef struct kh_unique_id { char internal[128]; } kh_unique_id;  /* ncclUniqueId, opaque */
int kh_comm_unique_id(kh_unique_id *out);
/* Collective over all ranks (it blocks until every rank has called it).  The context must be on the GPU this
 * rank uses; a context has at most one communicator (KH_ERR_STATE otherwise); kh_destroy releases it. */
int kh_comm_init(kh_ctx *ctx, uint32_t nranks, uint32_t rank, const kh_unique_id *id);

#define KH_ROUTE_NONE 0           /* single rank: nothing to exchange */
#define KH_ROUTE_DENSE 1          /* 2k <= 26: dense 4^k counts, one all-reduce(sum) */
#define KH_ROUTE_REGIONS_HEADS 2  /* region-ordered 32-bit heads */
#define KH_ROUTE_REGIONS_PACKED 3 /* region-ordered packed u64 */
#define KH_ROUTE_REGIONS_WIDE 4   /* region-ordered (u64 key, u64 count) */
#define KH_ROUTE_PAIRS 5          /* owner-grouped pairs, device-atomic re-insert (any world size / table sizes) */
typedef struct kh_merge_info {
    uint32_t route;           /* KH_ROUTE_* */
    uint32_t pieces;          /* pipeline pieces used (1 = one shot) */
    uint32_t unit_bytes;      /* bytes per exchanged unit */
    uint32_t nranks;
    uint64_t local_distinct;  /* entries of this rank's table before the merge */
    uint64_t sent_units;      /* units that left this rank (its own share excluded) */
    uint64_t recv_units;      /* units this rank merged (its own share included) */
    uint64_t owned_distinct;  /* entries of this rank's shard after the merge */
    double   export_ms;       /* host wall time in the export calls */
    double   wait_ms;         /* host wall time waiting for transfers / small collectives */
    double   merge_ms;        /* host wall time in the merge calls */
    double   total_ms;
    /* conservation (round 5): the merge checks itself.  Every sender digests what it exports per destination (units, sum of
     * the counts they carry, a wrapping checksum of the unit words), the digests travel with the small gathers, every receiver
     * digests what ARRIVED per source and the merge kernels add up the counts they put into the shard: any difference fails
     * the merge on every rank (KH_ERR_RCCL on the rank that saw it, KH_ERR_PEER on the others) -- a transport that loses
     * half a message, or a merge kernel that drops a unit, cannot produce a quietly smaller table. */
    uint32_t nranks_seen;      /* ranks the TRANSPORT reports (ncclCommCount; the process-local hub's size): == nranks */
    uint32_t conserved;        /* 1: every check above held on this rank (always 1 when the call returned KH_OK) */
    uint64_t sent_count_sum;   /* sum of the counts of all units this rank exported (its own share included) */
    uint64_t merged_count_sum; /* sum of the counts this rank's merge put into its shard == occurrences it now holds */
} kh_merge_info;
int kh_merge_across(kh_ctx *ctx, kh_merge_info *info /* may be NULL */);

/* Single-process form: one context per device and one host thread per context inside the library
 * ("internal HIP streams / one host thread per GPU").  devices[i] = HIP ordinal of rank i.
 * Distinct devices talk RCCL (xGMI); a device listed more than once (tests on a 1-GPU box; RCCL refuses
 * duplicate devices) makes the whole group exchange by device-to-device copies inside the process instead.
 * cfg->device and cfg->stream are ignored (every context owns its stream).  The contexts belong to the
 * group: use them with every kh_* call, but destroy them only through kh_group_destroy. */
typedef struct kh_group kh_group;
int kh_group_create(kh_group **out, const kh_config *cfg, const int32_t *devices, uint32_t ndevices);
kh_ctx *kh_group_ctx(kh_group *g, uint32_t rank);
uint32_t kh_group_size(const kh_group *g);
/* kh_merge_across on every context, each on its own host thread; infos: ndevices entries or NULL.
 * Returns KH_OK, else the status of a rank that failed itself (before any peer's KH_ERR_PEER). */
int kh_group_merge(kh_group *g, kh_merge_info *infos);
void kh_group_destroy(kh_group *g);

/* ---- pure helpers (host, no device) ------------------------------------- */
/* pack_bytes / Kmer::pack, src/kmer.rs:304-312,467-471.  KH_ERR_BAD_ARG if a
 * byte is outside ACGTacgt (*err_pos, if given, = its position: kmer.rs:277-280). */
int kh_pack(const uint8_t *bases, uint32_t k, uint64_t *packed, uint32_t *err_pos);
/* unpack_to_bytes, src/kmer.rs:431-440; writes k bytes (no terminator). */
int kh_unpack(uint64_t packed, uint32_t k, uint8_t *out);
/* canonical form of a packed k-mer, src/kmer.rs:348-390; *is_rc optional. */
int kh_canonical(uint64_t packed, uint32_t k, uint64_t *canonical, int *is_rc);

/* ---- diagnostics -------------------------------------------------------- */
const char *kh_strerror(int status);
/* Text of the last failure on this context (HIP error string etc.). */
const char *kh_last_error(const kh_ctx *ctx);
int kh_abi_version(void);

/* ---- deterministic synthetic reads (bench / tests) -------- */
/* Writes reads [first_read, first_read+n_reads) as stride-(read_len+1) records
 * (read_len bases then '\n') into device buffers of n_reads*(read_len+1) bytes.
 * d_qual may be NULL.  Launches on `stream` (hipStream_t or NULL) of `device`. */
int kh_synth_reads_device(int device, void *stream, uint64_t seed, uint64_t genome_len,
                          uint32_t read_len, uint64_t first_read, uint64_t n_reads,
                          uint8_t *d_bases, uint8_t *d_qual);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* KMERHIP_H */
