/*
 * kmer_oracle.h -- CPU restatement of krust's canonical k-mer counting path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load or
 * call it, and there only as the checker / reported baseline.  The product path
 * (krust_amd/, include/kmerhip.h) never links or imports this.
 *
 * Parity pinning: the reference is Rust (crate kmerust 0.3.1) and cannot be
 * compiled in this image (no rustc/cargo), so there is no oracle/_ref build.
 * The restatement is pinned against every golden vector / known-answer test
 * the reference's own tests hold for this path; see
 * tests/test_oracle_golden.py and tests/golden/.
 *
 * Every function cites the reference file:line it follows
 * (paths relative to the reference crate root).
 */
#ifndef KMER_ORACLE_H
#define KMER_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- src/kmer.rs ------------------------------------------------------- */

/* KmerLength::new, src/kmer.rs:100-110: 1 <= k <= 32, else error (-1). */
int ko_kmer_length_ok(uint64_t k);

/* Kmer::from_sub, src/kmer.rs:266-286.  Returns 0 if all k bytes are one of
 * ACGTacgt and writes the upper-cased copy to norm[0..k); otherwise returns -1
 * and reports the FIRST offending byte and its position inside the window. */
int ko_from_sub(const uint8_t *sub, size_t k, uint8_t *norm, uint8_t *err_base,
                size_t *err_pos);

/* pack_bytes, src/kmer.rs:467-471 with PACK_TABLE 21-32: acc=(acc<<2)|code,
 * first base most significant, A=0 C=1 G=2 T=3 (either case). */
uint64_t ko_pack_bytes(const uint8_t *bytes, size_t k);

/* Kmer::<Packed>::canonical, src/kmer.rs:348-390 (COMPLEMENT_TABLE 36-47):
 * two-ended lexicographic compare of the k-mer with its reverse complement;
 * reverse complement chosen only if strictly smaller (palindrome keeps the
 * original).  bytes must be upper-case ACGT.  Returns the canonical packed
 * bits; *is_rc (optional) receives the is_reverse_complement flag. */
uint64_t ko_canonical(const uint8_t *bytes, size_t k, int *is_rc);

/* unpack_to_bytes, src/kmer.rs:431-440: shift=(k-1-i)*2, UNPACK_TABLE 50.
 * Writes k chars + NUL to out. */
void ko_unpack(uint64_t packed, size_t k, char *out);

/* ---- count map (stands in for DashMap<u64,u64,Fx>, src/run.rs:489) ----- */

typedef struct ko_map ko_map;
ko_map  *ko_map_new(void);
void     ko_map_free(ko_map *m);
uint64_t ko_map_len(const ko_map *m);
/* entry(key).and_modify(saturating_add(addend)).or_insert(addend);
 * addend==1 is src/run.rs:565-571. */
void     ko_map_add(ko_map *m, uint64_t key, uint64_t addend);
uint64_t ko_map_get(const ko_map *m, uint64_t key); /* 0 if absent */
/* Copies up to cap (key,count) pairs, unordered; returns number written. */
uint64_t ko_map_dump(const ko_map *m, uint64_t *keys, uint64_t *counts, uint64_t cap);
/* Sum of all counts. */
uint64_t ko_map_total(const ko_map *m);

/* ---- src/run.rs hot loop ------------------------------------------------ */

/* KmerMap::process_sequence_with_quality, src/run.rs:526-563 (duplicates:
 * src/streaming.rs:622-660, 1068-1105): the LITERAL algorithm, including the
 * skip-ahead on bad quality / invalid base, per-window from_sub, O(k) pack and
 * O(k) canonical.  qual may be NULL; min_quality < 0 means None.
 * Threshold is min_quality.saturating_add(33) compared on raw ASCII. */
void ko_process_sequence(ko_map *m, const uint8_t *seq, size_t len,
                         const uint8_t *qual, size_t k, int min_quality);

/* Same results by the rolling formulation (forward/revcomp registers plus a
 * valid-run counter) -- the form the device kernel uses.  Kept separate so the
 * tests can prove literal == rolling on random inputs. */
void ko_process_sequence_rolling(ko_map *m, const uint8_t *seq, size_t len,
                                 const uint8_t *qual, size_t k, int min_quality);

/* Rolling scan of one flat buffer on nthreads threads (segments overlap by k-1
 * bytes so every window is seen exactly once).  Only keys with
 * (ko_mix64(key) & sample_mask) == 0 are added to m (sample_mask 0 = all keys);
 * returns the TOTAL number of valid windows (sampled or not).  Used by the
 * full-size parity checks, where a complete CPU map would not finish in seconds. */
uint64_t ko_scan_flat_sampled_mt(ko_map *m, const uint8_t *seq, size_t len,
                                 const uint8_t *qual, size_t k, int min_quality,
                                 uint64_t sample_mask, int nthreads);

/* Number of windows the literal loop counts (sum of counts it would add). */
uint64_t ko_count_valid_windows(const uint8_t *seq, size_t len, const uint8_t *qual,
                                size_t k, int min_quality);

/* compute_histogram_packed, src/histogram.rs:110-116, applied after the
 * min_count filter exactly as output_counts does (src/run.rs:447-450,471-481).
 * Writes ascending (count,freq) pairs; returns number of distinct counts
 * (may exceed cap; only cap are written). */
uint64_t ko_histogram(const ko_map *m, uint64_t min_count, uint64_t *count,
                      uint64_t *freq, uint64_t cap);

/* crc32 (IEEE, reflected, as src/index.rs:404-431). */
uint32_t ko_crc32(const uint8_t *data, size_t n);

/* ---- krust-equivalent threaded baseline (bench cpu_baseline "port") ----- */

/* KmerMap::build / build_with_quality, src/run.rs:500-520: one task per record
 * on nthreads threads (rayon for_each over records), each running the literal
 * per-window algorithm above and upserting into a sharded lock-per-shard map
 * (DashMap 5.5.3: shards = 4*nthreads rounded up to a power of two; the lock is a
 * parking mutex, as DashMap's RwLock parks its waiters).
 * Records are given as offsets into one flat buffer: record i is
 * seq[off[i] .. off[i]+lens[i]).  qual may be NULL.  Results are merged into m.
 * Returns the number of k-mers counted. */
uint64_t ko_count_records_mt(ko_map *m, const uint8_t *seq, const uint8_t *qual,
                             const uint64_t *off, const uint32_t *lens,
                             uint64_t nrec, size_t k, int min_quality, int nthreads);

/* The same with every shard's table sized up front for its share of expect_distinct keys (0 = start small and
 * grow under the lock, as above), and lock statistics: stats[0] contended acquisitions, stats[1] nanoseconds
 * spent waiting for shard locks (summed over threads), stats[2] upserts, stats[3] rehashes that still happened. */
uint64_t ko_count_records_mt2(ko_map *m, const uint8_t *seq, const uint8_t *qual,
                              const uint64_t *off, const uint32_t *lens,
                              uint64_t nrec, size_t k, int min_quality, int nthreads,
                              uint64_t expect_distinct, uint64_t *stats);

/* Optimised CPU formulation, reported beside the krust-equivalent port so that the GPU figure is
 * not flattered by the port's allocations and locks (implementation B): rolling
 * forward / reverse-complement registers, then a two-phase radix count -- every thread scatters
 * the keys of its slice of the flat buffer into 256 partitions by hash, then the partitions are
 * counted independently in private open-addressing tables (no locks, no merge).  Returns the
 * number of k-mers; *distinct and *digest (order-independent: sum of mix64(key ^ mix64(count)))
 * describe the result without materialising one big map. */
uint64_t ko_count_flat_radix_mt(const uint8_t *seq, size_t len, const uint8_t *qual, size_t k,
                                int min_quality, int nthreads, uint64_t *distinct, uint64_t *digest);
/* The same count in `npasses` passes over the input (bounded memory: pass j holds the keys of 256 / npasses
 * hash partitions only), optionally with the count-of-counts histogram after the min_count filter, ascending
 * (compute_histogram, src/histogram.rs:88-94 as used by output_counts, src/run.rs:447-450,471-481):
 * count / freq may be NULL; *n_pairs = number of distinct counts (only cap are written). */
uint64_t ko_hist_flat_radix_mt(const uint8_t *seq, size_t len, const uint8_t *qual, size_t k, int min_quality,
                               int nthreads, int npasses, uint64_t min_count, uint64_t *count, uint64_t *freq,
                               uint64_t cap, uint64_t *n_pairs, uint64_t *distinct, uint64_t *digest);
/* the same digest of an existing map (to cross-check the two CPU formulations and the GPU) */
uint64_t ko_map_digest(const ko_map *m);

/* ---- deterministic synthetic reads --------------- */

/* Counter-based generator shared (bit-exactly) with the device generator in
 * krust_amd/csrc.  See DESIGN.md "Synthetic workload" for the definition. */
uint64_t ko_mix64(uint64_t z);
/* Fills stride-(read_len+1) flat buffers: read_len bases then '\n'.  qual may
 * be NULL.  Reads [first_read, first_read+n_reads). */
void ko_synth_reads(uint64_t seed, uint64_t genome_len, uint32_t read_len,
                    uint64_t first_read, uint64_t n_reads, uint8_t *bases,
                    uint8_t *qual);


/* ---- "hg-like" synthetic assembly (BASELINE.json configs[4] stand-in) --- */

/* Counter-based generator of an assembly-shaped FASTA: nrec records of lens[r] bases (the caller passes
 * hg38's chromosome lengths), ~50 % soft-masked, ~5 % N in long runs, repeat families with copy numbers
 * from 1 to ~10^5, tandem repeats.  Writes sum(lens) + nrec bytes to out: every record followed by '\n'
 * (the flat layout the scan functions above take). */
void ko_synth_hg(uint64_t seed, const uint64_t *lens, uint64_t nrec, uint8_t *out, int nthreads);
/* The same records as FASTA text (">chr{r+1} ..." headers, lines of `width` columns); 0 or -1 (I/O). */
int ko_write_fasta(const char *path, const uint8_t *flat, const uint64_t *lens, uint64_t nrec, uint32_t width);

#ifdef __cplusplus
}
#endif
#endif
