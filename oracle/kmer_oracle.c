/*
 * kmer_oracle.c -- CPU restatement of krust's canonical k-mer counting path.
 *
 * TEST INFRASTRUCTURE ONLY (see kmer_oracle.h).  Plain C11 + pthreads.
 * Parity is pinned against the reference's own known-answer tests
 * (tests/test_oracle_golden.py); the reference itself (Rust) cannot be built
 * in this image, so there is no oracle/_ref binary.
 */
#define _GNU_SOURCE /* PTHREAD_MUTEX_ADAPTIVE_NP */
#include "kmer_oracle.h"

#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ======================================================================== */
/* src/kmer.rs                                                              */
/* ======================================================================== */

int ko_kmer_length_ok(uint64_t k) { return (k >= 1 && k <= 32) ? 0 : -1; } /* kmer.rs:100-110 */

/* PACK_TABLE, src/kmer.rs:21-32.  Anything else maps to 0 there; callers only
 * look up validated bytes. */
static inline uint64_t pack_code(uint8_t b) {
    switch (b) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 0;
    }
}

/* COMPLEMENT_TABLE, src/kmer.rs:36-47 (result is always upper case). */
static inline uint8_t complement(uint8_t b) {
    switch (b) {
    case 'A': case 'a': return 'T';
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    case 'T': case 't': return 'A';
    default: return 0;
    }
}

int ko_from_sub(const uint8_t *sub, size_t k, uint8_t *norm, uint8_t *err_base,
                size_t *err_pos) {
    /* src/kmer.rs:266-286: first failing byte wins (iterator short-circuits) */
    for (size_t i = 0; i < k; i++) {
        uint8_t b = sub[i];
        switch (b) {
        case 'A': case 'C': case 'G': case 'T':
            norm[i] = b;
            break;
        case 'a': case 'c': case 'g': case 't':
            norm[i] = (uint8_t)(b - 32); /* to_ascii_uppercase */
            break;
        default:
            if (err_base) *err_base = b;
            if (err_pos) *err_pos = i;
            return -1;
        }
    }
    return 0;
}

uint64_t ko_pack_bytes(const uint8_t *bytes, size_t k) {
    /* src/kmer.rs:467-471 */
    uint64_t acc = 0;
    for (size_t i = 0; i < k; i++) acc = (acc << 2) | pack_code(bytes[i]);
    return acc;
}

uint64_t ko_canonical(const uint8_t *bytes, size_t k, int *is_rc) {
    /* src/kmer.rs:348-390 */
    int use_rc = 0; /* unwrap_or(false): palindrome keeps the original */
    for (size_t i = 0; i < k; i++) {
        uint8_t fwd = bytes[i];
        uint8_t rc = complement(bytes[k - 1 - i]);
        if (fwd < rc) { use_rc = 0; break; }
        if (fwd > rc) { use_rc = 1; break; }
    }
    if (is_rc) *is_rc = use_rc;
    if (!use_rc) return ko_pack_bytes(bytes, k);
    uint8_t rcb[32];
    for (size_t i = 0; i < k; i++) rcb[i] = complement(bytes[k - 1 - i]);
    return ko_pack_bytes(rcb, k); /* kmer.rs:369-375 */
}

void ko_unpack(uint64_t packed, size_t k, char *out) {
    /* src/kmer.rs:431-440 */
    static const char UNPACK_TABLE[4] = {'A', 'C', 'G', 'T'}; /* kmer.rs:50 */
    for (size_t i = 0; i < k; i++) {
        unsigned shift = (unsigned)((k - 1 - i) * 2);
        out[i] = UNPACK_TABLE[(packed >> shift) & 3u];
    }
    out[k] = '\0';
}

/* ======================================================================== */
/* count map                                                                */
/* ======================================================================== */

typedef struct { uint64_t key, val; } ko_entry; /* val == 0 marks a free entry */

struct ko_map {
    ko_entry *e;
    uint64_t cap; /* power of two */
    uint64_t len;
};

uint64_t ko_mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27; z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}

static void map_alloc(ko_map *m, uint64_t cap) {
    m->e = (ko_entry *)calloc(cap, sizeof(ko_entry));
    if (!m->e) abort();
    m->cap = cap;
    m->len = 0;
}

ko_map *ko_map_new(void) {
    ko_map *m = (ko_map *)malloc(sizeof(ko_map));
    if (!m) abort();
    map_alloc(m, 1024);
    return m;
}

void ko_map_free(ko_map *m) {
    if (!m) return;
    free(m->e); free(m);
}

uint64_t ko_map_len(const ko_map *m) { return m->len; }

static inline uint64_t sat_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    return s < a ? UINT64_MAX : s; /* saturating_add, run.rs:569 */
}

static void map_insert_raw(ko_map *m, uint64_t key, uint64_t addend) {
    uint64_t mask = m->cap - 1;
    uint64_t i = ko_mix64(key) & mask;
    while (m->e[i].val) {
        if (m->e[i].key == key) { m->e[i].val = sat_add(m->e[i].val, addend); return; }
        i = (i + 1) & mask;
    }
    m->e[i].key = key; m->e[i].val = addend; m->len++;
}

static void map_grow(ko_map *m) {
    ko_map old = *m;
    map_alloc(m, old.cap * 2);
    for (uint64_t i = 0; i < old.cap; i++)
        if (old.e[i].val) map_insert_raw(m, old.e[i].key, old.e[i].val);
    free(old.e);
}

void ko_map_add(ko_map *m, uint64_t key, uint64_t addend) {
    if (addend == 0) return; /* counts are >= 1 on this path; 0 is the free marker */
    if ((m->len + 1) * 10 > m->cap * 6) map_grow(m);
    map_insert_raw(m, key, addend);
}

uint64_t ko_map_get(const ko_map *m, uint64_t key) {
    uint64_t mask = m->cap - 1;
    uint64_t i = ko_mix64(key) & mask;
    while (m->e[i].val) {
        if (m->e[i].key == key) return m->e[i].val;
        i = (i + 1) & mask;
    }
    return 0;
}

uint64_t ko_map_dump(const ko_map *m, uint64_t *keys, uint64_t *counts, uint64_t cap) {
    uint64_t n = 0;
    for (uint64_t i = 0; i < m->cap && n < cap; i++)
        if (m->e[i].val) { keys[n] = m->e[i].key; counts[n] = m->e[i].val; n++; }
    return n;
}

uint64_t ko_map_total(const ko_map *m) {
    uint64_t t = 0;
    for (uint64_t i = 0; i < m->cap; i++)
        if (m->e[i].val) t = sat_add(t, m->e[i].val);
    return t;
}

/* ======================================================================== */
/* src/run.rs hot loop                                                      */
/* ======================================================================== */

typedef void (*emit_fn)(void *ctx, uint64_t key);

static inline int qual_threshold(int min_quality) {
    /* min_quality.map(|q| q.saturating_add(33)), run.rs:538 (u8 arithmetic) */
    int t = min_quality + 33;
    return t > 255 ? 255 : t;
}

/* The literal loop of src/run.rs:526-563 with an abstract upsert. */
static uint64_t literal_loop(const uint8_t *seq, size_t len, const uint8_t *qual,
                             size_t k, int min_quality, emit_fn emit, void *ctx) {
    uint64_t counted = 0;
    if (len < k) return 0; /* run.rs:533-535 */
    int have_thr = (qual != NULL && min_quality >= 0);
    int thr = have_thr ? qual_threshold(min_quality) : 0;
    size_t i = 0;
    uint8_t norm[32];
    while (i <= len - k) { /* run.rs:541 */
        if (have_thr) {    /* run.rs:543-548 */
            size_t bad = k;
            for (size_t j = 0; j < k; j++)
                if ((int)qual[i + j] < thr) { bad = j; break; }
            if (bad != k) { i += bad + 1; continue; }
        }
        uint8_t eb; size_t ep;
        if (ko_from_sub(seq + i, k, norm, &eb, &ep) == 0) { /* run.rs:552-556 */
            uint64_t key = ko_canonical(norm, k, NULL);     /* run.rs:566 */
            if (emit) emit(ctx, key);
            counted++;
            i += 1;
        } else {
            i += ep + 1; /* run.rs:557-560 */
        }
    }
    return counted;
}

static void emit_to_map(void *ctx, uint64_t key) { ko_map_add((ko_map *)ctx, key, 1); }

void ko_process_sequence(ko_map *m, const uint8_t *seq, size_t len,
                         const uint8_t *qual, size_t k, int min_quality) {
    literal_loop(seq, len, qual, k, min_quality, emit_to_map, m);
}

uint64_t ko_count_valid_windows(const uint8_t *seq, size_t len, const uint8_t *qual,
                                size_t k, int min_quality) {
    return literal_loop(seq, len, qual, k, min_quality, NULL, NULL);
}

void ko_process_sequence_rolling(ko_map *m, const uint8_t *seq, size_t len,
                                 const uint8_t *qual, size_t k, int min_quality) {
    /* Equivalent formulation: a window is counted iff all of its k positions
     * are "good" (base in ACGTacgt and, when filtering, qual >= thr); integer
     * min(fwd, rc) equals the lexicographic choice of kmer.rs:348-365 because
     * codes A<C<G<T are in ASCII order; on a tie both are the same bits. */
    int have_thr = (qual != NULL && min_quality >= 0);
    int thr = have_thr ? qual_threshold(min_quality) : 0;
    uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
    unsigned rcshift = (unsigned)(2 * (k - 1));
    uint64_t fwd = 0, rc = 0;
    size_t run = 0;
    for (size_t i = 0; i < len; i++) {
        uint8_t b = seq[i];
        uint8_t u = (uint8_t)(b & 0xDF);
        int good = (u == 'A' || u == 'C' || u == 'G' || u == 'T');
        if (good && have_thr && (int)qual[i] < thr) good = 0;
        if (!good) { run = 0; fwd = 0; rc = 0; continue; }
        uint64_t c = pack_code(b);
        fwd = ((fwd << 2) | c) & mask;
        rc = (rc >> 2) | ((3 - c) << rcshift);
        if (++run >= k) ko_map_add(m, fwd < rc ? fwd : rc, 1);
    }
}

/* ---- threaded sampled scan (full-size parity checks) -------------------- */

typedef struct {
    const uint8_t *seq, *qual;
    size_t lo, hi, len; /* count windows ENDING in [lo, hi) */
    size_t k;
    int min_quality;
    uint64_t sample_mask;
    ko_map map;
    uint64_t total;
} scan_t;

static void *scan_main(void *arg) {
    scan_t *s = (scan_t *)arg;
    const size_t k = s->k;
    int have_thr = (s->qual != NULL && s->min_quality >= 0);
    int thr = have_thr ? qual_threshold(s->min_quality) : 0;
    uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
    unsigned rcshift = (unsigned)(2 * (k - 1));
    uint64_t fwd = 0, rc = 0, total = 0;
    size_t run = 0;
    size_t begin = s->lo >= k - 1 ? s->lo - (k - 1) : 0; /* warm-up overlap */
    for (size_t i = begin; i < s->hi; i++) {
        uint8_t b = s->seq[i];
        uint8_t u = (uint8_t)(b & 0xDF);
        int good = (u == 'A' || u == 'C' || u == 'G' || u == 'T');
        if (good && have_thr && (int)s->qual[i] < thr) good = 0;
        if (!good) { run = 0; fwd = 0; rc = 0; continue; }
        uint64_t c = pack_code(b);
        fwd = ((fwd << 2) | c) & mask;
        rc = (rc >> 2) | ((3 - c) << rcshift);
        if (++run >= k && i >= s->lo) {
            uint64_t key = fwd < rc ? fwd : rc;
            total++;
            if ((ko_mix64(key) & s->sample_mask) == 0) ko_map_add(&s->map, key, 1);
        }
    }
    s->total = total;
    return NULL;
}

uint64_t ko_scan_flat_sampled_mt(ko_map *m, const uint8_t *seq, size_t len,
                                 const uint8_t *qual, size_t k, int min_quality,
                                 uint64_t sample_mask, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    scan_t *ss = (scan_t *)calloc((size_t)nthreads, sizeof(scan_t));
    if (!th || !ss) abort();
    size_t per = (len + (size_t)nthreads - 1) / (size_t)nthreads;
    for (int t = 0; t < nthreads; t++) {
        size_t lo = per * (size_t)t, hi = lo + per;
        if (lo > len) lo = len;
        if (hi > len) hi = len;
        ss[t] = (scan_t){seq, qual, lo, hi, len, k, min_quality, sample_mask, {0}, 0};
        map_alloc(&ss[t].map, 1024);
        pthread_create(&th[t], NULL, scan_main, &ss[t]);
    }
    uint64_t total = 0;
    for (int t = 0; t < nthreads; t++) {
        pthread_join(th[t], NULL);
        total += ss[t].total;
        ko_map *sm = &ss[t].map;
        for (uint64_t j = 0; j < sm->cap; j++)
            if (sm->e[j].val) ko_map_add(m, sm->e[j].key, sm->e[j].val);
        free(sm->e);
    }
    free(th); free(ss);
    return total;
}

/* ---- histogram, src/histogram.rs:110-116 + src/run.rs:447-450 ---------- */

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

uint64_t ko_histogram(const ko_map *m, uint64_t min_count, uint64_t *count,
                      uint64_t *freq, uint64_t cap) {
    uint64_t n = 0;
    uint64_t *v = (uint64_t *)malloc((m->len ? m->len : 1) * sizeof(uint64_t));
    if (!v) abort();
    for (uint64_t i = 0; i < m->cap; i++)
        if (m->e[i].val && m->e[i].val >= min_count) v[n++] = m->e[i].val;
    qsort(v, n, sizeof(uint64_t), cmp_u64);
    uint64_t nd = 0;
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i;
        while (j < n && v[j] == v[i]) j++;
        if (nd < cap) { count[nd] = v[i]; freq[nd] = j - i; }
        nd++;
        i = j;
    }
    free(v);
    return nd;
}

/* ---- crc32, src/index.rs:404-431 --------------------------------------- */

uint32_t ko_crc32(const uint8_t *data, size_t n) {
    uint32_t crc = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) {
        crc ^= data[i];
        for (int b = 0; b < 8; b++) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
    }
    return ~crc;
}

/* ======================================================================== */
/* krust-equivalent threaded baseline                                       */
/* ======================================================================== */

typedef struct {
    /* DashMap 5.5.3 guards every shard with a RwLock that spins briefly and then PARKS the thread
     * (parking_lot style).  A pthread mutex does the same through a futex; a pure test-and-set spinlock --
     * what stood here in round 1 -- burns the cgroup's CPU quota in the waiters and made this port ~12x
     * slower than it should be on a box that shows 256 CPUs but grants 16. */
    pthread_mutex_t lock;
    ko_map map;
    uint64_t contended, wait_ns; /* updated under the lock */
    char pad[64];
} shard_t;

typedef struct {
    shard_t *shards;
    uint64_t nshards; /* power of two */
    unsigned shift;
} sharded_t;

static inline uint64_t now_ns(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ULL + (uint64_t)ts.tv_nsec;
}

static void emit_to_shards(void *ctx, uint64_t key) {
    /* DashMap: hash -> shard index from the high bits, then lock + upsert
     * (run.rs:565-571).  FxHash is a multiplicative hash; any well-mixed hash
     * gives the same results, only speed differs. */
    sharded_t *s = (sharded_t *)ctx;
    uint64_t h = key * 0x517cc1b727220a95ULL; /* Fx-style multiply */
    shard_t *sh = &s->shards[(h >> s->shift) & (s->nshards - 1)];
    if (pthread_mutex_trylock(&sh->lock) != 0) { /* contended: time the wait (the uncontended path pays nothing) */
        uint64_t t0 = now_ns();
        pthread_mutex_lock(&sh->lock);
        sh->contended++;
        sh->wait_ns += now_ns() - t0;
    }
    ko_map_add(&sh->map, key, 1);
    pthread_mutex_unlock(&sh->lock);
}

typedef struct {
    sharded_t *sh;
    const uint8_t *seq, *qual;
    const uint64_t *off;
    const uint32_t *lens;
    uint64_t nrec;
    size_t k;
    int min_quality;
    atomic_ullong *next;
    uint64_t counted;
} worker_t;

static void *worker_main(void *arg) {
    worker_t *w = (worker_t *)arg;
    const uint64_t CHUNK = 256; /* records per steal, stands in for rayon splitting */
    uint64_t counted = 0;
    for (;;) {
        uint64_t b = atomic_fetch_add(w->next, CHUNK);
        if (b >= w->nrec) break;
        uint64_t e = b + CHUNK < w->nrec ? b + CHUNK : w->nrec;
        for (uint64_t r = b; r < e; r++)
            counted += literal_loop(w->seq + w->off[r], w->lens[r],
                                    w->qual ? w->qual + w->off[r] : NULL, w->k,
                                    w->min_quality, emit_to_shards, w->sh);
    }
    w->counted = counted;
    return NULL;
}

/* expect_distinct > 0: every shard's table is sized for its share up front, so that no thread ever rehashes
 * while holding a shard lock (hashbrown inside DashMap amortises its growth; a doubling under the lock of a
 * hot shard is not what a steady-state krust run looks like).  stats (optional, 4 entries): contended lock
 * acquisitions, nanoseconds spent waiting for locks (summed over threads), upserts, rehashes under a lock. */
uint64_t ko_count_records_mt2(ko_map *m, const uint8_t *seq, const uint8_t *qual,
                              const uint64_t *off, const uint32_t *lens,
                              uint64_t nrec, size_t k, int min_quality, int nthreads,
                              uint64_t expect_distinct, uint64_t *stats) {
    if (nthreads < 1) nthreads = 1;
    sharded_t sh;
    uint64_t ns = 1;
    while (ns < (uint64_t)nthreads * 4) ns <<= 1; /* DashMap default_shard_amount */
    sh.nshards = ns;
    unsigned bits = 0;
    while ((1ULL << bits) < ns) bits++;
    sh.shift = 64 - bits; /* ns >= 4, so bits >= 2 */
    sh.shards = (shard_t *)calloc(ns, sizeof(shard_t));
    if (!sh.shards) abort();
    uint64_t cap0 = 1024;
    if (expect_distinct) {
        uint64_t per = expect_distinct / ns + 1;
        while (cap0 * 6 < per * 10 + 16) cap0 <<= 1; /* stays under ko_map_add's 0.6 load limit */
        cap0 <<= 1;                                  /* and the shards are not perfectly even */
    }
    pthread_mutexattr_t attr; /* spin briefly, then park: what parking_lot's lock under DashMap does */
    pthread_mutexattr_init(&attr);
    pthread_mutexattr_settype(&attr, PTHREAD_MUTEX_ADAPTIVE_NP);
    for (uint64_t i = 0; i < ns; i++) {
        pthread_mutex_init(&sh.shards[i].lock, &attr);
        map_alloc(&sh.shards[i].map, cap0);
    }
    pthread_mutexattr_destroy(&attr);
    atomic_ullong next;
    atomic_init(&next, 0);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    worker_t *ws = (worker_t *)calloc((size_t)nthreads, sizeof(worker_t));
    if (!th || !ws) abort();
    for (int t = 0; t < nthreads; t++) {
        ws[t] = (worker_t){&sh, seq, qual, off, lens, nrec, k, min_quality, &next, 0};
        pthread_create(&th[t], NULL, worker_main, &ws[t]);
    }
    uint64_t counted = 0;
    for (int t = 0; t < nthreads; t++) {
        pthread_join(th[t], NULL);
        counted += ws[t].counted;
    }
    uint64_t contended = 0, wait_ns = 0, grown = 0;
    for (uint64_t i = 0; i < ns; i++) {
        ko_map *sm = &sh.shards[i].map;
        contended += sh.shards[i].contended;
        wait_ns += sh.shards[i].wait_ns;
        for (uint64_t c = cap0; c < sm->cap; c <<= 1) grown++;
        for (uint64_t j = 0; j < sm->cap; j++)
            if (sm->e[j].val) ko_map_add(m, sm->e[j].key, sm->e[j].val);
        free(sm->e);
        pthread_mutex_destroy(&sh.shards[i].lock);
    }
    if (stats) { stats[0] = contended; stats[1] = wait_ns; stats[2] = counted; stats[3] = grown; }
    free(sh.shards); free(th); free(ws);
    return counted;
}

uint64_t ko_count_records_mt(ko_map *m, const uint8_t *seq, const uint8_t *qual,
                             const uint64_t *off, const uint32_t *lens,
                             uint64_t nrec, size_t k, int min_quality, int nthreads) {
    return ko_count_records_mt2(m, seq, qual, off, lens, nrec, k, min_quality, nthreads, 0, NULL);
}

/* ======================================================================== */
/* optimised CPU formulation: rolling scan + two-phase radix count          */
/* ======================================================================== */

#define KO_RADIX_P 256
#define KO_HIST_DENSE 65536

typedef struct { uint64_t *v; uint64_t n, cap; } kvec_t;

typedef struct {
    const uint8_t *seq, *qual;
    size_t lo, hi;
    size_t k;
    int min_quality;
    unsigned plo, phi; /* only keys of partitions [plo, phi) are kept in this pass */
    kvec_t part[KO_RADIX_P];
    uint64_t total;
} rscan_t;

static inline void kvec_push(kvec_t *a, uint64_t x) {
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 4096;
        a->v = (uint64_t *)realloc(a->v, a->cap * sizeof(uint64_t));
        if (!a->v) abort();
    }
    a->v[a->n++] = x;
}

static void *rscan_main(void *arg) {
    rscan_t *s = (rscan_t *)arg;
    const size_t k = s->k;
    int have_thr = (s->qual != NULL && s->min_quality >= 0);
    int thr = have_thr ? qual_threshold(s->min_quality) : 0;
    uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
    unsigned rcshift = (unsigned)(2 * (k - 1));
    uint64_t fwd = 0, rc = 0, total = 0;
    size_t run = 0;
    size_t begin = s->lo >= k - 1 ? s->lo - (k - 1) : 0;
    for (size_t i = begin; i < s->hi; i++) {
        uint8_t b = s->seq[i];
        uint8_t u = (uint8_t)(b & 0xDF);
        int good = (u == 'A' || u == 'C' || u == 'G' || u == 'T');
        if (good && have_thr && (int)s->qual[i] < thr) good = 0;
        if (!good) { run = 0; fwd = 0; rc = 0; continue; }
        uint64_t c = pack_code(b);
        fwd = ((fwd << 2) | c) & mask;
        rc = (rc >> 2) | ((3 - c) << rcshift);
        if (++run >= k && i >= s->lo) {
            uint64_t key = fwd < rc ? fwd : rc;
            total++;
            unsigned p = (unsigned)(ko_mix64(key) >> 56);
            if (p >= s->plo && p < s->phi) kvec_push(&s->part[p], key);
        }
    }
    s->total = total;
    return NULL;
}

typedef struct {
    rscan_t *scans;
    int nscan;
    atomic_ullong *next;
    unsigned phi;
    uint64_t min_count;
    uint64_t *dense; /* [KO_HIST_DENSE] per thread, or NULL: no histogram wanted */
    kvec_t big;      /* counts >= KO_HIST_DENSE */
    uint64_t distinct, digest;
} rcount_t;

uint64_t ko_map_digest(const ko_map *m) {
    uint64_t d = 0;
    for (uint64_t i = 0; i < m->cap; i++)
        if (m->e[i].val) d += ko_mix64(m->e[i].key ^ ko_mix64(m->e[i].val));
    return d;
}

static void *rcount_main(void *arg) {
    rcount_t *c = (rcount_t *)arg;
    uint64_t distinct = 0, digest = 0;
    for (;;) {
        uint64_t p = atomic_fetch_add(c->next, 1);
        if (p >= c->phi) break;
        uint64_t n = 0;
        for (int t = 0; t < c->nscan; t++) n += c->scans[t].part[p].n;
        if (!n) continue;
        uint64_t cap = 1024;
        while (cap < n * 2) cap <<= 1; /* sized once: no growth, no rehash */
        ko_map m;
        map_alloc(&m, cap);
        for (int t = 0; t < c->nscan; t++) {
            kvec_t *a = &c->scans[t].part[p];
            for (uint64_t i = 0; i < a->n; i++) map_insert_raw(&m, a->v[i], 1);
            free(a->v); /* the partition's keys are not needed again */
            a->v = NULL; a->n = a->cap = 0;
        }
        distinct += m.len;
        digest += ko_map_digest(&m);
        if (c->dense)
            for (uint64_t i = 0; i < m.cap; i++) {
                uint64_t v = m.e[i].val;
                if (!v || v < c->min_count) continue;
                if (v < KO_HIST_DENSE) c->dense[v]++;
                else kvec_push(&c->big, v);
            }
        free(m.e);
    }
    c->distinct = distinct;
    c->digest = digest;
    return NULL;
}

/* The radix count in `npasses` passes over the input (pass j keeps only the keys of 256 / npasses hash
 * partitions, so the key lists never hold more than ~1 / npasses of all k-mers at a time: an hg38-sized
 * input would otherwise need > 25 GB of lists).  With count != NULL also the count-of-counts histogram
 * after the min_count filter, ascending (compute_histogram, src/histogram.rs:88-94 as used by
 * src/run.rs:447-450,471-481); *n_pairs receives the number of distinct counts (only cap are written). */
uint64_t ko_hist_flat_radix_mt(const uint8_t *seq, size_t len, const uint8_t *qual, size_t k, int min_quality,
                               int nthreads, int npasses, uint64_t min_count, uint64_t *count, uint64_t *freq,
                               uint64_t cap, uint64_t *n_pairs, uint64_t *distinct, uint64_t *digest) {
    if (nthreads < 1) nthreads = 1;
    if (npasses < 1) npasses = 1;
    if (npasses > KO_RADIX_P) npasses = KO_RADIX_P;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    rscan_t *ss = (rscan_t *)calloc((size_t)nthreads, sizeof(rscan_t));
    rcount_t *cs = (rcount_t *)calloc((size_t)nthreads, sizeof(rcount_t));
    uint64_t *dense = count ? (uint64_t *)calloc((size_t)nthreads * KO_HIST_DENSE, sizeof(uint64_t)) : NULL;
    if (!th || !ss || !cs || (count && !dense)) abort();
    size_t per = (len + (size_t)nthreads - 1) / (size_t)nthreads;
    uint64_t total = 0, d = 0, g = 0;
    for (int pass = 0; pass < npasses; pass++) {
        const unsigned plo = (unsigned)((uint64_t)KO_RADIX_P * (uint64_t)pass / (uint64_t)npasses);
        const unsigned phi = (unsigned)((uint64_t)KO_RADIX_P * (uint64_t)(pass + 1) / (uint64_t)npasses);
        for (int t = 0; t < nthreads; t++) {
            size_t lo = per * (size_t)t, hi = lo + per;
            if (lo > len) lo = len;
            if (hi > len) hi = len;
            ss[t].seq = seq; ss[t].qual = qual; ss[t].lo = lo; ss[t].hi = hi; ss[t].k = k;
            ss[t].min_quality = min_quality; ss[t].plo = plo; ss[t].phi = phi;
            pthread_create(&th[t], NULL, rscan_main, &ss[t]);
        }
        uint64_t tot = 0;
        for (int t = 0; t < nthreads; t++) { pthread_join(th[t], NULL); tot += ss[t].total; }
        total = tot; /* every pass sees every window */
        atomic_ullong next;
        atomic_init(&next, plo);
        for (int t = 0; t < nthreads; t++) {
            cs[t].scans = ss; cs[t].nscan = nthreads; cs[t].next = &next; cs[t].phi = phi;
            cs[t].min_count = min_count;
            cs[t].dense = dense ? dense + (size_t)t * KO_HIST_DENSE : NULL;
            pthread_create(&th[t], NULL, rcount_main, &cs[t]);
        }
        for (int t = 0; t < nthreads; t++) { pthread_join(th[t], NULL); d += cs[t].distinct; g += cs[t].digest; }
    }
    for (int t = 0; t < nthreads; t++)
        for (int p = 0; p < KO_RADIX_P; p++) free(ss[t].part[p].v);
    if (count) {
        uint64_t nd = 0;
        for (uint64_t v = 1; v < KO_HIST_DENSE; v++) {
            uint64_t f = 0;
            for (int t = 0; t < nthreads; t++) f += dense[(size_t)t * KO_HIST_DENSE + v];
            if (!f) continue;
            if (nd < cap) { count[nd] = v; freq[nd] = f; }
            nd++;
        }
        uint64_t nbig = 0;
        for (int t = 0; t < nthreads; t++) nbig += cs[t].big.n;
        uint64_t *bv = (uint64_t *)malloc((nbig ? nbig : 1) * sizeof(uint64_t));
        if (!bv) abort();
        uint64_t o = 0;
        for (int t = 0; t < nthreads; t++) {
            for (uint64_t i = 0; i < cs[t].big.n; i++) bv[o++] = cs[t].big.v[i];
            free(cs[t].big.v);
        }
        qsort(bv, nbig, sizeof(uint64_t), cmp_u64);
        for (uint64_t i = 0; i < nbig;) {
            uint64_t j = i;
            while (j < nbig && bv[j] == bv[i]) j++;
            if (nd < cap) { count[nd] = bv[i]; freq[nd] = j - i; }
            nd++;
            i = j;
        }
        free(bv);
        if (n_pairs) *n_pairs = nd;
    }
    free(dense); free(th); free(ss); free(cs);
    if (distinct) *distinct = d;
    if (digest) *digest = g;
    return total;
}

uint64_t ko_count_flat_radix_mt(const uint8_t *seq, size_t len, const uint8_t *qual, size_t k,
                                int min_quality, int nthreads, uint64_t *distinct, uint64_t *digest) {
    return ko_hist_flat_radix_mt(seq, len, qual, k, min_quality, nthreads, 1, 1, NULL, NULL, 0, NULL, distinct, digest);
}

/* ======================================================================== */
/* deterministic synthetic reads                                            */
/* ======================================================================== */

#define KO_GOLDEN 0x9E3779B97F4A7C15ULL

static inline uint64_t stream_key(uint64_t seed, uint64_t s) {
    return ko_mix64(seed + (s + 1) * KO_GOLDEN);
}
static inline uint64_t draw(uint64_t key, uint64_t idx) {
    return ko_mix64(key + idx * KO_GOLDEN);
}

void ko_synth_reads(uint64_t seed, uint64_t genome_len, uint32_t read_len,
                    uint64_t first_read, uint64_t n_reads, uint8_t *bases,
                    uint8_t *qual) {
    static const char ACGT[4] = {'A', 'C', 'G', 'T'};
    const uint64_t kg = stream_key(seed, 0), ks = stream_key(seed, 1),
                   kd = stream_key(seed, 2), ke = stream_key(seed, 3);
    const uint64_t span = genome_len - read_len + 1;
    const uint64_t stride = (uint64_t)read_len + 1;
    for (uint64_t i = 0; i < n_reads; i++) {
        uint64_t r = first_read + i;
        uint64_t start = draw(ks, r) % span;
        int strand = (int)(draw(kd, r) & 1);
        uint8_t *bo = bases + i * stride;
        uint8_t *qo = qual ? qual + i * stride : NULL;
        for (uint32_t j = 0; j < read_len; j++) {
            uint64_t c;
            if (!strand) c = draw(kg, start + j) & 3;
            else c = 3 - (draw(kg, start + (read_len - 1 - j)) & 3);
            uint64_t u = draw(ke, r * read_len + j);
            int subst = ((u & 0xFF) == 0);
            if (subst) c = (c + 1 + ((u >> 8) % 3)) & 3;
            int isn = (((u >> 16) & 0x3FF) == 0);
            bo[j] = isn ? 'N' : (uint8_t)ACGT[c];
            if (qo) {
                unsigned v = (unsigned)((u >> 32) & 0xFF);
                uint8_t q = v < 230 ? 'I' : (v < 250 ? '5' : '#');
                if (subst && ((u >> 40) & 1)) q = '#';
                qo[j] = q;
            }
        }
        bo[read_len] = '\n';
        if (qo) qo[read_len] = '\n';
    }
}

/* ======================================================================== */
/* "hg-like" synthetic assembly (BASELINE.json configs[4] stand-in)          */
/* ======================================================================== */
/* hg38 itself is not on the box.  This generator has the features of a mammalian assembly that matter to
 * the counting path: a few very long records (the caller passes hg38's chromosome lengths), ~50 % of the
 * bases soft-masked (lowercase, in 1 KiB blocks), ~5 % N in long runs (gaps) plus a run at every record
 * start (telomere), interspersed repeat families whose copy numbers span five orders of magnitude (diverged
 * copies: 1 substitution in 64), and tandem repeats / homopolymer tracts (counts far above 2^16).  It is
 * counter based -- base i depends only on (seed, i) -- so any range can be produced by any thread. */

#define HG_BLOCK 4096ULL

static inline uint8_t hg_base_at(uint64_t kg, uint64_t kb, uint64_t km, uint64_t kn, uint64_t kc, uint64_t i) {
    static const char ACGT[4] = {'A', 'C', 'G', 'T'};
    const uint64_t blk = i / HG_BLOCK, j = i % HG_BLOCK;
    if (draw(kn, blk >> 4) % 20 == 0) return 'N';               /* gaps: runs of 16 blocks, ~5 % */
    const uint64_t u = draw(kb, blk);
    const unsigned type = (unsigned)(u & 7);
    uint64_t c;
    if (type == 7 && ((u >> 3) & 15) == 0) {                    /* tandem repeat / homopolymer block */
        const unsigned unit = 1 + (unsigned)((u >> 8) % 6);     /* unit length 1..6 */
        const uint64_t motif = draw(kg, (u >> 16) % 5 + (1ULL << 50)); /* five motifs in all */
        c = (motif >> (2 * (j % unit))) & 3;
    } else if (type < 3) {                                      /* copy of a repeat-family consensus */
        const unsigned fbits = (unsigned)((u >> 40) % 17);      /* family drawn from 2^fbits: copy numbers vary */
        const uint64_t fam = (u >> 8) & ((1ULL << fbits) - 1);
        c = draw(kg, (1ULL << 40) + fam * HG_BLOCK + j) & 3;
        const uint64_t m = draw(km, i);
        if ((m & 63) == 0) c = (c + 1 + ((m >> 8) % 3)) & 3;    /* diverged copies */
    } else {
        c = draw(kg, i) & 3;                                    /* unique sequence */
    }
    uint8_t b = (uint8_t)ACGT[c];
    if (draw(kc, i >> 10) & 1) b |= 0x20;                       /* soft-masked 1 KiB blocks, ~50 % */
    return b;
}

typedef struct {
    uint64_t seed;
    const uint64_t *rec_off; /* global start of every record, nrec + 1 entries */
    uint64_t nrec;
    uint8_t *out;            /* flat: records separated by '\n' */
    uint64_t lo, hi;         /* output byte range of this thread */
} hg_job_t;

static void *hg_main(void *arg) {
    hg_job_t *j = (hg_job_t *)arg;
    const uint64_t kg = stream_key(j->seed, 10), kb = stream_key(j->seed, 11), km = stream_key(j->seed, 12),
                   kn = stream_key(j->seed, 13), kc = stream_key(j->seed, 14);
    /* output position o of record r, offset x: o = rec_off[r] + r + x  (one '\n' after every record) */
    uint64_t r = 0;
    while (r + 1 < j->nrec && j->rec_off[r + 1] + (r + 1) <= j->lo) r++;
    for (uint64_t o = j->lo; o < j->hi; o++) {
        while (r + 1 < j->nrec && j->rec_off[r + 1] + (r + 1) <= o) r++;
        const uint64_t x = o - (j->rec_off[r] + r);
        const uint64_t rlen = j->rec_off[r + 1] - j->rec_off[r];
        if (x == rlen) { j->out[o] = '\n'; continue; }
        const uint64_t gi = j->rec_off[r] + x;
        j->out[o] = (x < 10000 && rlen > 1000000) ? (uint8_t)'N' : hg_base_at(kg, kb, km, kn, kc, gi);
    }
    return NULL;
}

/* Writes sum(lens) + nrec bytes: record r (lens[r] bases) followed by '\n'. */
void ko_synth_hg(uint64_t seed, const uint64_t *lens, uint64_t nrec, uint8_t *out, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    uint64_t *off = (uint64_t *)malloc((nrec + 1) * sizeof(uint64_t));
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    hg_job_t *jobs = (hg_job_t *)calloc((size_t)nthreads, sizeof(hg_job_t));
    if (!off || !th || !jobs) abort();
    off[0] = 0;
    for (uint64_t r = 0; r < nrec; r++) off[r + 1] = off[r] + lens[r];
    const uint64_t total = off[nrec] + nrec;
    const uint64_t per = (total + (uint64_t)nthreads - 1) / (uint64_t)nthreads;
    for (int t = 0; t < nthreads; t++) {
        uint64_t lo = per * (uint64_t)t, hi = lo + per;
        if (lo > total) lo = total;
        if (hi > total) hi = total;
        jobs[t] = (hg_job_t){seed, off, nrec, out, lo, hi};
        pthread_create(&th[t], NULL, hg_main, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(off); free(th); free(jobs);
}

#include <stdio.h>
/* The same records as FASTA text: ">chr{r+1} ...\n" then lines of `width` columns.  Returns 0, or -1 on
 * an I/O error. */
int ko_write_fasta(const char *path, const uint8_t *flat, const uint64_t *lens, uint64_t nrec, uint32_t width) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    static char iobuf[1 << 22];
    setvbuf(f, iobuf, _IOFBF, sizeof iobuf);
    uint64_t o = 0;
    int ok = 1;
    for (uint64_t r = 0; r < nrec && ok; r++) {
        ok = fprintf(f, ">chr%llu hg-like synthetic record, %llu bp\n", (unsigned long long)(r + 1),
                     (unsigned long long)lens[r]) > 0;
        for (uint64_t x = 0; x < lens[r] && ok; x += width) {
            const uint64_t n = lens[r] - x < width ? lens[r] - x : width;
            ok = fwrite(flat + o + x, 1, n, f) == n && fputc('\n', f) != EOF;
        }
        o += lens[r] + 1;
    }
    if (fclose(f) != 0) ok = 0;
    return ok ? 0 : -1;
}
