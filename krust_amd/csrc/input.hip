// input.hip -- kh_push* and kh_push_text*: pinned staging, the device accumulation buffers, record scanning on the device
// (rawparse.hip.h).  Reference counterpart: the readers of src/reader.rs / src/streaming.rs feeding KmerMap::build.
#include "ctx.hip.h"
#include "rawparse.hip.h"

namespace khi {

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

// The copy stream and its events; and, for `need` > 0, the two pinned staging buffers through which PAGEABLE caller memory
// travels -- each 2 x stage_bytes (bases and qualities; or one text chunk), stage_bytes = the smallest of 1 / 8 / 64 MiB that
// holds `need` (round 5: every context that saw a host push pinned 256 MiB for them, ~50 ms -- of a ten-byte test input, and
// of every command-line run, whose chunks are pinned and never touch the staging at all: need = 0).
int ensure_stage(kh_ctx *c, u64 need) {
    if (!c->cstream) HIP_TRY(c, hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        if (!c->stage_done[i]) HIP_TRY(c, hipEventCreateWithFlags(&c->stage_done[i], hipEventDisableTiming));
        if (!c->acc_free[i]) HIP_TRY(c, hipEventCreateWithFlags(&c->acc_free[i], hipEventDisableTiming));
    }
    if (!need) return KH_OK;
    u64 want = 1ull << 20;
    while (want < need && want < STAGE_BYTES) want *= 8;
    want = std::min(want, STAGE_BYTES);
    if (c->stage_bytes >= want) return KH_OK;
    HIP_TRY(c, hipStreamSynchronize(c->cstream));  // (nothing in flight out of the buffers about to go)
    for (int i = 0; i < 2; ++i) {
        if (c->h_stage[i]) (void)hipHostFree(c->h_stage[i]);
        c->h_stage[i] = nullptr;
        c->stage_used[i] = false;
    }
    c->stage_bytes = 0;
    for (int i = 0; i < 2; ++i) {
        hipError_t e = hipHostMalloc((void **)&c->h_stage[i], 2 * want, hipHostMallocDefault);
        if (e != hipSuccess) return fail(c, KH_ERR_OOM, "hipHostMalloc(stage)", e);
    }
    c->stage_bytes = want;
    return KH_OK;
}

// Device -> pageable host memory through the two pinned staging buffers: the D2H of chunk i+1 runs
// while chunk i is copied out (by several threads: first-touch page faults of a fresh destination
// array cost more than the copy itself).
int d2h_staged(kh_ctx *c, void *dst, const void *d_src, u64 bytes) {
    const bool pinned_dst = is_pinned_host(dst);
    int rc = ensure_stage(c, pinned_dst ? 0 : (bytes + 1) / 2);
    if (rc != KH_OK) return rc;
    if (pinned_dst) {  // a registered destination takes the DMA itself: no bounce, no first-touch faults
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
    const u64 CH = 2 * c->stage_bytes;
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


}  // namespace khi
using namespace khi;

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
    // Pinned / registered source (kh_host_alloc, kh_host_register): the copy engine reads the caller's memory itself --
    // no staging memcpy (which, not PCIe, bounded kh_push from pageable memory: ~20-30 against 57 GB/s), no staging buffers.
    const bool direct = is_pinned_host(bases) && (!with_qual || is_pinned_host(qual));
    rc = ensure_stage(c, direct ? 0 : n);
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
        const u64 want = std::min(c->stage_bytes, n - off);
        if (c->acc_len && c->acc_len + want + 1 > c->acc_cap) {
            rc = flush_acc(c, off != 0);          // inside a push the seam needs the k-1 look-back
            if (rc != KH_OK) return rc;
        }
        const u64 len = std::min(want, c->acc_cap - c->acc_len - 1);
        const int p = c->stage_next;
        c->stage_next ^= 1;
        if (c->stage_used[p]) HIP_TRY(c, hipEventSynchronize(c->stage_done[p]));
        staged_memcpy(c->h_stage[p], bases + off, len);
        if (with_qual) staged_memcpy(c->h_stage[p] + c->stage_bytes, qual + off, len);
        hipEvent_t t0, t1;
        HIP_TRY(c, hipEventCreate(&t0));
        HIP_TRY(c, hipEventCreate(&t1));
        HIP_TRY(c, hipEventRecord(t0, c->cstream));
        uint8_t *dst = c->acc[c->acc_cur] + HALO + c->acc_len;
        HIP_TRY(c, hipMemcpyAsync(dst, c->h_stage[p], len, hipMemcpyHostToDevice, c->cstream));
        if (with_qual) HIP_TRY(c, hipMemcpyAsync(dst + stride, c->h_stage[p] + c->stage_bytes, len, hipMemcpyHostToDevice, c->cstream));
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
namespace khi {

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
    if ((rc = tbuf(&c->txt_err, &c->txt_err_cap, (u64)4, "hipMalloc(text err)")) != KH_OK) return rc;
    const unsigned grid = (unsigned)std::min<u64>(ntiles, (u64)GRID_CAP);
    u64 out_len = 0;
    {
        StageTimer tm(c, ST_TEXT, s);
        HIP_TRY(c, hipMemsetAsync(c->txt_err, 0, sizeof(uint32_t), s));
        HIP_TRY(c, hipMemcpyAsync(&c->h_txt->first, d_text, 1, hipMemcpyDeviceToHost, s));
        if (fastq) {
            if ((rc = tbuf(&c->txt_tnl, &c->txt_tnl_cap, ntiles, "hipMalloc(text tiles)")) != KH_OK) return rc;
            if ((rc = tbuf(&c->txt_tbase, &c->txt_tbase_cap, ntiles + 1, "hipMalloc(text tiles)")) != KH_OK) return rc;
            hipLaunchKernelGGL(kh::raw_nl_count_kernel, dim3(grid), dim3(kh::BLOCK), 0, s, d_text, n, ntiles, c->txt_tnl);
            if ((rc = text_device_scan(c, s, c->txt_tnl, ntiles, c->txt_tbase)) != KH_OK) return rc;
            HIP_TRY(c, hipMemcpyAsync(&c->h_txt->total, c->txt_tbase + ntiles, sizeof(u64), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipMemcpyAsync(&c->h_txt->last, d_text + n - 1, 1, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            const bool open_end = c->h_txt->last != '\n';           // no final newline: the text end closes the line
            const u64 nlines = c->h_txt->total + (open_end ? 1 : 0);
            if (c->h_txt->first != '@') return text_fail(c, "text does not start with '@'");
            if (nlines & 3) return text_fail(c, "FASTQ line count is not a multiple of 4");
            if ((rc = tbuf(&c->txt_ls, &c->txt_ls_cap, nlines + 2, "hipMalloc(line starts)")) != KH_OK) return rc;
            HIP_TRY(c, hipMemsetAsync(c->txt_ls, 0, sizeof(u64), s));
            hipLaunchKernelGGL(kh::raw_line_starts_kernel, dim3(grid), dim3(kh::BLOCK), 0, s, d_text, n, ntiles,
                               (const u64 *)c->txt_tbase, c->txt_ls);
            if (open_end) {
                c->h_txt->end_mark = n + 1;
                HIP_TRY(c, hipMemcpyAsync(c->txt_ls + nlines, &c->h_txt->end_mark, sizeof(u64), hipMemcpyHostToDevice, s));
            }
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
            // (no line starts: where header lines are is carried from one 1 KiB unit of the text to the next, rawparse.hip.h)
            const u64 nunits = ntiles * (kh::RAW_TILE / kh::FASTA_UNIT);  // (every unit of every tile has a state: those behind the text's end hold no line end)
            if ((rc = tbuf(&c->txt_st, &c->txt_st_cap, nunits + 1, "hipMalloc(line states)")) != KH_OK) return rc;
            if ((rc = tbuf(&c->txt_tkeep, &c->txt_tkeep_cap, ntiles, "hipMalloc(text tiles)")) != KH_OK) return rc;
            if ((rc = tbuf(&c->txt_tout, &c->txt_tout_cap, ntiles + 1, "hipMalloc(text tiles)")) != KH_OK) return rc;
            hipLaunchKernelGGL(kh::fasta_line_state_kernel, dim3((unsigned)std::min<u64>((nunits + 3) / 4, (u64)GRID_CAP)), dim3(kh::BLOCK),
                               0, s, d_text, n, nunits, c->txt_st);
            hipLaunchKernelGGL(kh::fasta_compact_kernel<0>, dim3(grid), dim3(kh::BLOCK), 0, s, d_text, n, ntiles,
                               (const uint8_t *)c->txt_st, c->txt_tkeep, (const u64 *)nullptr, (uint8_t *)nullptr, c->txt_err);
            if ((rc = text_device_scan(c, s, c->txt_tkeep, ntiles, c->txt_tout)) != KH_OK) return rc;
            HIP_TRY(c, hipMemcpyAsync(&c->h_txt->total, c->txt_tout + ntiles, sizeof(u64), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipMemcpyAsync(&c->h_txt->err, c->txt_err, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            if (c->h_txt->first != '>') return text_fail(c, "text does not start with '>'");
            if (c->h_txt->err) return text_fail(c, "blank before a line end, or a CR not followed by LF, inside a FASTA record");
            out_len = c->h_txt->total;
            hipLaunchKernelGGL(kh::fasta_compact_kernel<1>, dim3(grid), dim3(kh::BLOCK), 0, s, d_text, n, ntiles,
                               (const uint8_t *)c->txt_st, (uint32_t *)nullptr, (const u64 *)c->txt_tout, out, (uint32_t *)nullptr);
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

}  // namespace khi
namespace khi {
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

}  // namespace khi

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

namespace khi {
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
}  // namespace khi

extern "C" int kh_push_text(kh_ctx *c, const uint8_t *text, uint64_t n, int format) {
    // (what earlier calls have accumulated stays where it is: it is counted when its buffer is full, or by whatever looks
    //  at the table next)
    int rc = enter(c, false, false, false, true);
    if (rc != KH_OK) return rc;
    if ((rc = text_args(c, text, n, format)) != KH_OK) return rc;
    if (n == 0) return KH_OK;
    const bool pinned_text = is_pinned_host(text);
    if ((rc = ensure_stage(c, pinned_text ? 0 : (n + 1) / 2)) != KH_OK) return rc;
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
    if (pinned_text) {  // pinned / registered text: DMA straight from the caller's memory, no staging memcpy
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
    for (u64 off = 0; off < n; off += 2 * c->stage_bytes) {
        const u64 len = std::min(2 * c->stage_bytes, n - off);
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

