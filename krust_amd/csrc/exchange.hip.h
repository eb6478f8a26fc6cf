// exchange.hip.h -- the multi-GPU exchange behind the C ABI: kh_comm_* / kh_merge_across / kh_group_*.
//
// Included by kmerhip.hip (same translation unit: it drives the export / merge entry points defined there).
// north_star: "reads shard naturally per GPU across the 8 x MI355X node with a final RCCL reduce of per-GPU
// hash tables over xGMI".  A hash table is not element-wise reducible, so the "reduce" is an all-to-all of
// region segments to owner = top bits of the table hash, then an LDS rebuild of every owner's shard
// (shard.hip.h).  xGMI is point to point (7 links per GPU): ncclSend / ncclRecv groups drive all links of a
// GPU at once, where a ring all-reduce would be per-link bound -- and wrong for a hash table anyway.
//
// Reference counterpart: none (single process; rayon over records, src/run.rs:500-503, is its only
// parallelism).  krust_amd/distributed.py is the same sequence over torch.distributed and stays as the
// test harness; tests assert that both leave identical shard tables.
//
// Transport: RCCL when every rank has its own device; a process-local hub (device-to-device copies between
// the threads' contexts) when a kh_group lists a device twice -- the 1-GPU test box -- because RCCL refuses
// duplicate devices.  Both sit behind `Xport`, so the merge sequence is one piece of code.
#pragma once

#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <mutex>

namespace {

// ---- process-local hub: ranks are threads of one process --------------------------------------------
struct LocalHub {
    uint32_t n;
    std::mutex m;
    std::condition_variable cv;
    uint32_t arrived = 0;
    u64 generation = 0;
    std::vector<const void *> base;              // posted per rank
    std::vector<std::vector<u64>> off, len;      // [rank][peer], bytes
    std::vector<std::vector<u64>> small;         // all-gather postings
    explicit LocalHub(uint32_t nr) : n(nr), base(nr, nullptr), off(nr), len(nr), small(nr) {}
    void barrier() {
        std::unique_lock<std::mutex> lk(m);
        const u64 gen = generation;
        if (++arrived == n) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen; });
        }
    }
};

struct Comm {
    uint32_t nranks = 1, rank = 0;
    ncclComm_t nccl = nullptr;
    LocalHub *hub = nullptr;     // not owned
    hipStream_t xs = nullptr;    // the exchange stream (transfers overlap the kernels on ctx->stream)
    u64 *d_small = nullptr;      // device staging of the small all-gathers: (1 + nranks) * SMALL_MAX u64
    u64 *h_small = nullptr;      // pinned twin
};
constexpr uint32_t SMALL_MAX = 128;  // u64 per rank in one small all-gather

int rccl_fail(kh_ctx *c, const char *what, ncclResult_t r) {
    c->last_error = std::string(what) + ": " + ncclGetErrorString(r);
    return KH_ERR_RCCL;
}
#define NCCL_TRY(c, call)                                          \
    do {                                                           \
        ncclResult_t r_ = (call);                                  \
        if (r_ != ncclSuccess) return rccl_fail((c), #call, r_);   \
    } while (0)

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// all-gather of n (<= SMALL_MAX) host integers: all[r * n + i] = rank r's mine[i].  Blocks.
int xp_allgather(kh_ctx *c, const u64 *mine, uint32_t n, u64 *all) {
    Comm *cm = c->comm;
    if (n > SMALL_MAX) return fail(c, KH_ERR_BAD_ARG, "xp_allgather: too many values");
    if (cm->hub) {
        cm->hub->small[cm->rank].assign(mine, mine + n);
        cm->hub->barrier();
        for (uint32_t r = 0; r < cm->nranks; ++r) memcpy(all + (size_t)r * n, cm->hub->small[r].data(), n * sizeof(u64));
        cm->hub->barrier();  // nobody overwrites its posting before everybody has read it
        return KH_OK;
    }
    memcpy(cm->h_small, mine, n * sizeof(u64));
    HIP_TRY(c, hipMemcpyAsync(cm->d_small, cm->h_small, n * sizeof(u64), hipMemcpyHostToDevice, cm->xs));
    NCCL_TRY(c, ncclAllGather(cm->d_small, cm->d_small + SMALL_MAX, n, ncclUint64, cm->nccl, cm->xs));
    HIP_TRY(c, hipMemcpyAsync(cm->h_small + SMALL_MAX, cm->d_small + SMALL_MAX, (size_t)cm->nranks * n * sizeof(u64),
                              hipMemcpyDeviceToHost, cm->xs));
    HIP_TRY(c, hipStreamSynchronize(cm->xs));
    memcpy(all, cm->h_small + SMALL_MAX, (size_t)cm->nranks * n * sizeof(u64));
    return KH_OK;
}

// all-to-all of device byte ranges: peer p receives send[soff[p] .. +slen[p]) and this rank receives peer
// p's range for it at recv + roff[p] (rlen[p] bytes; the caller has exchanged the sizes).  Enqueued on the
// exchange stream: returns once the transfers are IN FLIGHT; xp_done() records their completion.
// `send` must be complete in device memory (the export calls block until it is) and stay untouched until
// the completion event has been waited for -- with the hub also until the closing barrier of the merge.
int xp_alltoallv(kh_ctx *c, const void *send, const u64 *soff, const u64 *slen, void *recv, const u64 *roff, const u64 *rlen) {
    Comm *cm = c->comm;
    if (cm->hub) {
        LocalHub *h = cm->hub;
        h->base[cm->rank] = send;
        h->off[cm->rank].assign(soff, soff + cm->nranks);
        h->len[cm->rank].assign(slen, slen + cm->nranks);
        h->barrier();
        int rc = KH_OK;  // (no return between the two barriers: the other threads would wait for ever)
        for (uint32_t p = 0; p < cm->nranks && rc == KH_OK; ++p) {
            const u64 n = h->len[p][cm->rank];
            if (n != rlen[p]) rc = fail(c, KH_ERR_STATE, "local exchange: announced and posted segment sizes differ");
            else if (n && hipMemcpyAsync((char *)recv + roff[p], (const char *)h->base[p] + h->off[p][cm->rank], n, hipMemcpyDefault,
                                         cm->xs) != hipSuccess)
                rc = fail(c, KH_ERR_HIP, "hipMemcpyAsync(local exchange)");
        }
        h->barrier();  // postings may be replaced (the COPIES are still in flight: buffers stay alive, see above)
        return rc;
    }
    NCCL_TRY(c, ncclGroupStart());
    for (uint32_t p = 0; p < cm->nranks; ++p) {
        if (slen[p]) NCCL_TRY(c, ncclSend((const char *)send + soff[p], slen[p], ncclUint8, (int)p, cm->nccl, cm->xs));
        if (rlen[p]) NCCL_TRY(c, ncclRecv((char *)recv + roff[p], rlen[p], ncclUint8, (int)p, cm->nccl, cm->xs));
    }
    NCCL_TRY(c, ncclGroupEnd());
    return KH_OK;
}

int xp_allreduce_sum_u64(kh_ctx *c, u64 *d_buf, u64 n) {
    Comm *cm = c->comm;
    if (cm->hub) return fail(c, KH_ERR_STATE, "dense all-reduce is not available on the process-local hub");
    NCCL_TRY(c, ncclAllReduce(d_buf, d_buf, n, ncclUint64, ncclSum, cm->nccl, cm->xs));
    HIP_TRY(c, hipStreamSynchronize(cm->xs));
    return KH_OK;
}

struct DevBuf {  // scratch of one merge; freed when it goes out of scope
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(kh_ctx *c, u64 bytes, const char *what) {
        if (p) (void)hipFree(p);
        p = nullptr;
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            return fail(c, KH_ERR_OOM, what, e);
        }
        return KH_OK;
    }
};

uint32_t merge_pieces_default() {
    const char *e = getenv("KMERHIP_MERGE_PIECES");
    const int v = e ? atoi(e) : 4;
    return v >= 1 ? (uint32_t)v : 1u;
}

// agreement among the ranks on a few integers: true iff every rank's value i equals `mine[i]`... the
// callers only need min / all-equal, so the raw gather is handed back
int vote(kh_ctx *c, std::initializer_list<u64> mine, std::vector<u64> &all) {
    std::vector<u64> m(mine);
    all.assign((size_t)c->comm->nranks * m.size(), 0);
    return xp_allgather(c, m.data(), (uint32_t)m.size(), all.data());
}

// ---- the merge sequence -----------------------------------------------------------------------------
int merge_across_impl(kh_ctx *c, kh_merge_info *info) {
    Comm *cm = c->comm;
    const uint32_t W = cm->nranks, R = cm->rank;
    const double t_begin = now_ms();
    double t_export = 0, t_wait = 0, t_merge = 0;
    kh_merge_info mi;
    memset(&mi, 0, sizeof(mi));
    mi.nranks = W;
    mi.pieces = 1;
    auto done = [&](int rc) {
        (void)kh_set_region_window(c, 0, 1);  // whatever happened: later exports / merges cover the whole range again
        mi.export_ms = t_export;
        mi.wait_ms = t_wait;
        mi.merge_ms = t_merge;
        mi.total_ms = now_ms() - t_begin;
        if (info) *info = mi;
        return rc;
    };
    int rc = kh_finish(c, nullptr);
    if (rc != KH_OK) return done(rc);
    const u64 n_local = c->h_ctr->distinct;
    const u64 nreg = c->cap / kh::REGION_SLOTS;
    mi.local_distinct = n_local;
    std::vector<u64> all;

    // ---- small k: the key space as a dense count array IS element-wise reducible ----
    if (2 * c->k <= 26 && !cm->hub) {
        const u64 n = 1ull << (2 * c->k);
        DevBuf dense;
        if ((rc = dense.alloc(c, n * sizeof(u64), "hipMalloc(dense)")) != KH_OK) return done(rc);
        double t0 = now_ms();
        if ((rc = kh_export_dense_device(c, (uint64_t *)dense.p, n)) != KH_OK) return done(rc);  // blocks until complete
        t_export += now_ms() - t0;
        t0 = now_ms();
        if ((rc = xp_allreduce_sum_u64(c, (u64 *)dense.p, n)) != KH_OK) return done(rc);
        t_wait += now_ms() - t0;
        t0 = now_ms();
        if ((rc = kh_reset(c)) != KH_OK) return done(rc);
        if ((rc = kh_merge_dense_device(c, (const uint64_t *)dense.p, n, R, W)) != KH_OK) return done(rc);
        if ((rc = kh_finish(c, nullptr)) != KH_OK) return done(rc);
        t_merge += now_ms() - t0;
        mi.route = KH_ROUTE_DENSE;
        mi.unit_bytes = 8;
        mi.sent_units = mi.recv_units = n;
        mi.owned_distinct = c->h_ctr->distinct;
        return done(KH_OK);
    }

    const bool pow2 = (W & (W - 1)) == 0;
    const bool regions_ok = pow2 && W <= (uint32_t)kh::MAX_SENDERS && nreg >= W;
    uint32_t npieces = merge_pieces_default();
    bool piped = regions_ok && npieces > 1 && (npieces & (npieces - 1)) == 0 && npieces <= 64 && (nreg / W) >= 64ull * npieces;
    if (!piped) npieces = 1;

    // send buffer: heads (2 per key), packed (1 u64 per key) and one array of wide pairs all fit 8 B x n_local
    DevBuf sendbuf, sendcnt, rcnt, counts_all;
    const u64 cap_units64 = std::max<u64>(n_local, 1);
    if ((rc = sendbuf.alloc(c, cap_units64 * 8, "hipMalloc(exchange send buffer)")) != KH_OK) return done(rc);
    if ((rc = rcnt.alloc(c, std::max<u64>(nreg, 1) * sizeof(uint32_t), "hipMalloc(region counts)")) != KH_OK) return done(rc);
    std::vector<uint64_t> parts(W, 0);
    uint64_t treg = 0;
    // fmt: 2 heads, 1 packed, 0 = neither fits
    auto export_fmt = [&](int fmt, void *dst, u64 cap) -> int {
        const double t0 = now_ms();
        int r = export_regions(c, fmt == 2 ? XF_HEADS32 : XF_PACKED64, W, dst, nullptr, cap, (uint32_t *)rcnt.p, nreg, parts.data(), &treg);
        t_export += now_ms() - t0;
        return r;
    };
    int my_fmt = 0;
    if (regions_ok) {
        if (piped && (rc = kh_set_region_window(c, 0, npieces)) != KH_OK) return done(rc);
        for (int fmt : {2, 1}) {  // speculative: the export itself finds out whether the counts fit
            rc = export_fmt(fmt, sendbuf.p, fmt == 2 ? 2 * n_local : n_local);
            if (rc == KH_OK) {
                my_fmt = fmt;
                break;
            }
            if (rc != KH_ERR_RANGE) return done(rc);
        }
    }
    double t0 = now_ms();
    if ((rc = vote(c, {nreg, (u64)my_fmt, (u64)piped}, all)) != KH_OK) return done(rc);
    t_wait += now_ms() - t0;
    bool same_size = true, all_piped = true;
    u64 agreed = 3;
    for (uint32_t r = 0; r < W; ++r) {
        same_size &= all[3 * r] == nreg;
        agreed = std::min(agreed, all[3 * r + 1]);
        all_piped &= all[3 * r + 2] != 0;
    }
    if (!(regions_ok && same_size)) agreed = 0;
    if (piped && !(agreed && agreed == (u64)my_fmt && all_piped)) {
        // some rank cannot run the pipeline in this rank's format: everybody takes the one-shot route
        piped = false;
        npieces = 1;
        (void)kh_set_region_window(c, 0, 1);
        my_fmt = -1;  // the windowed export does not describe the whole table: export again below
    }
    if (piped) {
        // every piece's size is known before anything is sent: a table needing more units than the send
        // buffer holds is found out HERE, and all ranks leave the pipeline together
        const uint32_t ub = agreed == 2 ? 4 : 8;
        if ((rc = counts_all.alloc(c, nreg * sizeof(uint32_t), "hipMalloc(unit counts)")) != KH_OK) return done(rc);
        uint64_t treg2 = 0;
        t0 = now_ms();
        rc = kh_region_unit_counts_device(c, ub, (uint32_t *)counts_all.p, nreg, &treg2);
        t_export += now_ms() - t0;
        bool fits = rc == KH_OK;
        if (rc != KH_OK && rc != KH_ERR_RANGE) return done(rc);
        std::vector<uint32_t> hcounts;
        if (fits) {
            hcounts.resize(nreg);
            HIP_TRY(c, hipMemcpy(hcounts.data(), counts_all.p, nreg * sizeof(uint32_t), hipMemcpyDeviceToHost));
            u64 total = 0;
            for (uint32_t v : hcounts) total += v;
            fits = total <= (agreed == 2 ? 2 * n_local : n_local);
        }
        t0 = now_ms();
        if ((rc = vote(c, {(u64)fits}, all)) != KH_OK) return done(rc);
        t_wait += now_ms() - t0;
        bool all_fit = true;
        for (uint32_t r = 0; r < W; ++r) all_fit &= all[r] != 0;
        if (!all_fit) {
            piped = false;
            npieces = 1;
            (void)kh_set_region_window(c, 0, 1);
            my_fmt = -1;
        } else {
            // ---- pipeline over the pieces: [export i+1 | transfer i], then [merge i | transfers > i] ----
            const u64 per = nreg / W, wper = per / npieces;
            // sizes of all pieces, announced up front: send_mat[owner][piece]
            std::vector<u64> send_mat((size_t)W * npieces, 0), recv_mat((size_t)W * npieces, 0);
            for (uint32_t o = 0; o < W; ++o)
                for (uint32_t i = 0; i < npieces; ++i) {
                    u64 s = 0;
                    const uint32_t *p = hcounts.data() + (u64)o * per + (u64)i * wper;
                    for (u64 q = 0; q < wper; ++q) s += p[q];
                    send_mat[(size_t)o * npieces + i] = s;
                }
            // recv_mat[sender][piece]: what each sender has for ME -- gather everybody's matrix (W * npieces <= 4096
            // values, in SMALL_MAX slices)
            {
                t0 = now_ms();
                std::vector<u64> allm((size_t)W * W * npieces);
                const uint32_t tot = W * npieces;
                for (uint32_t b = 0; b < tot; b += SMALL_MAX) {
                    const uint32_t n = std::min<uint32_t>(SMALL_MAX, tot - b);
                    std::vector<u64> g((size_t)W * n);
                    if ((rc = xp_allgather(c, send_mat.data() + b, n, g.data())) != KH_OK) return done(rc);
                    for (uint32_t r = 0; r < W; ++r) memcpy(&allm[(size_t)r * tot + b], &g[(size_t)r * n], n * sizeof(u64));
                }
                for (uint32_t s = 0; s < W; ++s)
                    for (uint32_t i = 0; i < npieces; ++i) recv_mat[(size_t)s * npieces + i] = allm[(size_t)s * tot + (size_t)R * npieces + i];
                t_wait += now_ms() - t0;
            }
            // every sender's unit counts of MY regions: W slices of `per` counts
            DevBuf rrc_full;
            if ((rc = rrc_full.alloc(c, nreg * sizeof(uint32_t), "hipMalloc(region counts in)")) != KH_OK) return done(rc);
            {
                std::vector<u64> so(W), sl(W), ro(W), rl(W);
                for (uint32_t p = 0; p < W; ++p) {
                    so[p] = (u64)p * per * 4;
                    sl[p] = per * 4;
                    ro[p] = (u64)p * per * 4;
                    rl[p] = per * 4;
                }
                t0 = now_ms();
                if ((rc = xp_alltoallv(c, counts_all.p, so.data(), sl.data(), rrc_full.p, ro.data(), rl.data())) != KH_OK) return done(rc);
                HIP_TRY(c, hipStreamSynchronize(cm->xs));
                t_wait += now_ms() - t0;
            }
            struct Flight {
                DevBuf buf;
                std::vector<u64> roff;  // byte offsets per sender
                u64 units = 0;
                hipEvent_t ev = nullptr;
            };
            std::vector<Flight> flights(npieces);
            u64 used = 0;  // units of the send buffer in use
            const u64 cap_total = agreed == 2 ? 2 * n_local : n_local;
            int frc = KH_OK;
            for (uint32_t i = 0; i < npieces && frc == KH_OK; ++i) {
                void *dst = (char *)sendbuf.p + used * ub;
                if (i > 0) {
                    if ((frc = kh_set_region_window(c, i, npieces)) != KH_OK) break;
                    if ((frc = export_fmt((int)agreed, dst, cap_total - used)) != KH_OK) break;  // (sizes were checked: cannot be RANGE)
                }
                std::vector<u64> so(W), sl(W), rl(W);
                u64 o = 0, rtot = 0;
                flights[i].roff.resize(W);
                for (uint32_t p = 0; p < W; ++p) {
                    if (parts[p] != send_mat[(size_t)p * npieces + i]) frc = fail(c, KH_ERR_STATE, "piece sizes differ from the announced ones");
                    so[p] = o * ub;
                    sl[p] = parts[p] * ub;
                    o += parts[p];
                    flights[i].roff[p] = rtot * ub;
                    rl[p] = recv_mat[(size_t)p * npieces + i] * ub;
                    rtot += recv_mat[(size_t)p * npieces + i];
                }
                if (frc != KH_OK) break;
                mi.sent_units += o - parts[R];
                used += o;
                flights[i].units = rtot;
                if ((frc = flights[i].buf.alloc(c, rtot * ub, "hipMalloc(exchange receive buffer)")) != KH_OK) break;
                t0 = now_ms();
                if ((frc = xp_alltoallv(c, dst, so.data(), sl.data(), flights[i].buf.p, flights[i].roff.data(), rl.data())) != KH_OK) break;
                if (hipEventCreateWithFlags(&flights[i].ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(flights[i].ev, cm->xs) != hipSuccess)
                    frc = fail(c, KH_ERR_HIP, "hipEventRecord(exchange)");
                t_wait += now_ms() - t0;
            }
            if (frc == KH_OK) frc = kh_set_region_window(c, 0, 1);
            if (frc == KH_OK) frc = kh_reset(c);
            if (frc == KH_OK) frc = kh_set_shard(c, R, W);
            // the senders' region counts as the merge of piece i wants them: zero outside the piece
            DevBuf rrc;
            if (frc == KH_OK) frc = rrc.alloc(c, nreg * sizeof(uint32_t), "hipMalloc(region counts piece)");
            for (uint32_t i = 0; i < npieces && frc == KH_OK; ++i) {
                t0 = now_ms();
                if (hipMemsetAsync(rrc.p, 0, nreg * sizeof(uint32_t), c->stream) != hipSuccess) frc = fail(c, KH_ERR_HIP, "hipMemsetAsync(rrc)");
                for (uint32_t s = 0; s < W && frc == KH_OK; ++s) {
                    const u64 o = ((u64)s * per + (u64)i * wper) * 4;
                    if (hipMemcpyAsync((char *)rrc.p + o, (const char *)rrc_full.p + o, wper * 4, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
                        frc = fail(c, KH_ERR_HIP, "hipMemcpyAsync(rrc)");
                }
                if (frc == KH_OK && hipEventSynchronize(flights[i].ev) != hipSuccess) frc = fail(c, KH_ERR_HIP, "hipEventSynchronize(exchange)");
                t_wait += now_ms() - t0;
                if (frc != KH_OK) break;
                t0 = now_ms();
                std::vector<const void *> kp(W);
                std::vector<const uint32_t *> rp(W);
                for (uint32_t s = 0; s < W; ++s) {
                    kp[s] = (const char *)flights[i].buf.p + flights[i].roff[s];
                    rp[s] = (const uint32_t *)rrc.p + (u64)s * per;
                }
                if ((frc = kh_set_region_window(c, i, npieces)) != KH_OK) break;
                frc = merge_regions(c, agreed == 2 ? XF_HEADS32 : XF_PACKED64, W, nreg, kp.data(), nullptr, rp.data());
                mi.recv_units += flights[i].units;
                t_merge += now_ms() - t0;
            }
            for (auto &f : flights)
                if (f.ev) {
                    (void)hipEventSynchronize(f.ev);  // (error paths: nothing may still write into buffers about to be freed)
                    (void)hipEventDestroy(f.ev);
                }
            if (cm->hub) cm->hub->barrier();  // every peer has finished copying out of this rank's send buffer
            if (frc != KH_OK) return done(frc);
            (void)kh_set_region_window(c, 0, 1);
            if ((rc = kh_finish(c, nullptr)) != KH_OK) return done(rc);
            mi.route = agreed == 2 ? KH_ROUTE_REGIONS_HEADS : KH_ROUTE_REGIONS_PACKED;
            mi.pieces = npieces;
            mi.unit_bytes = ub;
            mi.owned_distinct = c->h_ctr->distinct;
            return done(KH_OK);
        }
    }
    if (my_fmt < 0 && regions_ok) {  // left the pipeline: one shot, whole table, narrowest unit that fits
        my_fmt = 0;
        for (int fmt : {2, 1}) {
            rc = export_fmt(fmt, sendbuf.p, fmt == 2 ? 2 * n_local : n_local);
            if (rc == KH_OK) {
                my_fmt = fmt;
                break;
            }
            if (rc != KH_ERR_RANGE) return done(rc);
        }
        t0 = now_ms();
        if ((rc = vote(c, {(u64)my_fmt}, all)) != KH_OK) return done(rc);
        t_wait += now_ms() - t0;
        agreed = 3;
        for (uint32_t r = 0; r < W; ++r) agreed = std::min(agreed, all[r]);
        if (!same_size) agreed = 0;
    }
    if (agreed && agreed != (u64)my_fmt) {  // another rank could not go as narrow: redo in the common format
        rc = export_fmt((int)agreed, sendbuf.p, agreed == 2 ? 2 * n_local : n_local);
        if (rc != KH_OK) return done(rc);
    }

    // one all-to-all of per-owner unit counts, then the data; shared by the three one-shot routes below
    auto exchange_sizes = [&](const std::vector<uint64_t> &send_units_in, std::vector<u64> &recv_units) -> int {
        std::vector<u64> send_units(send_units_in.begin(), send_units_in.end());
        std::vector<u64> g;
        recv_units.assign(W, 0);
        for (uint32_t b = 0; b < W; b += SMALL_MAX) {
            const uint32_t n = std::min<uint32_t>(SMALL_MAX, W - b);
            g.assign((size_t)W * n, 0);
            int r = xp_allgather(c, send_units.data() + b, n, g.data());
            if (r != KH_OK) return r;
            if (R >= b && R < b + n)
                for (uint32_t s = 0; s < W; ++s) recv_units[s] = g[(size_t)s * n + (R - b)];
        }
        return KH_OK;
    };
    auto a2a_units = [&](const void *send, const std::vector<uint64_t> &su, const std::vector<u64> &ru, u64 ub, void *recv) -> int {
        std::vector<u64> so(W), sl(W), ro(W), rl(W);
        u64 a = 0, b = 0;
        for (uint32_t p = 0; p < W; ++p) {
            so[p] = a * ub;
            sl[p] = su[p] * ub;
            a += su[p];
            ro[p] = b * ub;
            rl[p] = ru[p] * ub;
            b += ru[p];
        }
        return xp_alltoallv(c, send, so.data(), sl.data(), recv, ro.data(), rl.data());
    };

    if (agreed || (regions_ok && same_size)) {
        const bool wide = !agreed;
        const u64 ub = agreed == 2 ? 4 : 8;
        const u64 per = nreg / W;
        if (wide) {  // (u64 key, u64 count): two arrays
            if ((rc = sendcnt.alloc(c, cap_units64 * 8, "hipMalloc(exchange send counts)")) != KH_OK) return done(rc);
            t0 = now_ms();
            rc = export_regions(c, XF_WIDE, W, sendbuf.p, (uint64_t *)sendcnt.p, n_local, (uint32_t *)rcnt.p, nreg, parts.data(), &treg);
            t_export += now_ms() - t0;
            if (rc != KH_OK) return done(rc);
        }
        std::vector<u64> recv_units;
        t0 = now_ms();
        if ((rc = exchange_sizes(parts, recv_units)) != KH_OK) return done(rc);
        u64 rtot = 0;
        for (u64 v : recv_units) rtot += v;
        DevBuf rbuf, rbuf2, rrc;
        if ((rc = rbuf.alloc(c, rtot * ub, "hipMalloc(exchange receive buffer)")) != KH_OK) return done(rc);
        if ((rc = rrc.alloc(c, nreg * sizeof(uint32_t), "hipMalloc(region counts in)")) != KH_OK) return done(rc);
        if ((rc = a2a_units(sendbuf.p, parts, recv_units, ub, rbuf.p)) != KH_OK) return done(rc);
        if (wide) {
            if ((rc = rbuf2.alloc(c, rtot * 8, "hipMalloc(exchange receive counts)")) != KH_OK) return done(rc);
            if ((rc = a2a_units(sendcnt.p, parts, recv_units, 8, rbuf2.p)) != KH_OK) return done(rc);
        }
        {
            std::vector<u64> so(W), sl(W, per * 4);
            for (uint32_t p = 0; p < W; ++p) so[p] = (u64)p * per * 4;
            if ((rc = xp_alltoallv(c, rcnt.p, so.data(), sl.data(), rrc.p, so.data(), sl.data())) != KH_OK) return done(rc);
        }
        HIP_TRY(c, hipStreamSynchronize(cm->xs));  // the merge kernels run on the context's stream
        t_wait += now_ms() - t0;
        t0 = now_ms();
        if ((rc = kh_reset(c)) != KH_OK) return done(rc);
        if ((rc = kh_set_shard(c, R, W)) != KH_OK) return done(rc);
        std::vector<const void *> kp(W);
        std::vector<const uint64_t *> cp(W);
        std::vector<const uint32_t *> rp(W);
        u64 o = 0;
        for (uint32_t s = 0; s < W; ++s) {
            kp[s] = (const char *)rbuf.p + o * ub;
            cp[s] = wide ? (const uint64_t *)rbuf2.p + o : nullptr;
            rp[s] = (const uint32_t *)rrc.p + (u64)s * per;
            o += recv_units[s];
        }
        rc = merge_regions(c, wide ? XF_WIDE : (agreed == 2 ? XF_HEADS32 : XF_PACKED64), W, nreg, kp.data(), wide ? cp.data() : nullptr, rp.data());
        if (cm->hub) cm->hub->barrier();
        if (rc != KH_OK) return done(rc);
        if ((rc = kh_finish(c, nullptr)) != KH_OK) return done(rc);
        t_merge += now_ms() - t0;
        mi.route = wide ? KH_ROUTE_REGIONS_WIDE : (agreed == 2 ? KH_ROUTE_REGIONS_HEADS : KH_ROUTE_REGIONS_PACKED);
        mi.unit_bytes = wide ? 16 : (uint32_t)ub;
        for (uint32_t p = 0; p < W; ++p) mi.sent_units += p == R ? 0 : parts[p];
        mi.recv_units = rtot;
        mi.owned_distinct = c->h_ctr->distinct;
        return done(KH_OK);
    }

    // ---- generic route: any world size, tables of any size; device-atomic re-insert ----
    {
        if ((rc = sendcnt.alloc(c, cap_units64 * 8, "hipMalloc(exchange send counts)")) != KH_OK) return done(rc);
        t0 = now_ms();
        rc = kh_export_by_owner_device(c, W, (uint64_t *)sendbuf.p, (uint64_t *)sendcnt.p, n_local, parts.data());  // blocks until complete
        t_export += now_ms() - t0;
        if (rc != KH_OK) return done(rc);
        std::vector<u64> recv_units;
        t0 = now_ms();
        if ((rc = exchange_sizes(parts, recv_units)) != KH_OK) return done(rc);
        u64 rtot = 0;
        for (u64 v : recv_units) rtot += v;
        DevBuf rk, rcn;
        if ((rc = rk.alloc(c, rtot * 8, "hipMalloc(exchange receive keys)")) != KH_OK) return done(rc);
        if ((rc = rcn.alloc(c, rtot * 8, "hipMalloc(exchange receive counts)")) != KH_OK) return done(rc);
        if ((rc = a2a_units(sendbuf.p, parts, recv_units, 8, rk.p)) != KH_OK) return done(rc);
        if ((rc = a2a_units(sendcnt.p, parts, recv_units, 8, rcn.p)) != KH_OK) return done(rc);
        HIP_TRY(c, hipStreamSynchronize(cm->xs));
        t_wait += now_ms() - t0;
        t0 = now_ms();
        if ((rc = kh_reset(c)) != KH_OK) return done(rc);
        rc = kh_merge_pairs_device(c, (const uint64_t *)rk.p, (const uint64_t *)rcn.p, rtot);
        if (rc == KH_OK) rc = kh_finish(c, nullptr);  // (also: the kernels are done with rk / rcn before they are freed)
        if (cm->hub) cm->hub->barrier();
        if (rc != KH_OK) return done(rc);
        t_merge += now_ms() - t0;
        mi.route = KH_ROUTE_PAIRS;
        mi.unit_bytes = 16;
        for (uint32_t p = 0; p < W; ++p) mi.sent_units += p == R ? 0 : parts[p];
        mi.recv_units = rtot;
        mi.owned_distinct = c->h_ctr->distinct;
        return done(KH_OK);
    }
}

void comm_release(kh_ctx *c) {
    Comm *cm = c->comm;
    if (!cm) return;
    if (cm->xs) (void)hipStreamSynchronize(cm->xs);
    if (cm->nccl) (void)ncclCommDestroy(cm->nccl);
    if (cm->d_small) (void)hipFree(cm->d_small);
    if (cm->h_small) (void)hipHostFree(cm->h_small);
    if (cm->xs) (void)hipStreamDestroy(cm->xs);
    delete cm;
    c->comm = nullptr;
}

int comm_setup(kh_ctx *c, uint32_t nranks, uint32_t rank, const ncclUniqueId *id, LocalHub *hub) {
    if (c->comm) return fail(c, KH_ERR_STATE, "the context already has a communicator");
    if (nranks < 1 || rank >= nranks) return fail(c, KH_ERR_BAD_ARG, "kh_comm_init: rank must be < nranks");
    HIP_TRY(c, hipSetDevice(c->device));
    Comm *cm = new (std::nothrow) Comm();
    if (!cm) return fail(c, KH_ERR_OOM, "Comm");
    cm->nranks = nranks;
    cm->rank = rank;
    cm->hub = hub;
    c->comm = cm;
    int rc = KH_OK;
    if (hipStreamCreateWithFlags(&cm->xs, hipStreamNonBlocking) != hipSuccess) rc = fail(c, KH_ERR_HIP, "hipStreamCreate(exchange)");
    if (rc == KH_OK && !hub) {
        const size_t n = (size_t)(1 + nranks) * SMALL_MAX * sizeof(u64);
        if (hipMalloc((void **)&cm->d_small, n) != hipSuccess || hipHostMalloc((void **)&cm->h_small, n, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            rc = fail(c, KH_ERR_OOM, "exchange staging");
        }
        if (rc == KH_OK) {
            ncclResult_t r = ncclCommInitRank(&cm->nccl, (int)nranks, *id, (int)rank);
            if (r != ncclSuccess) {
                cm->nccl = nullptr;
                rc = rccl_fail(c, "ncclCommInitRank", r);
            }
        }
    }
    if (rc != KH_OK) comm_release(c);
    return rc;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" int kh_comm_unique_id(kh_unique_id *out) {
    static_assert(sizeof(kh_unique_id) == sizeof(ncclUniqueId), "kh_unique_id must be an ncclUniqueId");
    if (!out) return KH_ERR_BAD_ARG;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return KH_ERR_RCCL;
    memcpy(out, &id, sizeof(id));
    return KH_OK;
}

extern "C" int kh_comm_init(kh_ctx *c, uint32_t nranks, uint32_t rank, const kh_unique_id *id) {
    int rc = enter(c);
    if (rc != KH_OK) return rc;
    if (!id) return fail(c, KH_ERR_BAD_ARG, "kh_comm_init: id is NULL");
    ncclUniqueId nid;
    memcpy(&nid, id, sizeof(nid));
    return comm_setup(c, nranks, rank, &nid, nullptr);
}

extern "C" int kh_merge_across(kh_ctx *c, kh_merge_info *info) {
    int rc = enter(c);
    if (rc != KH_OK) return rc;
    if (c->shard_shift) return fail(c, KH_ERR_STATE, "the table is already a shard (merged before); kh_reset first");
    if (!c->comm) {  // a lone context is its own world
        if (info) {
            memset(info, 0, sizeof(*info));
            info->nranks = 1;
            info->pieces = 1;
        }
        rc = kh_finish(c, nullptr);
        if (rc == KH_OK && info) info->local_distinct = info->owned_distinct = c->h_ctr->distinct;
        return rc;
    }
    return merge_across_impl(c, info);
}

// ---- single-process form: one context and one host thread per device --------------------------------
struct kh_group {
    std::vector<kh_ctx *> ctx;
    LocalHub *hub = nullptr;
};

extern "C" void kh_group_destroy(kh_group *g) {
    if (!g) return;
    for (kh_ctx *c : g->ctx)
        if (c) kh_destroy(c);
    delete g->hub;
    delete g;
}

extern "C" int kh_group_create(kh_group **out, const kh_config *cfg, const int32_t *devices, uint32_t ndev) {
    if (!out || !cfg || !devices || ndev < 1 || ndev > 64) return KH_ERR_BAD_ARG;
    *out = nullptr;
    if (cfg->struct_size != sizeof(kh_config)) return KH_ERR_BAD_ARG;
    kh_group *g = new (std::nothrow) kh_group();
    if (!g) return KH_ERR_OOM;
    bool dup = false;
    for (uint32_t i = 0; i < ndev; ++i)
        for (uint32_t j = 0; j < i; ++j) dup |= devices[i] == devices[j];
    int rc = KH_OK;
    for (uint32_t i = 0; i < ndev && rc == KH_OK; ++i) {
        kh_config c2 = *cfg;
        c2.device = devices[i];
        c2.stream = nullptr;
        c2.flags &= ~KH_FLAG_CALLER_STREAM;
        kh_ctx *c = nullptr;
        rc = kh_create(&c, &c2);
        g->ctx.push_back(c);
    }
    if (rc == KH_OK && ndev > 1) {
        if (dup) {
            g->hub = new (std::nothrow) LocalHub(ndev);
            if (!g->hub) rc = KH_ERR_OOM;
            for (uint32_t i = 0; i < ndev && rc == KH_OK; ++i) rc = comm_setup(g->ctx[i], ndev, i, nullptr, g->hub);
        } else {
            // ncclCommInitRank blocks until every rank has arrived: one thread per rank
            ncclUniqueId id;
            if (ncclGetUniqueId(&id) != ncclSuccess) rc = KH_ERR_RCCL;
            std::vector<int> rcs(ndev, KH_OK);
            if (rc == KH_OK) {
                std::vector<std::thread> th;
                for (uint32_t i = 0; i < ndev; ++i)
                    th.emplace_back([&, i] { rcs[i] = comm_setup(g->ctx[i], ndev, i, &id, nullptr); });
                for (auto &t : th) t.join();
                for (int r : rcs)
                    if (r != KH_OK && rc == KH_OK) rc = r;
            }
        }
    }
    if (rc != KH_OK) {
        kh_group_destroy(g);
        return rc;
    }
    *out = g;
    return KH_OK;
}

extern "C" kh_ctx *kh_group_ctx(kh_group *g, uint32_t rank) { return (g && rank < g->ctx.size()) ? g->ctx[rank] : nullptr; }
extern "C" uint32_t kh_group_size(const kh_group *g) { return g ? (uint32_t)g->ctx.size() : 0; }

extern "C" int kh_group_merge(kh_group *g, kh_merge_info *infos) {
    if (!g) return KH_ERR_BAD_ARG;
    const uint32_t n = (uint32_t)g->ctx.size();
    std::vector<int> rcs(n, KH_OK);
    if (n == 1) return kh_merge_across(g->ctx[0], infos);
    std::vector<std::thread> th;
    for (uint32_t i = 0; i < n; ++i) th.emplace_back([&, i] { rcs[i] = kh_merge_across(g->ctx[i], infos ? infos + i : nullptr); });
    for (auto &t : th) t.join();
    for (int r : rcs)
        if (r != KH_OK) return r;
    return KH_OK;
}
