"""Multi-GPU leg of the counting path: one process per GPU, reads sharded by contiguous
read-index ranges, per-GPU tables merged by ONE key-partitioned exchange step.

A hash table is not element-wise reducible, so the "reduce of per-GPU tables" is an
all-to-all of compacted (key,count) pairs to owner = kh_owner(key, world) followed by a local
re-insert with `count` as the addend.  On ROCm torch.distributed's "nccl"
backend is RCCL; an all-to-all drives all 7 xGMI links of a GPU at once, where a ring
all-reduce would be per-link bound (and semantically wrong for a hash table).  The result stays
key-sharded across the ranks.

The reference has no counterpart (it is single-process; src/run.rs:500-503 is its only
parallelism).  torch is used for device buffers and the collective only.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) range of `n_items` for `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _host_staged(group):
    """gloo has no device all-to-all: with it (CPU tests, and several ranks sharing ONE GPU in
    tests/test_gpu_dist.py) device tensors make the trip through host memory.  RCCL ("nccl") moves
    device buffers directly over xGMI."""
    return dist.get_backend(group) == "gloo"


def _all_to_all(out, inp, out_splits=None, in_splits=None, group=None):
    # NOTE (round 4, tools/world1_merge_probe.py): over RCCL a message of 2^30 bytes or more arrived half.  The exchange inside
    # the library (exchange.hip, xp_alltoallv) cuts its segments into 256 MiB messages; this harness sends a segment as ONE
    # message -- fine for the sizes the tests run and for worlds of 4+ at the bench's sizes, not for two ranks with S100M tables.
    _refuse_huge_messages(out, inp, out_splits, in_splits, group)
    if out.is_cuda and _host_staged(group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)


MAX_MESSAGE_BYTES = 1 << 30


def _refuse_huge_messages(out, inp, out_splits, in_splits, group):
    """This harness sends a segment as ONE message.  Until tools/rccl/rccl_big_msg.hip has cleared the transport (round 4: a
    message of >= 2^30 bytes arrived half), a segment that large is refused LOUDLY instead of risking a quietly smaller
    table; kh_merge_across cuts its segments into 256 MiB messages and checks what arrived (exchange.hip)."""
    if _host_staged(group):
        return
    world = dist.get_world_size(group)
    for t, splits in ((inp, in_splits), (out, out_splits)):
        sizes = splits if splits is not None else [t.numel() // world] * world
        if sizes and max(int(x) for x in sizes) * t.element_size() >= MAX_MESSAGE_BYTES:
            raise RuntimeError(f"krust_amd.distributed: a segment of {max(int(x) for x in sizes) * t.element_size()} bytes would travel as one RCCL "
                               "message (>= 2^30 bytes); use kh_merge_across (DeviceCounter.merge_across), which splits and verifies")


def exchange_pairs(keys, counts, part_counts, group=None, return_sizes=False):
    """All-to-all of owner-grouped pairs.

    keys/counts: int64 tensors (bit views of u64) laid out as [pairs for owner 0 | owner 1 | ...];
    part_counts: per-owner pair counts (len == world).  Returns (recv_keys, recv_counts): every
    pair this rank owns, from all ranks (its own included), sender by sender; with return_sizes
    also the per-sender counts."""
    world = dist.get_world_size(group)
    send_sizes = [int(x) for x in part_counts]
    assert len(send_sizes) == world and sum(send_sizes) == keys.numel() == counts.numel()
    dev = keys.device
    ssz = torch.tensor(send_sizes, dtype=torch.int64, device=dev)
    rsz = torch.empty(world, dtype=torch.int64, device=dev)
    _all_to_all(rsz, ssz, group=group)
    recv_sizes = [int(x) for x in rsz.tolist()]
    rk = torch.empty(sum(recv_sizes), dtype=torch.int64, device=dev)
    rc = torch.empty(sum(recv_sizes), dtype=torch.int64, device=dev)
    _all_to_all(rk, keys, recv_sizes, send_sizes, group=group)
    _all_to_all(rc, counts, recv_sizes, send_sizes, group=group)
    if return_sizes:
        return rk, rc, recv_sizes
    return rk, rc


def _gather_ints(values, group=None):
    """all_gather of a short list of integers: returns a [world][len(values)] list."""
    world = dist.get_world_size(group)
    dev = torch.device("cpu") if _host_staged(group) else torch.device("cuda", torch.cuda.current_device())
    mine = torch.tensor([int(v) for v in values], dtype=torch.int64, device=dev)
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine, group=group)
    return [[int(x) for x in v.tolist()] for v in allv]


class PeerFailure(RuntimeError):
    """Another rank of the collective failed; every rank leaves the merge from the same gather (KH_ERR_PEER of the C ABI)."""


def _checked_gather(values, err, group, where):
    """The gather every stretch of the merge ends in (exchange.hip, xp_gather): this rank's status in front of its
    values.  A rank that failed locally (err = its exception) reports here and re-raises; every other rank raises
    PeerFailure from the SAME call -- nobody walks on into an all-to-all a peer will never post."""
    vals = [0] * len(values) if err is not None else [int(v) for v in values]
    rows = _gather_ints([0 if err is None else 1] + vals, group)
    if err is not None:
        raise err
    bad = [r for r, row in enumerate(rows) if row[0]]
    if bad:
        raise PeerFailure(f"rank {bad[0]} failed before {where}; every rank leaves the merge")
    return [row[1:] for row in rows]


def _same_everywhere(value, group=None):
    """True iff every rank holds the same integer."""
    return all(v[0] == int(value) for v in _gather_ints([value], group))


def exchange_segments(buf, part_counts, group=None):
    """All-to-all of ONE owner-grouped buffer (any dtype).  Returns (received, per-sender sizes)."""
    world = dist.get_world_size(group)
    send_sizes = [int(x) for x in part_counts]
    assert len(send_sizes) == world and sum(send_sizes) == buf.numel()
    dev = buf.device
    ssz = torch.tensor(send_sizes, dtype=torch.int64, device=dev)
    rsz = torch.empty(world, dtype=torch.int64, device=dev)
    _all_to_all(rsz, ssz, group=group)
    recv_sizes = [int(x) for x in rsz.tolist()]
    out = torch.empty(sum(recv_sizes), dtype=buf.dtype, device=dev)
    _all_to_all(out, buf, recv_sizes, send_sizes, group=group)
    return out, recv_sizes


def _all_to_all_async(out, inp, out_splits=None, in_splits=None, group=None):
    """Starts the all-to-all and returns a work handle (None when it already completed: gloo, host staged).
    `inp` must be complete on the device (the export calls block until it is)."""
    if out.is_cuda and _host_staged(group):
        _all_to_all(out, inp, out_splits, in_splits, group=group)
        return None
    _refuse_huge_messages(out, inp, out_splits, in_splits, group)
    return dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group, async_op=True)


def _wait_all(works):
    """Host-side wait: the merge kernels run on the counter's own stream, not on torch's."""
    waited = False
    for w in works:
        if w is not None:
            w.wait()
            waited = True
    if waited and torch.cuda.is_available():
        torch.cuda.current_stream().synchronize()


def _stream_sync():
    if torch.cuda.is_available():
        torch.cuda.current_stream().synchronize()


def _device_sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def default_pieces():
    import os
    return int(os.environ.get("KMERHIP_MERGE_PIECES", "4"))


def _merge_pipelined(counter, group, agreed, npieces, exported, counts_all, keys, rcnt, nreg, n_local, dev, timing, t_last):
    """The heads / packed route as a pipeline over `npieces` equal shares of every owner's region range:
    [export i+1 | all-to-all i], then [merge i | all-to-all > i].  The caller has verified (collectively)
    that every piece fits the send buffer and holds piece 0's export in `exported`; it also restores the
    region window afterwards."""
    import os
    import time
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = nreg // world
    unit32 = agreed == 2
    ub = 4 if unit32 else 8
    sendbuf = keys.view(torch.int32) if unit32 else keys
    cap_total = sendbuf.numel() if unit32 else n_local
    wper = per // npieces
    t_exp = t_wait = t_merge = 0.0
    t0 = time.perf_counter()
    # every piece's sizes and every region's unit count, announced up front: ONE small exchange each,
    # so that nothing but the big all-to-alls sits on the communicator's stream afterwards
    send_mat = counts_all.view(world, npieces, wper).sum(dim=2, dtype=torch.int64).contiguous()  # [owner, piece]
    recv_mat = torch.empty_like(send_mat)                                                        # [sender, piece]
    _all_to_all(recv_mat, send_mat, group=group)
    rrc_full = torch.empty(nreg, dtype=torch.int32, device=dev)  # world slices: sender s's counts of MY regions
    _all_to_all(rrc_full, counts_all, group=group)
    send_h, recv_h = send_mat.cpu().numpy(), recv_mat.cpu().numpy()
    t1 = time.perf_counter()
    t_wait += t1 - t0
    t0 = t1
    flights, used, sent, parts = [], 0, 0, exported[0]
    # every receive buffer is allocated before the first transfer: running out of memory is something the ranks hear
    # about in the gather below, not a peer that never posts its receive
    err, bufs = None, []
    try:
        bufs = [torch.empty(int(recv_h[:, i].sum()), dtype=sendbuf.dtype, device=dev) for i in range(npieces)]
    except Exception as e:
        err = e
    _checked_gather([], err, group, "the first transfer")
    for i in range(npieces):
        err = buf = None
        send_sizes = recv_sizes = None
        try:
            if i > 0:
                counter.set_region_window(i, npieces)
                ptr = keys.data_ptr() + used * ub
                if unit32:
                    res = counter.export_regions_heads_device(world, ptr, cap_total - used, rcnt.data_ptr(), nreg)
                else:
                    res = counter.export_regions_packed_device(world, ptr, cap_total - used, rcnt.data_ptr(), nreg)
                if res is None:  # (cannot happen: the sizes were checked against the buffer before the first send)
                    raise RuntimeError("a later piece of the table does not fit the exchange format of the first")
                parts = res[0]
            send_sizes = [int(x) for x in parts.tolist()]
            if send_sizes != [int(x) for x in send_h[:, i]]:
                raise RuntimeError("piece sizes differ from the announced ones")
            recv_sizes = [int(x) for x in recv_h[:, i]]
            buf = sendbuf[used:used + sum(send_sizes)]
        except Exception as e:
            err = e
        # piece i leaves only when every rank has it ready
        _checked_gather([], err, group, f"the transfer of piece {i}")
        total = sum(send_sizes)
        used += total
        sent += total - send_sizes[rank]
        rp = bufs[i]
        t1 = time.perf_counter()
        t_exp += t1 - t0
        work = _all_to_all_async(rp, buf, recv_sizes, send_sizes, group=group)
        flights.append((rp, recv_sizes, work, buf))  # (buf stays alive while in flight)
        t0 = time.perf_counter()
        t_wait += t0 - t1
    n_recv, st2, err = 0, None, None
    try:
        counter.set_region_window(0, 1)
        counter.reset()
        counter.set_shard(rank, world)
        merge = counter.merge_regions_heads_device if unit32 else counter.merge_regions_packed_device
        rrc_v = rrc_full.view(world, npieces, wper)
        for i, (rp, recv_sizes, work, _) in enumerate(flights):
            t1 = time.perf_counter()
            rrc = torch.zeros_like(rrc_v)  # the senders' region counts as the merge of piece i wants them: zero elsewhere
            rrc[:, i, :] = rrc_v[:, i, :]
            _wait_all([work])
            _stream_sync()  # (rrc is written on torch's stream, read on the counter's; NOT a device-wide sync:
                            #  the later all-to-alls stay in flight)
            t2 = time.perf_counter()
            t_wait += t2 - t1
            offs = np.concatenate([[0], np.cumsum(recv_sizes)]).astype(np.int64)
            unit = rp.element_size()
            counter.set_region_window(i, npieces)
            merge(nreg, [rp.data_ptr() + unit * int(offs[s]) for s in range(world)],
                  [rrc.data_ptr() + 4 * per * s for s in range(world)])
            n_recv += rp.numel()
            t_merge += time.perf_counter() - t2
        counter.set_region_window(0, 1)
        st2 = counter.finish()
    except Exception as e:
        err = e
        _wait_all([f[2] for f in flights])  # (nothing may still write into buffers about to be dropped)
    _checked_gather([], err, group, "the end of the merge")  # "my merge failed" reaches every rank too
    if timing is not None:
        _device_sync()
        timing.update({"export": timing.get("export", 0.0) + t_exp * 1e3, "exchange_wait": t_wait * 1e3, "merge": t_merge * 1e3,
                       "pieces": npieces})
        t_last[0] = time.perf_counter()
        if os.environ.get("KMERHIP_MERGE_TIMING"):
            print("[merge timing ms]", {k: round(v, 2) for k, v in timing.items()}, flush=True)
    return {"path": ("regions-heads" if unit32 else "regions-packed") + f"-x{npieces}", "local_distinct": n_local,
            "sent_pairs": int(sent), "recv_pairs": int(n_recv), "owned_distinct": int(st2["distinct"]), "phase_ms": timing}


def merge_across_ranks(counter, group=None, packed=True, dense=True, phase_times=False, pieces=None):
    """Turns per-rank tables (each built from that rank's read shard) into a key-sharded global
    table: afterwards `counter` on rank r holds exactly the keys with kh_owner(key, k, world) == r,
    with counts summed over all ranks.  Returns a dict of sizes for reporting.

    Power-of-two world and equal table sizes: the region-ordered fast path (export in region order,
    one all-to-all of pairs + one of region counts, LDS rebuild of the shard: no scatter kernel, no
    global atomics).  The pairs travel in the narrowest unit every rank's table allows:
      "regions-heads"   u32 heads  = hash bits below the region index | addend - 1   (4 B per pair)
      "regions-packed"  u64        = count << 32 | 32 hash bits                      (8 B per pair)
      "regions"         u64 key + u64 count                                          (16 B per pair)
    -- xGMI is point to point, so at world 2 everything crosses ONE link: bytes are what matters.
    Otherwise: pairs grouped by owner and re-inserted with device atomics ("pairs").
    k <= 13 ("dense"): the 4^k key space as a dense count array, merged by ONE all-reduce(sum).

    pieces (default: KMERHIP_MERGE_PIECES or 4; 1 = off): the heads / packed routes run as a pipeline over
    `pieces` equal shares of every owner's region range (kh_set_region_window): the all-to-all of piece i
    is in flight (RCCL's own stream) while piece i + 1 is exported, and later while earlier pieces are
    merged -- export 6.8 ms + merge 13.4 ms at headline size are otherwise serial with the exchange.  The
    sizes of all pieces and all region counts are exchanged once, up front (kh_region_unit_counts_device)."""
    import os
    import time
    timing = {} if (phase_times or os.environ.get("KMERHIP_MERGE_TIMING")) else None
    t_last = [time.perf_counter()]

    def lap(name):  # phase wall times (with a device sync), only when asked for
        if timing is not None:
            _device_sync()
            now = time.perf_counter()
            timing[name] = timing.get(name, 0.0) + (now - t_last[0]) * 1e3
            t_last[0] = now

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    # Shape of every stretch below (as in exchange.hip): local, fallible steps run under `attempt`, which keeps the
    # first exception; the stretch ends in _checked_gather, where every rank learns about it and all leave together.
    err = [None]

    def attempt(fn, *a):
        if err[0] is not None:
            return None
        try:
            return fn(*a)
        except Exception as e:  # (reported at the next gather)
            err[0] = e
            return None

    def gather(values, where):
        e, err[0] = err[0], None
        return _checked_gather(values, e, group, where)

    st = attempt(counter.finish) or {"distinct": 0, "table_slots": 0}
    n_local = int(st["distinct"])
    nreg = int(st["table_slots"]) // 4096
    # (a counter may name the device its buffers live on: the CPU stand-in of tests/test_dist_gloo.py does)
    dev = getattr(counter, "torch_device", None) or torch.device("cuda", torch.cuda.current_device())
    if dense and 2 * counter.k <= 26:
        # Small k: the key space itself is a dense array of 4^k counts, and THAT is element-wise
        # reducible: one all-reduce(sum), any world size, then every rank keeps the keys it owns.
        n = 1 << (2 * counter.k)
        arr = attempt(lambda: torch.empty(n, dtype=torch.int64, device=dev))
        attempt(lambda: counter.export_dense_device(arr.data_ptr(), n))
        lap("export")
        gather([], "the dense all-reduce")
        if _host_staged(group):
            h = arr.cpu()
            dist.all_reduce(h, group=group)
            arr.copy_(h)
        else:
            dist.all_reduce(arr, group=group)
        _stream_sync()  # the collective runs on torch's stream, the merge on the counter's own
        lap("all_reduce")
        attempt(counter.reset)
        attempt(lambda: counter.merge_dense_device(arr.data_ptr(), n, rank, world))
        st2 = attempt(counter.finish)
        lap("merge")
        gather([], "the end of the merge")
        if timing is not None and os.environ.get("KMERHIP_MERGE_TIMING"):
            print("[merge timing ms]", {k: round(v, 2) for k, v in timing.items()}, flush=True)
        return {"path": "dense", "local_distinct": n_local, "sent_pairs": n, "recv_pairs": n,
                "owned_distinct": int(st2["distinct"]), "phase_ms": timing}
    keys = attempt(lambda: torch.empty(max(n_local, 1), dtype=torch.int64, device=dev))
    lap("setup")
    pow2 = world & (world - 1) == 0
    # (a table of 1024 x b2 regions, b2 not a power of two, nests with its hash-range shards only if the world divides b2:
    #  krust_amd/csrc/merge.hip merge_regions)
    b2 = nreg >> 10 if nreg > 1024 else 1
    geo_ok = b2 & (b2 - 1) == 0 or b2 % max(world, 1) == 0
    regions_ok = pow2 and world <= 64 and nreg >= world and nreg % world == 0 and geo_ok
    rcnt = attempt(lambda: torch.empty(nreg, dtype=torch.int32, device=dev)) if regions_ok else None

    def export(fmt):
        if fmt == 2:
            return counter.export_regions_heads_device(world, keys.data_ptr(), 2 * n_local, rcnt.data_ptr(), nreg)
        if fmt == 1:
            return counter.export_regions_packed_device(world, keys.data_ptr(), n_local, rcnt.data_ptr(), nreg)
        return None

    def export_narrowest():  # speculative: the export itself finds out whether the counts fit
        for fmt in (2, 1):
            ex = export(fmt)
            if ex is not None:
                return ex, fmt
        return None, 0

    npieces = default_pieces() if pieces is None else int(pieces)
    piped = (regions_ok and packed and npieces > 1 and npieces & (npieces - 1) == 0 and npieces <= 64
             and (nreg // world) >= 64 * npieces and (nreg // world) % npieces == 0)
    exported, my_fmt = None, 0
    if regions_ok and packed:
        if piped:
            attempt(counter.set_region_window, 0, npieces)
        exported, my_fmt = attempt(export_narrowest) or (None, 0)
    lap("export")
    votes = gather([nreg, my_fmt, int(piped)], "the format vote")
    # everything decided from here on is decided from the gathered words only: every rank decides the same
    same_size = all(v[0] == votes[0][0] for v in votes)
    agreed = min(v[1] for v in votes) if (regions_ok and same_size) else 0
    redo = False
    if piped and not (agreed and all(v[1] == agreed and v[2] for v in votes)):
        # some rank cannot run the pipeline, or not in the others' format: everybody takes the one-shot route
        piped, redo = False, True
    per = nreg // world if regions_ok else 0
    counts_all = None
    if piped:
        # Every piece's size is known before anything is sent (kh_region_unit_counts_device): a table that needs
        # more heads than the send buffer holds (counts far above 2^cb) is found out HERE, and all ranks leave
        # the pipeline together -- after the first all-to-all a rank that bails out would hang the others.
        unit32 = agreed == 2
        ub = 4 if unit32 else 8
        cap_total = 2 * n_local if unit32 else n_local

        def unit_counts():
            ca = torch.empty(nreg, dtype=torch.int32, device=dev)
            ok = counter.region_unit_counts_device(ub, ca.data_ptr(), nreg) == nreg
            if ok:
                _stream_sync()
                ok = int(ca.sum(dtype=torch.int64)) <= cap_total
            return ca, ok

        counts_all, fits = attempt(unit_counts) or (None, False)
        if not all(v[0] for v in gather([int(bool(fits))], "the piece sizes")):
            piped, redo = False, True
    if piped:
        try:
            return _merge_pipelined(counter, group, agreed, npieces, exported, counts_all, keys, rcnt, nreg, n_local, dev,
                                    timing, t_last)
        finally:
            counter.set_region_window(0, 1)  # whatever happened: later exports / merges cover the whole range again
    if redo:  # left the pipeline: whole table, narrowest unit that fits, and a second vote
        attempt(counter.set_region_window, 0, 1)
        exported, my_fmt = attempt(export_narrowest) or (None, 0)
        votes = gather([my_fmt], "the second format vote")
        agreed = min(v[0] for v in votes) if (regions_ok and same_size) else 0
    if agreed and agreed != my_fmt:  # another rank could not go as narrow: redo in the common format
        exported = attempt(export, agreed)
        if err[0] is None and exported is None:
            err[0] = RuntimeError("the table does not fit the exchange unit the ranks agreed on")
    if agreed:
        parts = exported[0] if exported is not None else np.zeros(world, dtype=np.uint64)
        total = int(parts.sum())
        buf = attempt(lambda: keys.view(torch.int32)[:total] if agreed == 2 else keys[:total])
        lap("vote")
        gather([], "the exchange")
        rp, recv_sizes = exchange_segments(buf, parts.tolist(), group=group)
        rrc = torch.empty(nreg, dtype=torch.int32, device=dev)  # world slices of nreg / world region counts
        _all_to_all(rrc, rcnt, group=group)
        _stream_sync()  # a non-async collective only orders torch's stream; the merge kernels run on the counter's
        lap("all_to_all")

        def merge_narrow():
            counter.reset()
            counter.set_shard(rank, world)
            offs = np.concatenate([[0], np.cumsum(recv_sizes)]).astype(np.int64)
            unit = rp.element_size()
            merge = counter.merge_regions_heads_device if agreed == 2 else counter.merge_regions_packed_device
            merge(nreg, [rp.data_ptr() + unit * int(offs[s]) for s in range(world)],
                  [rrc.data_ptr() + 4 * per * s for s in range(world)])

        attempt(merge_narrow)
        path, n_recv = ("regions-heads" if agreed == 2 else "regions-packed"), rp.numel()
        lap("merge")
    elif regions_ok and same_size:
        cnts = attempt(lambda: torch.empty(max(n_local, 1), dtype=torch.int64, device=dev))
        res = attempt(lambda: counter.export_regions_device(world, keys.data_ptr(), cnts.data_ptr(), n_local, rcnt.data_ptr(), nreg))
        parts = res[0] if res is not None else np.zeros(world, dtype=np.uint64)
        gather([], "the exchange")
        rk, rc, recv_sizes = exchange_pairs(keys[:n_local], cnts[:n_local], parts.tolist(), group=group, return_sizes=True)
        rrc = torch.empty(nreg, dtype=torch.int32, device=dev)
        _all_to_all(rrc, rcnt, group=group)
        _stream_sync()

        def merge_wide():
            counter.reset()
            counter.set_shard(rank, world)
            offs = np.concatenate([[0], np.cumsum(recv_sizes)]).astype(np.int64)
            counter.merge_regions_device(nreg,
                                         [rk.data_ptr() + 8 * int(offs[s]) for s in range(world)],
                                         [rc.data_ptr() + 8 * int(offs[s]) for s in range(world)],
                                         [rrc.data_ptr() + 4 * per * s for s in range(world)])

        attempt(merge_wide)
        path, n_recv = "regions", rk.numel()
    else:
        cnts = attempt(lambda: torch.empty(max(n_local, 1), dtype=torch.int64, device=dev))
        parts = attempt(lambda: counter.export_by_owner_device(world, keys.data_ptr(), cnts.data_ptr(), n_local))
        if parts is None:
            parts = np.zeros(world, dtype=np.uint64)
        gather([], "the exchange")
        rk, rc = exchange_pairs(keys[:n_local], cnts[:n_local], parts.tolist(), group=group)
        _stream_sync()

        def merge_pairs():
            counter.reset()
            counter.merge_pairs_device(rk.data_ptr(), rc.data_ptr(), rk.numel())

        attempt(merge_pairs)
        path, n_recv = "pairs", rk.numel()
    st2 = attempt(counter.finish)
    lap("finish")
    gather([], "the end of the merge")  # "my merge failed" reaches every rank too
    if timing is not None and os.environ.get("KMERHIP_MERGE_TIMING"):
        print("[merge timing ms]", {k: round(v, 2) for k, v in timing.items()}, flush=True)
    return {"path": path, "local_distinct": n_local, "sent_pairs": int(parts.sum() - parts[rank]),  # in exchange units
            "recv_pairs": int(n_recv), "owned_distinct": int(st2["distinct"]), "phase_ms": timing}


def group_pairs_by_owner(keys, counts, world, owner_fn):
    """Host-side twin of kh_export_by_owner_device for numpy pairs (used where the pairs are
    already on the host, e.g. gathering a final result).  owner_fn(key, world) -> shard."""
    keys = np.asarray(keys, dtype=np.uint64)
    counts = np.asarray(counts, dtype=np.uint64)
    own = np.fromiter((owner_fn(int(k), world) for k in keys), dtype=np.int64, count=keys.size)
    order = np.argsort(own, kind="stable")
    parts = np.bincount(own, minlength=world).astype(np.int64)
    return keys[order], counts[order], parts
