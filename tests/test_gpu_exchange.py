"""The exchange behind the C ABI (kh_comm_init / kh_merge_across / kh_group_*, krust_amd/csrc/exchange.hip):
north_star's "final RCCL reduce of per-GPU hash tables" reachable without Python or PyTorch.

On the 1-GPU box:
  * RCCL for real at world size 1 (ncclCommInitRank, ncclSend / ncclRecv groups to self, ncclAllGather,
    ncclAllReduce) through every route -- 32-bit heads, packed u64, 16-byte pairs, dense all-reduce -- one shot
    and pipelined in pieces;
  * N ranks as threads of one process sharing the device (kh_group with a device listed N times: RCCL refuses
    duplicate devices, so the transport is the process-local hub; the merge sequence, the votes, the piece
    pipeline and the LDS merges are the code an 8-GPU node runs).
Every shard is compared with the oracle's map restricted to the keys that rank owns -- the same tables
krust_amd/distributed.py (the torch.distributed harness) is held to in test_gpu_dist.py."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
SEED = 20260130


@pytest.fixture(scope="module")
def K():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU (run through gpurun)"
    import krust_amd
    krust_amd.lib()
    return krust_amd


@pytest.fixture(scope="module")
def reads():
    bases, _ = O.synth_reads(SEED, 1 << 18, 150, 0, 30000, with_qual=False)
    return bases


_ORACLE_CACHE = {}


def oracle_arrays(bases, k):
    """(keys, counts) of the oracle's map of `bases`; the same few read sets come back in dozens of cases: kept."""
    key = (bases.size, int(bases[:: max(1, bases.size // 4096)].astype(np.uint64).sum()), k)
    if key not in _ORACLE_CACHE:
        m = O.OracleMap()
        m.scan_flat(bases, k, nthreads=4)
        _ORACLE_CACHE[key] = m.arrays()
    return _ORACLE_CACHE[key]


# capacity_hint 3 M -> 2^11 regions.  k = 19: 27 hash bits below the region index -> u32 heads; k = 21: 31 bits ->
# packed u64; k = 31: neither -> key + count; k = 11: dense counts + all-reduce
@pytest.mark.parametrize("self_send", ["1", "0"], ids=["own-share-through-rccl", "own-share-read-in-place"])
@pytest.mark.parametrize("k,pieces,expect", [(19, "1", "regions-heads"), (19, None, "regions-heads-x4"), (19, "8", "regions-heads-x8"),
                                             (21, "1", "regions-packed"), (21, "2", "regions-packed-x2"), (31, None, "regions"),
                                             (11, None, "dense")])
def test_rccl_world1_through_the_c_abi(K, reads, monkeypatch, k, pieces, expect, self_send):
    # (round 6: a rank's own share of the region routes is read where the export put it; KMERHIP_SELF_SEND=1 sends it through the
    #  transport as rounds 2-5 did -- ncclSend / ncclRecv to self: the only data RCCL can carry on a one-GPU box)
    monkeypatch.setenv("KMERHIP_SELF_SEND", self_send)
    if pieces is None:
        monkeypatch.delenv("KMERHIP_MERGE_PIECES", raising=False)
    else:
        monkeypatch.setenv("KMERHIP_MERGE_PIECES", pieces)
    ok, oc = oracle_arrays(reads, k)
    with K.DeviceCounter(k, capacity_hint=3_000_000) as dc:
        dc.comm_init(1, 0, K.comm_unique_id())
        with pytest.raises(K.KmerHipError) as e:   # one communicator per context
            dc.comm_init(1, 0, K.comm_unique_id())
        assert e.value.status == K.native.KH_ERR_STATE
        for rep in range(2):  # twice: the second merge starts from a lazily reset (dirty) table
            dc.reset()
            dc.push(reads)
            info = dc.merge_across()
            assert info["path"] == expect and info["nranks"] == 1, info
            assert info["sent_units"] == (4 ** k if expect == "dense" else 0)   # (the dense route reduces all 4^k counts)
            assert info["owned_distinct"] == len(ok) == info["local_distinct"]
            keys, cnts = dc.result()
            assert np.array_equal(keys, ok) and np.array_equal(cnts, oc)
            info2 = dc.merge_across()                   # a world of one: merging again changes nothing
            assert info2["owned_distinct"] == len(ok) and dc.result_size() == len(ok)
        # an empty table merges to an empty shard
        dc.reset()
        info = dc.merge_across()
        assert info["owned_distinct"] == 0 and dc.result_size() == 0


def test_more_than_64_ranks_are_refused_before_any_collective(K):
    """every route keeps per-sender state in arrays of 64 entries; a larger world used to reach them."""
    with K.DeviceCounter(21) as dc:
        with pytest.raises(K.KmerHipError) as e:
            dc.comm_init(65, 0, K.comm_unique_id())
        assert e.value.status == K.native.KH_ERR_BAD_ARG and "64 ranks" in str(e.value)
        dc.comm_init(1, 0, K.comm_unique_id())   # (the context is still good for a communicator)


def test_merge_without_a_communicator_is_a_no_op(K, reads):
    ok, oc = oracle_arrays(reads, 21)
    with K.DeviceCounter(21) as dc:
        dc.push(reads)
        info = dc.merge_across()
        assert info["path"] == "none" and info["owned_distinct"] == len(ok)
        keys, cnts = dc.result()
        assert np.array_equal(keys, ok) and np.array_equal(cnts, oc)


@pytest.mark.parametrize("world,k,pieces,expect,path", [
    (2, 19, "1", "regions-heads", "partition"), (2, 19, None, "regions-heads-x4", "partition"), (4, 21, "1", "regions-packed", None),
    (4, 21, "4", "regions-packed-x4", None), (4, 17, "2", "regions-heads-x2", None), (8, 21, None, "regions-packed-x4", None),
    (2, 31, None, "regions", None), (3, 21, None, "pairs", None), (3, 13, None, "pairs", None), (2, 9, None, "regions-heads-x4", None)])
def test_group_of_ranks_sharing_the_device(K, monkeypatch, world, k, pieces, expect, path):
    """N ranks, one process, one host thread per rank inside the library (kh_group_merge); each rank counts
    ITS shard of the reads into its own device table first."""
    if pieces is None:
        monkeypatch.delenv("KMERHIP_MERGE_PIECES", raising=False)
    else:
        monkeypatch.setenv("KMERHIP_MERGE_PIECES", pieces)
    n_reads = 120_000
    full_b, _ = O.synth_reads(SEED, 1 << 20, 150, 0, n_reads, with_qual=False)
    fk, fc = oracle_arrays(full_b, k)
    owners = O.owners(K, fk, k, world)
    per = n_reads // world
    with K.DeviceGroup(k, [0] * world, capacity_hint=3_000_000, path=path) as g:
        assert len(g) == world
        for r, dc in enumerate(g.counters):
            lo, hi = r * per, (n_reads if r == world - 1 else (r + 1) * per)
            dc.push(full_b[lo * 151: hi * 151])
        locals_ = [dc.finish()["distinct"] for dc in g.counters]
        infos = g.merge()
        total_sent = 0
        for r, (dc, info) in enumerate(zip(g.counters, infos)):
            assert info["path"] == expect and info["nranks"] == world, info
            assert info["local_distinct"] == locals_[r]
            keys, cnts = dc.result()
            sel = owners == r
            assert np.array_equal(keys, fk[sel]) and np.array_equal(cnts, fc[sel]), f"shard {r} differs from the oracle"
            assert info["owned_distinct"] == int(sel.sum())
            assert np.array_equal(dc.lookup(keys[:1000]), cnts[:1000])      # lookups use the sharded placement
            # (round 6) heads / packed pairs build the shard as the 8-byte image; its lookups must not take another shard's key
            # for one of this shard's (the image names a key by the hash bits below the owner's)
            if expect.startswith("regions"):
                assert dc.finish()["slot_bytes"] == (16 if expect == "regions" else 8)
            foreign = fk[~sel][:2000]
            assert not dc.lookup(foreign).any()
            if expect.startswith("regions"):      # (the pairs route leaves an ordinary table that holds the rank's keys)
                with pytest.raises(K.KmerHipError) as e:                     # a hash-range shard refuses reads until reset,
                    dc.push(b"ACGTACGTACGTACGTACGTACGTACGTACGT\n")
                assert e.value.status == K.native.KH_ERR_STATE
                with pytest.raises(K.KmerHipError) as e:                     # and a second merge
                    dc.merge_across()
                assert e.value.status == K.native.KH_ERR_STATE
            total_sent += info["sent_units"]
        assert total_sent > 0
        # second round on the same group: reset, count again, merge again (dirty tables, same communicators)
        for r, dc in enumerate(g.counters):
            dc.reset()
            lo, hi = r * per, (n_reads if r == world - 1 else (r + 1) * per)
            dc.push(full_b[lo * 151: hi * 151])
        g.merge()
        got = 0
        for r, dc in enumerate(g.counters):
            keys, cnts = dc.result()
            assert np.array_equal(keys, fk[owners == r]) and np.array_equal(cnts, fc[owners == r])
            got += int(cnts.sum())
        assert got == int(fc.sum())


def test_group_with_large_counts_leaves_the_pipeline_together(K, monkeypatch):
    """Counts far above 2^cb need more 32-bit heads than the send buffer holds (> 2 per key): piece 0 alone would
    fit, so only the up-front size check can tell; every rank must then take the one-shot packed route."""
    monkeypatch.delenv("KMERHIP_MERGE_PIECES", raising=False)
    k, world = 19, 2
    rec = O.synth_reads(SEED, 1 << 12, 150, 0, 30, with_qual=False)[0]
    reps = 200                                   # every k-mer ~200 x 30 x 150 / 4096 ~ 200+ times: cb = 5 -> > 2 heads each
    bases = np.tile(rec, reps)
    fk, fc = oracle_arrays(bases, k)
    owners = O.owners(K, fk, k, world)
    with K.DeviceGroup(k, [0] * world, capacity_hint=3_000_000) as g:
        half = (reps // 2) * rec.size
        g[0].push(bases[:half])
        g[1].push(bases[half:])
        infos = g.merge()
        assert all(i["path"] in ("regions-packed", "regions-heads") and i["pieces"] == 1 for i in infos), infos
        assert infos[0]["path"] == infos[1]["path"]
        for r, dc in enumerate(g.counters):
            keys, cnts = dc.result()
            assert np.array_equal(keys, fk[owners == r]) and np.array_equal(cnts, fc[owners == r])


# ---- liveness (round 3): failure is collective, every wait is bounded ---------------------------------------------
def _merge_each_rank_on_its_own_thread(K, g, timeout=120):
    """kh_merge_across per context from Python threads (what kh_group_merge does inside the library), keeping every
    rank's own status.  Returns [(status, last_error)] per rank; fails the test if a rank is still inside after `timeout`."""
    import ctypes as C
    import threading
    L = K.lib()
    out = [None] * len(g)

    def run(i):
        info = K.native.KhMergeInfo()
        rc = L.kh_merge_across(g[i]._h, C.byref(info))
        out[i] = (rc, L.kh_last_error(g[i]._h).decode(), K.native.merge_info_dict(info))

    th = [threading.Thread(target=run, args=(i,), daemon=True) for i in range(len(g))]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout)
    assert all(not t.is_alive() for t in th), "a rank is still inside kh_merge_across: the collective hung"
    return out


def test_one_rank_needs_a_wider_unit_than_the_others(K, monkeypatch):
    """rank 0 holds a k-mer whose count needs more than 64 heads (> 64 << cb), rank 1 does not, so
    rank 0 votes `packed` and rank 1 `heads`.  The ranks must leave the pipeline TOGETHER (round 2 compared the agreed
    unit with each rank's own: the rank whose unit was the agreed one stayed in and the next collectives mismatched)."""
    monkeypatch.delenv("KMERHIP_MERGE_PIECES", raising=False)
    k, world, n_reads = 19, 2, 60_000
    full_b, _ = O.synth_reads(SEED, 1 << 20, 150, 0, n_reads, with_qual=False)
    full_b = full_b.copy()
    v = full_b.reshape(n_reads, 151)
    v[:40, :150] = ord("A")                       # rank 0's half: A^19 5280 times; cb = 5 at 2^11 regions -> 64 << 5 = 2048 is the most heads carry
    fk, fc = oracle_arrays(full_b, k)
    assert int(fc.max()) > 2048
    owners = O.owners(K, fk, k, world)
    per = n_reads // world
    with K.DeviceGroup(k, [0] * world, capacity_hint=3_000_000) as g:
        for r, dc in enumerate(g.counters):
            dc.push(full_b[r * per * 151: (r + 1) * per * 151])
        infos = g.merge()
        assert infos[0]["path"] == infos[1]["path"] and infos[0]["pieces"] == infos[1]["pieces"], infos
        assert infos[0]["path"] == "regions-packed", infos   # the common unit of a one-shot exchange
        for r, dc in enumerate(g.counters):
            keys, cnts = dc.result()
            assert np.array_equal(keys, fk[owners == r]) and np.array_equal(cnts, fc[owners == r]), f"shard {r}"


POINTS_PIPED = ["start", "export0", "unit_counts", "alloc_recv", "export_piece", "merge_piece"]
POINTS_ONESHOT = ["start", "export0", "oneshot_sizes", "oneshot_alloc", "oneshot_merge"]


@pytest.mark.parametrize("world,k,pieces,point,bad", [(2, 19, None, p, 1) for p in POINTS_PIPED] + [(4, 21, "1", p, 2) for p in POINTS_ONESHOT]
                         + [(4, 19, "2", "export_piece", 0), (3, 21, None, "generic_export", 1), (3, 21, None, "generic_alloc", 2),
                            (3, 21, None, "generic_merge", 0), (2, 31, None, "oneshot_alloc", 0)])
def test_a_rank_that_fails_takes_every_rank_out_of_the_merge(K, monkeypatch, world, k, pieces, point, bad):
    """a rank-local failure anywhere in the sequence (KMERHIP_FAULT injects one at a named point)
    makes EVERY rank return from the same kh_merge_across -- the failing rank with its own status, the others with
    KH_ERR_PEER naming it -- instead of leaving the peers in the next all-gather / receive for ever.  Afterwards the
    same group (reset, counted again) merges correctly."""
    if pieces is None:
        monkeypatch.delenv("KMERHIP_MERGE_PIECES", raising=False)
    else:
        monkeypatch.setenv("KMERHIP_MERGE_PIECES", pieces)
    monkeypatch.setenv("KMERHIP_MERGE_TIMEOUT_S", "60")
    n_reads = 40_000
    full_b, _ = O.synth_reads(SEED, 1 << 20, 150, 0, n_reads, with_qual=False)
    fk, fc = oracle_arrays(full_b, k)
    owners = O.owners(K, fk, k, world)
    per = n_reads // world

    def count_all(g):
        for r, dc in enumerate(g.counters):
            dc.reset()
            lo, hi = r * per, (n_reads if r == world - 1 else (r + 1) * per)
            dc.push(full_b[lo * 151: hi * 151])

    import time
    with K.DeviceGroup(k, [0] * world, capacity_hint=3_000_000) as g:
        count_all(g)
        monkeypatch.setenv("KMERHIP_FAULT", f"{bad}:{point}:{K.native.KH_ERR_STATE}")
        t0 = time.time()
        res = _merge_each_rank_on_its_own_thread(K, g)
        assert time.time() - t0 < 30, "the ranks left, but only after a time-out"
        for r, (rc, err, _) in enumerate(res):
            if r == bad:
                assert rc == K.native.KH_ERR_STATE and "injected fault" in err, (r, rc, err)
            else:
                assert rc == K.native.KH_ERR_PEER and f"rank {bad} failed" in err, (r, rc, err)
        monkeypatch.delenv("KMERHIP_FAULT")
        count_all(g)
        g.merge()
        for r, dc in enumerate(g.counters):
            keys, cnts = dc.result()
            assert np.array_equal(keys, fk[owners == r]) and np.array_equal(cnts, fc[owners == r]), f"shard {r} after the failed merge"


def test_a_rank_that_never_arrives_times_out(K, monkeypatch):
    """Only rank 0 of a two-rank group calls kh_merge_across: its first gather waits KMERHIP_MERGE_TIMEOUT_S and comes
    back with an error instead of blocking for ever."""
    import time
    monkeypatch.setenv("KMERHIP_MERGE_TIMEOUT_S", "2")
    bases, _ = O.synth_reads(SEED, 1 << 18, 150, 0, 5000, with_qual=False)
    with K.DeviceGroup(21, [0, 0], capacity_hint=1_000_000) as g:
        g[0].push(bases)
        g[1].push(bases)
        t0 = time.time()
        with pytest.raises(K.KmerHipError) as e:
            g[0].merge_across()
        assert 1.5 < time.time() - t0 < 20
        assert e.value.status == K.native.KH_ERR_PEER
        monkeypatch.setenv("KMERHIP_MERGE_TIMEOUT_S", "60")
        g[0].reset(); g[1].reset()
        g[0].push(bases); g[1].push(bases)
        infos = g.merge()                                  # kh_group_merge starts from a revived hub
        assert infos[0]["owned_distinct"] + infos[1]["owned_distinct"] == infos[0]["local_distinct"]


@pytest.mark.parametrize("point", ["start", "export0", "unit_counts", "alloc_recv", "merge_piece", "dense_export"])
def test_rccl_world1_local_failures_and_an_aborted_communicator(K, reads, monkeypatch, point):
    """The same protocol over RCCL (world 1: the gathers, send / recv groups and waits are RCCL's): a local failure is
    reported through the gather and leaves the communicator usable; an abort (what a time-out does) kills the
    communicator, not the process: the context says so, is destroyed cleanly, and a new one works."""
    monkeypatch.delenv("KMERHIP_MERGE_PIECES", raising=False)
    k = 11 if point == "dense_export" else 19
    ok, oc = oracle_arrays(reads, k)
    with K.DeviceCounter(k, capacity_hint=3_000_000) as dc:
        dc.comm_init(1, 0, K.comm_unique_id())
        dc.push(reads)
        monkeypatch.setenv("KMERHIP_FAULT", f"0:{point}:{K.native.KH_ERR_STATE}")
        with pytest.raises(K.KmerHipError) as e:
            dc.merge_across()
        assert e.value.status == K.native.KH_ERR_STATE and "injected fault" in str(e.value)
        monkeypatch.delenv("KMERHIP_FAULT")
        dc.reset()
        dc.push(reads)
        info = dc.merge_across()
        assert info["owned_distinct"] == len(ok)
        keys, cnts = dc.result()
        assert np.array_equal(keys, ok) and np.array_equal(cnts, oc)
        if point == "start":
            dc.reset()
            dc.push(reads)
            monkeypatch.setenv("KMERHIP_FAULT", "0:abort")
            with pytest.raises(K.KmerHipError) as e:
                dc.merge_across()
            assert e.value.status == K.native.KH_ERR_RCCL
            monkeypatch.delenv("KMERHIP_FAULT")
            with pytest.raises(K.KmerHipError) as e:      # the communicator is gone for good
                dc.merge_across()
            assert e.value.status == K.native.KH_ERR_RCCL and "aborted" in str(e.value)
            keys, cnts = dc.result()                       # ... the context and its table are not
            assert np.array_equal(keys, ok) and np.array_equal(cnts, oc)
    with K.DeviceCounter(k, capacity_hint=3_000_000) as dc:
        dc.comm_init(1, 0, K.comm_unique_id())
        dc.push(reads)
        assert dc.merge_across()["owned_distinct"] == len(ok)


def test_a_rank_whose_context_is_poisoned_joins_the_first_gather(K, monkeypatch):
    """kh_merge_across used to return straight away when enter() failed on a rank (a poisoned
    context; deferred counting that runs out of memory), leaving its peers in their first gather until the time-out.
    Now that rank reports in the gather: every rank leaves within seconds, the bad one with KH_ERR_STATE, the others
    with KH_ERR_PEER naming it."""
    import ctypes as C
    import time
    monkeypatch.setenv("KMERHIP_MERGE_TIMEOUT_S", "60")
    bases, _ = O.synth_reads(SEED, 1 << 18, 150, 0, 6000, with_qual=False)
    L = K.lib()
    with K.DeviceGroup(21, [0, 0, 0], capacity_hint=1_000_000) as g:
        for dc in g.counters:
            dc.push(bases)
        # poison rank 1: an 8 TB device allocation (kh_lookup of 2^40 keys; fails before a single key is read)
        one = (C.c_uint64 * 1)(0)
        rc = L.kh_lookup(g[1]._h, one, C.c_uint64(1 << 40), one)
        assert rc == K.native.KH_ERR_OOM
        t0 = time.time()
        res = _merge_each_rank_on_its_own_thread(K, g)
        assert time.time() - t0 < 30, "the ranks left, but only after a time-out"
        assert res[1][0] == K.native.KH_ERR_STATE and "poisoned" in res[1][1], res[1]
        for r in (0, 2):
            assert res[r][0] == K.native.KH_ERR_PEER and "rank 1 failed" in res[r][1], res[r]


@pytest.mark.parametrize("world,k,b2,pieces", [(2, 21, 40, None), (4, 19, 24, "2"), (8, 21, 40, None), (2, 31, 6, None), (4, 21, 100, "1")])
def test_group_merge_of_tables_with_1024_x_b2_regions(K, monkeypatch, world, k, b2, pieces):
    """Round 4: tables of 1024 x b2 regions, b2 not a power of two (what a large input gets: 640 at 125 M reads).  Ownership is
    still the top bits of the hash, a rank's share of the sender regions still nests in the receiver's table when the world
    divides b2 (merge.hip merge_regions), and the exchange units carry x - xlo(bucket) instead of a bit field
    (kernels.hip.h kh_below_region).  KMERHIP_TABLE_REGIONS forces the geometry on every rank; every shard against the
    oracle, through the heads / packed / wide routes, one shot and in pieces."""
    monkeypatch.setenv("KMERHIP_TABLE_REGIONS", str(1024 * b2))
    if pieces is None:
        monkeypatch.delenv("KMERHIP_MERGE_PIECES", raising=False)
    else:
        monkeypatch.setenv("KMERHIP_MERGE_PIECES", pieces)
    n_reads = 120_000
    full_b, _ = O.synth_reads(SEED + b2, 1 << 20, 150, 0, n_reads, with_qual=False)
    fk, fc = oracle_arrays(full_b, k)
    owners = O.owners(K, fk, k, world)
    per = n_reads // world
    with K.DeviceGroup(k, [0] * world, capacity_hint=3_000_000, path="partition" if b2 >= 40 else None) as g:
        for r, dc in enumerate(g.counters):
            lo, hi = r * per, (n_reads if r == world - 1 else (r + 1) * per)
            dc.push(full_b[lo * 151: hi * 151])
        assert all(dc.finish()["table_slots"] == 1024 * b2 * 4096 for dc in g.counters)
        infos = g.merge()
        assert all(i["path"].startswith("regions") for i in infos), infos   # b2 is a multiple of the world: the shards nest
        for r, (dc, info) in enumerate(zip(g.counters, infos)):
            keys, cnts = dc.result()
            sel = owners == r
            assert np.array_equal(keys, fk[sel]) and np.array_equal(cnts, fc[sel]), f"shard {r} differs from the oracle"
            assert info["owned_distinct"] == int(sel.sum())
            assert np.array_equal(dc.lookup(keys[:1000]), cnts[:1000])


def test_a_world_that_does_not_divide_b2_takes_the_generic_route(K, monkeypatch):
    """1024 x 6 regions among 4 ranks: 6 / 4 is not an integer, a shard's regions do not nest -- every rank sees that from the
    same numbers and the merge takes the owner-partitioned route (correct, slower)."""
    monkeypatch.setenv("KMERHIP_TABLE_REGIONS", str(1024 * 6))
    k, world, n_reads = 21, 4, 40_000
    full_b, _ = O.synth_reads(SEED, 1 << 20, 150, 0, n_reads, with_qual=False)
    fk, fc = oracle_arrays(full_b, k)
    owners = O.owners(K, fk, k, world)
    per = n_reads // world
    with K.DeviceGroup(k, [0] * world, capacity_hint=3_000_000) as g:
        for r, dc in enumerate(g.counters):
            dc.push(full_b[r * per * 151: (r + 1) * per * 151])
        infos = g.merge()
        assert all(i["path"] == "pairs" for i in infos), infos
        for r, dc in enumerate(g.counters):
            keys, cnts = dc.result()
            assert np.array_equal(keys, fk[owners == r]) and np.array_equal(cnts, fc[owners == r])


@pytest.mark.parametrize("world,k", [(2, 21), (4, 31)])
def test_ranks_with_tables_of_different_sizes_agree_on_the_largest(K, world, k):
    """Tables are sized from each rank's own input (growth, or the level-1 sample): ranks can arrive at the merge with
    different geometries, whose regions do not correspond.  The merge's first gather is the table sizes; the smaller tables
    are re-laid-out to the largest (exchange.hip, "one table geometry for every rank") and the region routes apply.  Here
    rank 0 has eight times the reads of the others and no rank has a hint."""
    n_reads = 110_000
    full_b, _ = O.synth_reads(SEED + 5, 1 << 22, 150, 0, n_reads, with_qual=False)
    fk, fc = oracle_arrays(full_b, k)
    owners = O.owners(K, fk, k, world)
    small = 10_000
    cuts = [0, n_reads - small * (world - 1)] + [n_reads - small * (world - 1 - i) for i in range(1, world)]
    with K.DeviceGroup(k, [0] * world) as g:
        for r, dc in enumerate(g.counters):
            dc.push(full_b[cuts[r] * 151: cuts[r + 1] * 151])
        slots = [dc.finish()["table_slots"] for dc in g.counters]
        assert slots[0] > slots[1], slots
        infos = g.merge()
        assert all(i["path"].startswith("regions") for i in infos), infos
        for r, dc in enumerate(g.counters):
            keys, cnts = dc.result()
            sel = owners == r
            assert np.array_equal(keys, fk[sel]) and np.array_equal(cnts, fc[sel]), f"shard {r} differs from the oracle"


# ---- conservation (round 5): the merge checks itself --------------------------------------------------------------
@pytest.mark.parametrize("world,k,pieces,expect", [(2, 19, None, "regions-heads-x4"), (4, 21, "1", "regions-packed"), (2, 31, None, "regions"),
                                                   (3, 21, None, "pairs")])
def test_the_merge_reports_what_it_conserved(K, monkeypatch, world, k, pieces, expect):
    """kh_merge_info (round 5): nranks_seen = what the transport itself reports, conserved = 1, and the counts that left the
    ranks (sum of every exported unit's count, from the senders' digests) equal the counts the merge kernels put into the
    shards, equal the oracle's k-mer total."""
    if pieces is None:
        monkeypatch.delenv("KMERHIP_MERGE_PIECES", raising=False)
    else:
        monkeypatch.setenv("KMERHIP_MERGE_PIECES", pieces)
    n_reads = 60_000
    full_b, _ = O.synth_reads(SEED, 1 << 20, 150, 0, n_reads, with_qual=False)
    fk, fc = oracle_arrays(full_b, k)
    per = n_reads // world
    with K.DeviceGroup(k, [0] * world, capacity_hint=3_000_000) as g:
        for r, dc in enumerate(g.counters):
            lo, hi = r * per, (n_reads if r == world - 1 else (r + 1) * per)
            dc.push(full_b[lo * 151: hi * 151])
        locals_ = [dc.finish()["kmers"] for dc in g.counters]
        infos = g.merge()
        assert all(i["path"] == expect and i["nranks_seen"] == world and i["conserved"] == 1 for i in infos), infos
        assert [i["sent_count_sum"] for i in infos] == locals_          # every rank exported exactly what its table held
        assert sum(i["merged_count_sum"] for i in infos) == int(fc.sum()) == sum(locals_)
        for dc, i in zip(g.counters, infos):
            st = dc.finish()
            assert st["kmers"] == i["merged_count_sum"] == int(dc.result()[1].sum())   # a shard's kmers = the occurrences it holds


@pytest.mark.parametrize("world,k,pieces,bad", [(2, 19, None, 1), (4, 21, "1", 2), (2, 31, None, 0), (3, 21, None, 1)])
def test_half_a_message_lost_fails_the_merge_on_every_rank(K, monkeypatch, world, k, pieces, bad):
    """round 4's transport incident -- a message of >= 2^30 bytes arrived HALF, and the merged table was
    quietly smaller.  KMERHIP_FAULT=rank:drop_half zeroes the upper half of what arrived on `rank` (pipelined heads, one-shot
    packed, wide pairs, the generic route): that rank's digest of the arrival differs from what the senders announced, it
    returns KH_ERR_RCCL naming the sender, every other rank returns KH_ERR_PEER from the same call -- nobody keeps a table that
    lost keys.  Afterwards the same group merges correctly."""
    if pieces is None:
        monkeypatch.delenv("KMERHIP_MERGE_PIECES", raising=False)
    else:
        monkeypatch.setenv("KMERHIP_MERGE_PIECES", pieces)
    monkeypatch.setenv("KMERHIP_MERGE_TIMEOUT_S", "60")
    n_reads = 40_000
    full_b, _ = O.synth_reads(SEED, 1 << 20, 150, 0, n_reads, with_qual=False)
    fk, fc = oracle_arrays(full_b, k)
    owners = O.owners(K, fk, k, world)
    per = n_reads // world

    def count_all(g):
        for r, dc in enumerate(g.counters):
            dc.reset()
            lo, hi = r * per, (n_reads if r == world - 1 else (r + 1) * per)
            dc.push(full_b[lo * 151: hi * 151])

    with K.DeviceGroup(k, [0] * world, capacity_hint=3_000_000) as g:
        count_all(g)
        monkeypatch.setenv("KMERHIP_FAULT", f"{bad}:drop_half")
        res = _merge_each_rank_on_its_own_thread(K, g)
        for r, (rc, err, info) in enumerate(res):
            if r == bad:
                assert rc == K.native.KH_ERR_RCCL and "conservation" in err and "differs from what it sent" in err, (r, rc, err)
            else:
                assert rc == K.native.KH_ERR_PEER and f"rank {bad} failed" in err, (r, rc, err)
            assert info["conserved"] == 0
        monkeypatch.delenv("KMERHIP_FAULT")
        count_all(g)
        infos = g.merge()
        assert all(i["conserved"] == 1 for i in infos)
        for r, dc in enumerate(g.counters):
            keys, cnts = dc.result()
            assert np.array_equal(keys, fk[owners == r]) and np.array_equal(cnts, fc[owners == r]), f"shard {r} after the failed merge"


def test_rccl_world1_detects_a_lost_half_too(K, reads, monkeypatch):
    """The same through RCCL itself (a world of one: the send to self that lost half a message in round 4)."""
    monkeypatch.delenv("KMERHIP_MERGE_PIECES", raising=False)
    monkeypatch.setenv("KMERHIP_SELF_SEND", "1")   # (round 6: by default a rank's own share does not travel -- here it must, there is nothing else)
    ok, oc = oracle_arrays(reads, 19)
    with K.DeviceCounter(19, capacity_hint=3_000_000) as dc:
        dc.comm_init(1, 0, K.comm_unique_id())
        dc.push(reads)
        monkeypatch.setenv("KMERHIP_FAULT", "0:drop_half")
        with pytest.raises(K.KmerHipError) as e:
            dc.merge_across()
        assert e.value.status == K.native.KH_ERR_RCCL and "conservation" in str(e.value)
        monkeypatch.delenv("KMERHIP_FAULT")
        dc.reset()
        dc.push(reads)
        info = dc.merge_across()
        assert info["conserved"] == 1 and info["nranks_seen"] == 1 and info["owned_distinct"] == len(ok)
        assert info["sent_count_sum"] == info["merged_count_sum"] == int(oc.sum())
        keys, cnts = dc.result()
        assert np.array_equal(keys, ok) and np.array_equal(cnts, oc)


def test_small_count_then_merge_into_a_larger_geometry_then_a_large_partitioned_push(K, monkeypatch):
    """the merge raised the context's region capacity after growing only the three per-region arrays it
    uses itself; a later partitioned batch into the same context (little counted -> merged into the world's larger geometry ->
    kh_reset -> much counted) then skipped the allocation of the region pass's other arrays.  One helper grows them all now
    (kmerhip.hip ensure_region_scratch)."""
    monkeypatch.delenv("KMERHIP_MERGE_PIECES", raising=False)
    k, world = 21, 2
    big_b, _ = O.synth_reads(SEED + 9, 1 << 22, 150, 0, 400_000, with_qual=False)
    small_b = big_b[: 151 * 2_000]
    with K.DeviceGroup(k, [0] * world) as g:
        g[0].push(big_b)                   # rank 0: a partitioned batch, a table of several thousand regions
        g[1].push(small_b)                 # rank 1: the direct path, a small table, no region scratch of its own
        slots = [dc.finish()["table_slots"] for dc in g.counters]
        assert slots[0] > slots[1]
        infos = g.merge()                  # rank 1 is re-laid-out to rank 0's geometry and merges into it
        assert all(i["conserved"] == 1 for i in infos)
        g[0].reset()
        g[1].reset()
        g[1].push(big_b)                   # ... and now counts a large partitioned batch itself
        st = g[1].finish()
        m = O.OracleMap()
        m.scan_flat(big_b, k, nthreads=4)
        ok, oc = m.arrays()
        assert st["part_batches"] >= 1 and st["distinct"] == len(ok)
        keys, cnts = g[1].result()
        assert np.array_equal(keys, ok) and np.array_equal(cnts, oc)


@pytest.mark.parametrize("world,k,expect", [(2, 21, "regions"), (3, 21, "pairs")])
def test_merge_borrows_the_idle_partition_buffers_and_gives_them_back(K, monkeypatch, world, k, expect):
    """Round 5: a merge takes its scratch -- and the shard's 16-byte table -- out of the partition buffers the count has just
    finished with, instead of handing ~200 GB back to the driver and taking them again at the next count (bench.py
    --force-merge at configs[3]'s size: 514 -> 120 ms per count-and-merge step).  Here, at test size: partitioned counts (so
    the buffers exist), a merge (its table lives in them: KMERHIP_TRACE says so), the shards against the oracle, then
    (a) reset + count + merge again on the same group -- the loan ends with the reset -- and (b) on the route that leaves a
    table one can push into (pairs), a partitioned push AFTER the merge: the borrowed table has to move out of the buffers
    the push is about to overwrite (end_borrow), with its counts."""
    monkeypatch.delenv("KMERHIP_MERGE_PIECES", raising=False)
    n_reads = 240_000
    full_b, _ = O.synth_reads(SEED + 11, 1 << 20, 150, 0, n_reads, with_qual=False)
    fk, fc = oracle_arrays(full_b, k)
    owners = O.owners(K, fk, k, world)
    per = n_reads // world
    extra = full_b[: 151 * 60_000]
    ek, ec = oracle_arrays(extra, k)
    with K.DeviceGroup(k, [0] * world, capacity_hint=600_000, path="partition") as g:   # (a small table beside large batches: it fits the buffers)
        for rnd in range(2):
            for r, dc in enumerate(g.counters):
                dc.reset()
                lo, hi = r * per, (n_reads if r == world - 1 else (r + 1) * per)
                dc.push(full_b[lo * 151: hi * 151])
            assert all(dc.finish()["part_batches"] >= 1 for dc in g.counters)
            infos = g.merge()
            assert all(i["path"].startswith(expect) and i["conserved"] == 1 for i in infos), infos
            for r, dc in enumerate(g.counters):
                keys, cnts = dc.result()
                assert np.array_equal(keys, fk[owners == r]) and np.array_equal(cnts, fc[owners == r]), f"round {rnd}, shard {r}"
        if expect == "pairs":    # (a hash-range shard refuses pushes until reset; the owner-partitioned route leaves an ordinary table)
            dc = g[0]
            dc.push(extra)
            st = dc.finish()
            want = dict(zip(fk[owners == 0].tolist(), fc[owners == 0].tolist()))
            for a, b in zip(ek.tolist(), ec.tolist()):
                want[a] = want.get(a, 0) + b
            keys, cnts = dc.result()
            assert st["distinct"] == len(want) and dict(zip(keys.tolist(), cnts.tolist())) == want
