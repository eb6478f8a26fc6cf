"""The exchange behind the C ABI (kh_comm_init / kh_merge_across / kh_group_*, krust_amd/csrc/exchange.hip.h):
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


def oracle_arrays(bases, k):
    m = O.OracleMap()
    m.scan_flat(bases, k, nthreads=4)
    return m.arrays()


# capacity_hint 3 M -> 2^11 regions.  k = 19: 27 hash bits below the region index -> u32 heads; k = 21: 31 bits ->
# packed u64; k = 31: neither -> key + count; k = 11: dense counts + all-reduce
@pytest.mark.parametrize("k,pieces,expect", [(19, "1", "regions-heads"), (19, None, "regions-heads-x4"), (19, "8", "regions-heads-x8"),
                                             (21, "1", "regions-packed"), (21, "2", "regions-packed-x2"), (31, None, "regions"),
                                             (11, None, "dense")])
def test_rccl_world1_through_the_c_abi(K, reads, monkeypatch, k, pieces, expect):
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
    owners = np.array([K.owner(int(x), k, world) for x in fk])
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
    owners = np.array([K.owner(int(x), k, world) for x in fk])
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
