"""N>1 path on CPU: world_size-2 (and 3) gloo run of the read sharding + key-partitioned
all-to-all exchange in krust_amd/distributed.py.  The per-rank counting is done by the oracle
here (tests may use it as a stand-in; the product path uses the HIP table) -- what is under
test is the host logic of the exchange: shard ranges, owner grouping, split sizes, that every
key lands on exactly one owner and the union equals the single-process result."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import oracle_lib as O
import krust_amd
from krust_amd.distributed import shard_range, exchange_pairs, group_pairs_by_owner

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
N_READS, K, SEED = 6000, 21, 20260130
lo, hi = shard_range(N_READS, rank, world)
bases, _ = O.synth_reads(SEED, 1 << 15, 150, lo, hi - lo, with_qual=False)
m = O.OracleMap(); m.scan_flat(bases, K)
keys, cnts = m.arrays()
gk, gc, parts = group_pairs_by_owner(keys, cnts, world, lambda key, w: krust_amd.owner(key, K, w))
rk, rc = exchange_pairs(torch.from_numpy(gk.view(np.int64).copy()), torch.from_numpy(gc.view(np.int64).copy()), parts.tolist())
owned = O.OracleMap()
for k_, c_ in zip(rk.numpy().view(np.uint64).tolist(), rc.numpy().view(np.uint64).tolist()):
    assert krust_amd.owner(k_, K, world) == rank
    owned.add(k_, c_)
ok, oc = owned.arrays()
gathered = [None] * world
dist.all_gather_object(gathered, (ok.tolist(), oc.tolist()))
if rank == 0:
    full_b, _ = O.synth_reads(SEED, 1 << 15, 150, 0, N_READS, with_qual=False)
    full = O.OracleMap(); full.scan_flat(full_b, K)
    union = {}
    for ks, cs in gathered:
        for k_, c_ in zip(ks, cs):
            assert k_ not in union
            union[k_] = c_
    assert union == full.as_dict(), "sharded+merged result differs from single-process result"
    print("DIST_OK", world, len(union), sum(union.values()))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_sharded_merge(world, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {ROOT!r}\n" + WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    port = 29600 + world + (os.getpid() % 200)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert f"DIST_OK {world}" in out.stdout


# merge_across_ranks itself (format vote, region-ordered export, the pipeline over pieces of the region
# range with all sizes announced up front, async handles, shard merge) with a CPU stand-in for the
# device table: same calls, same exchange units (32-bit heads, region order, masked region counts),
# numpy instead of kernels.  What is under test is the host sequence in krust_amd/distributed.py.
STANDIN = r'''
import os, sys, ctypes
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import oracle_lib as O
import krust_amd
from krust_amd.distributed import merge_across_ranks, shard_range

FC = (0x9E3779B1, 0x85EBCA77, 0xC2B2AE3D, 0x27D4EB2F)
def ff(r, c, k, i):
    if i % 2:  # rounds 2 and 4 (kmer_bits.h kh_feistel_g): bits k .. 2k-1 of the full product
        return ((r * c) >> k) & ((1 << k) - 1)
    t = (r * ((c & 0xFFFFFF) | 1)) & 0xFFFFFFFF if 16 <= k <= 24 else (r * c) & 0xFFFFFFFF
    return t >> (32 - k) if k < 32 else t
def table_hash(key, k):
    mask = (1 << k) - 1
    L, R = (key >> k) & mask, key & mask
    for i, c in enumerate(FC):
        L, R = R, (L ^ ff(R, c, k, i)) & mask
    return (L << k) | R
def table_unhash(h, k):
    mask = (1 << k) - 1
    L, R = (h >> k) & mask, h & mask
    for i, c in reversed(list(enumerate(FC))):
        L, R = (R ^ ff(L, c, k, i)) & mask, L
    return (L << k) | R

def view(ptr, n, dtype):
    return np.ctypeslib.as_array(ctypes.cast(int(ptr), ctypes.POINTER(ctypes.c_uint8)), shape=(n * np.dtype(dtype).itemsize,)).view(dtype)

class StandIn:
    """The calls of krust_amd.DeviceCounter that merge_across_ranks makes, on host memory."""
    torch_device = torch.device("cpu")
    def __init__(self, k, rbits):
        self.k, self.rbits, self.table, self.shard, self.win = k, rbits, {}, None, (0, 1)
        self.hb = 2 * k - rbits; self.cb = 32 - self.hb
        assert 1 <= self.hb <= 28
    def finish(self):
        return {"distinct": len(self.table), "table_slots": 4096 << self.rbits}
    def reset(self):
        self.table, self.shard = {}, None
    def set_shard(self, index, count):
        assert not self.table
        self.shard = (index, count)
    def set_region_window(self, piece=0, npieces=1):
        self.win = (piece, npieces)
    def _heads_by_region(self):
        regs = [[] for _ in range(1 << self.rbits)]
        for key, cnt in self.table.items():
            h = table_hash(key, self.k)
            r, low = h >> self.hb, h & ((1 << self.hb) - 1)
            while cnt > 0:  # a large count travels as several heads
                take = min(cnt, 1 << self.cb)
                regs[r].append((low << self.cb) | (take - 1))
                cnt -= take
        return regs
    def region_unit_counts_device(self, unit_bytes, rc_ptr, region_cap):
        assert unit_bytes == 4 and self.shard is None
        view(rc_ptr, 1 << self.rbits, np.uint32)[:] = [len(r) for r in self._heads_by_region()]
        return 1 << self.rbits
    def export_regions_heads_device(self, nparts, ptr, cap, rc_ptr, region_cap):
        nreg = 1 << self.rbits
        per = nreg // nparts
        piece, npieces = self.win
        wper = per // npieces
        regs = self._heads_by_region()
        rc = view(rc_ptr, nreg, np.uint32)
        out, parts = [], np.zeros(nparts, dtype=np.uint64)
        for r in range(nreg):
            inside = (r % per) // wper == piece
            rc[r] = len(regs[r]) if inside else 0
            if inside:
                out += regs[r]
                parts[r // per] += len(regs[r])
        if len(out) > cap:
            return None  # KH_ERR_RANGE: more heads than the caller's buffer holds
        view(ptr, max(len(out), 1), np.uint32)[:len(out)] = out
        return parts, nreg
    allow_packed = False
    def export_regions_packed_device(self, nparts, ptr, cap, rc_ptr, region_cap):
        assert self.allow_packed, "heads are representable: the packed route must not be tried"
        nreg = 1 << self.rbits
        per = nreg // nparts
        assert self.win == (0, 1)
        regs = [[] for _ in range(nreg)]
        for key, cnt in self.table.items():
            h = table_hash(key, self.k)
            regs[h >> self.hb].append((cnt << 32) | ((h & ((1 << self.hb) - 1)) << (32 - self.hb)))
        rc = view(rc_ptr, nreg, np.uint32)
        out, parts = [], np.zeros(nparts, dtype=np.uint64)
        for r in range(nreg):
            rc[r] = len(regs[r]); out += regs[r]; parts[r // per] += len(regs[r])
        assert len(out) <= cap
        view(ptr, max(len(out), 1), np.uint64)[:len(out)] = out
        return parts, nreg
    def merge_regions_packed_device(self, sender_regions, ptrs, rc_ptrs):
        index, count = self.shard
        nr = sender_regions // count
        for ptr, rcp in zip(ptrs, rc_ptrs):
            rc = view(rcp, nr, np.uint32)
            offs = np.concatenate([[0], np.cumsum(rc.astype(np.int64))])
            units = view(ptr, max(int(offs[-1]), 1), np.uint64)
            for j in range(nr):
                for u in units[int(offs[j]):int(offs[j + 1])].tolist():
                    h = ((index * nr + j) << self.hb) | ((u & 0xFFFFFFFF) >> (32 - self.hb))
                    key = table_unhash(h, self.k)
                    self.table[key] = self.table.get(key, 0) + (u >> 32)
    def merge_regions_heads_device(self, sender_regions, ptrs, rc_ptrs):
        index, count = self.shard
        nr = sender_regions // count
        piece, npieces = self.win
        for ptr, rcp in zip(ptrs, rc_ptrs):
            rc = view(rcp, nr, np.uint32)
            offs = np.concatenate([[0], np.cumsum(rc.astype(np.int64))])
            assert all(rc[j] == 0 for j in range(nr) if j // (nr // npieces) != piece)
            units = view(ptr, max(int(offs[-1]), 1), np.uint32)
            for j in range(nr):
                for u in units[int(offs[j]):int(offs[j + 1])].tolist():
                    h = ((index * nr + j) << self.hb) | (u >> self.cb)
                    key = table_unhash(h, self.k)
                    self.table[key] = self.table.get(key, 0) + (u & ((1 << self.cb) - 1)) + 1

import datetime
dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
rank, world = dist.get_rank(), dist.get_world_size()
N_READS, K, SEED, RBITS = 4000, 19, 20260130, 11
lo, hi = shard_range(N_READS, rank, world)
bases, _ = O.synth_reads(SEED, 1 << 14, 150, lo, hi - lo, with_qual=False)
m = O.OracleMap(); m.scan_flat(bases, K)
full_b, _ = O.synth_reads(SEED, 1 << 14, 150, 0, N_READS, with_qual=False)
full = O.OracleMap(); full.scan_flat(full_b, K)
for pieces, path in ((1, "regions-heads"), (None, "regions-heads-x4"), (2, "regions-heads-x2")):
    c = StandIn(K, RBITS)
    c.table = dict(m.as_dict())
    c.table[next(iter(c.table))] += 5000  # one count that needs several heads
    bumped = next(iter(c.table))
    info = merge_across_ranks(c, pieces=pieces, phase_times=True)
    assert info["path"] == path, info
    assert all(krust_amd.owner(key, K, world) == rank for key in c.table)
    gathered = [None] * world
    dist.all_gather_object(gathered, (c.table, bumped))
    if rank == 0:
        union, want = {}, dict(full.as_dict())
        for tab, b in gathered:
            assert not (set(tab) & set(union))
            union.update(tab)
            want[b] += 5000
        assert union == want, "merged shards differ from the single-process result"
# A table whose counts need more heads than the send buffer holds (> 2 per key on average): piece 0 alone
# would fit, so only the up-front size check can tell -- every rank must then leave the pipeline TOGETHER and
# take the one-shot packed route (a rank that bailed out after the first all-to-all would hang the others),
# and the region window must be back to (0, 1) afterwards.
c = StandIn(K, RBITS)
c.allow_packed = True
c.table = {key: cnt + 100 for key, cnt in m.as_dict().items()} if rank == 0 else dict(m.as_dict())
info = merge_across_ranks(c, pieces=None, phase_times=False)
assert info["path"] == "regions-packed", info
assert c.win == (0, 1)
gathered = [None] * world
dist.all_gather_object(gathered, (c.table, len(m) if rank == 0 else 0))
if rank == 0:
    union = {}
    for tab, _ in gathered:
        assert not (set(tab) & set(union))
        union.update(tab)
    assert sum(union.values()) == full.total() + 100 * len(m) and set(union) == set(full.as_dict())
sys.stdout.write(f"PIPE_OK {world}\n"); sys.stdout.flush()
# ---- liveness: a rank that fails on its own takes EVERY rank out of the same merge, nobody hangs ----
# (VERDICT r2 next-1c.)  One rank's stand-in raises at a named call -- the export of a later piece (one transfer
# already in flight), the up-front unit counts, the merge of a piece, the very first finish, the one-shot export --
# and every rank must come out of merge_across_ranks with an exception: the failing one with its own, the others
# with PeerFailure naming it.  gloo's collective time-out (60 s) would turn a hang into a failure of this script.
from krust_amd.distributed import PeerFailure
import time
class Boom(RuntimeError):
    pass
class Faulty(StandIn):
    def __init__(self, k, rbits, fail_at, nth=1):
        super().__init__(k, rbits)
        self.fail_at, self.nth, self.seen = fail_at, nth, 0
    def _maybe(self, name):
        if name == self.fail_at:
            self.seen += 1
            if self.seen == self.nth:
                raise Boom(f"injected failure in {name}")
for name in ("finish", "region_unit_counts_device", "export_regions_heads_device", "merge_regions_heads_device", "reset", "set_shard"):
    def wrap(name=name, inner=getattr(StandIn, name)):
        def f(self, *a, **kw):
            self._maybe(name)
            return inner(self, *a, **kw)
        return f
    setattr(Faulty, name, wrap())
bad = world - 1
for pieces, fail_at, nth in ((None, "export_regions_heads_device", 2), (None, "region_unit_counts_device", 1), (None, "merge_regions_heads_device", 2),
                             (None, "finish", 1), (1, "export_regions_heads_device", 1), (1, "set_shard", 1), (2, "reset", 1)):
    c = Faulty(K, RBITS, fail_at if rank == bad else None, nth)
    c.table = dict(m.as_dict())
    t0 = time.time()
    try:
        merge_across_ranks(c, pieces=pieces)
        raise SystemExit(f"rank {rank}: the merge returned although rank {bad} failed in {fail_at}")
    except Boom as e:
        assert rank == bad, (rank, e)
    except PeerFailure as e:
        assert rank != bad and f"rank {bad} failed" in str(e), (rank, e)
    assert time.time() - t0 < 30
    dist.barrier()   # every rank is out, and the group still works:
    c = StandIn(K, RBITS)
    c.table = dict(m.as_dict())
    info = merge_across_ranks(c, pieces=pieces)
    assert all(krust_amd.owner(key, K, world) == rank for key in c.table) and c.win == (0, 1)
sys.stdout.write(f"FAULT_OK {world}\n"); sys.stdout.flush()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_merge_across_ranks_pipeline_with_cpu_standin(world, tmp_path):
    script = tmp_path / "standin.py"
    script.write_text(f"ROOT = {ROOT!r}\n" + STANDIN)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    port = 29650 + world + (os.getpid() % 200)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count(f"PIPE_OK {world}") == world
    assert out.stdout.count(f"FAULT_OK {world}") == world   # the failure-injection leg: every rank out of every failed merge


def test_shard_range_partition():
    from krust_amd.distributed import shard_range
    for n in (0, 1, 7, 100, 10**9 + 7):
        for w in (1, 2, 3, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1
