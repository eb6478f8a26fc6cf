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


def test_shard_range_partition():
    from krust_amd.distributed import shard_range
    for n in (0, 1, 7, 100, 10**9 + 7):
        for w in (1, 2, 3, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1
