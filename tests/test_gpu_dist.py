"""RCCL leg on one GPU: torch.distributed 'nccl' (= RCCL) at world size 1 driving
krust_amd.distributed.merge_across_ranks end to end (export by owner -> all_to_all_single ->
reset -> merge).  Multi-rank behaviour is covered on CPU by test_dist_gloo.py and as logical
shards by test_gpu_parity.py; this checks that the real collective path runs on the device."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import oracle_lib as O
import krust_amd
from krust_amd.distributed import merge_across_ranks
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
bases, _ = O.synth_reads(20260130, 1 << 18, 150, 0, 30000, with_qual=False)
# capacity_hint 3 M -> 2^11 regions.  k = 19: 27 hash bits below the region index -> u32 heads;
# k = 21: 31 bits -> the 64-bit packed unit; k = 31: neither -> key + count.
# pieces: the heads / packed routes as a pipeline of async all-to-alls over shares of the region range
for k, expect, pieces in ((19, "regions-heads", 1), (19, "regions-heads-x4", None), (19, "regions-heads-x8", 8), (21, "regions-packed", 1),
                          (21, "regions-packed-x2", 2), (31, "regions", None), (11, "dense", None)):
    m = O.OracleMap(); m.scan_flat(bases, k, nthreads=4)
    with krust_amd.DeviceCounter(k, capacity_hint=3_000_000) as dc:
        for rep in range(2):  # twice: the second merge starts from a lazily reset (dirty) table
            dc.reset()
            dc.push(bases)
            info = merge_across_ranks(dc, pieces=pieces, phase_times=bool(rep))
            keys, cnts = dc.result()
            ok, oc = m.arrays()
            assert np.array_equal(keys, ok) and np.array_equal(cnts, oc)
            assert (expect == "dense" or info["sent_pairs"] == 0) and info["recv_pairs"] >= len(m) == info["owned_distinct"]
            assert info["path"] == expect, info
with krust_amd.DeviceCounter(19, capacity_hint=3_000_000) as dc:  # nothing counted: every piece is empty
    info = merge_across_ranks(dc)
    assert info["owned_distinct"] == 0 and dc.result_size() == 0, info
    dc.reset()
    dc.push(bases[:151 * 100])  # and a table whose pieces are nearly empty
    want = O.OracleMap(); want.scan_flat(bases[:151 * 100], 19, nthreads=1)
    info = merge_across_ranks(dc)
    keys, cnts = dc.result()
    ok, oc = want.arrays()
    assert np.array_equal(keys, ok) and np.array_equal(cnts, oc), info
print("NCCL_OK", len(m))
dist.destroy_process_group()
'''


def test_rccl_world1_merge(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {ROOT!r}\n" + WORKER)
    port = 29700 + (os.getpid() % 200)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "NCCL_OK" in out.stdout


# Several ranks sharing the one GPU of the box (RCCL refuses duplicate devices, so the collective is
# gloo with host staging, krust_amd/distributed.py::_all_to_all): every rank counts ITS shard of the
# reads into a real device table, then the real export -> exchange -> set_shard -> LDS merge runs.
MULTI = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import oracle_lib as O
import krust_amd
from krust_amd.distributed import merge_across_ranks, shard_range
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
N_READS, SEED = 120000, 20260130
lo, hi = shard_range(N_READS, rank, world)
bases, _ = O.synth_reads(SEED, 1 << 20, 150, lo, hi - lo, with_qual=False)
full_b, _ = O.synth_reads(SEED, 1 << 20, 150, 0, N_READS, with_qual=False)
for PATH, K, EXPECT_PATH, PIECES in CASES:     # (one launch per world size: a torchrun start-up costs more than a case)
    with krust_amd.DeviceCounter(K, capacity_hint=3_000_000, path=PATH) as dc:
        dc.push(bases)
        info = merge_across_ranks(dc, pieces=PIECES)
        keys, cnts = dc.result()
        assert info["path"] == EXPECT_PATH, info
        # the shard answers lookups for its own keys and stays refusing reads until reset
        if len(keys):
            assert np.array_equal(dc.lookup(keys[:1000]), cnts[:1000])
    full = O.OracleMap(); full.scan_flat(full_b, K, nthreads=2)
    fk, fc = full.arrays()
    assert all(krust_amd.owner(int(k), K, world) == rank for k in keys[:20000])
    gathered = [None] * world
    dist.all_gather_object(gathered, (keys, cnts))
    if rank == 0:
        uk = np.concatenate([g[0] for g in gathered]); uc = np.concatenate([g[1] for g in gathered])
        order = np.argsort(uk, kind="stable")
        assert np.array_equal(uk[order], fk) and np.array_equal(uc[order], fc), (K, EXPECT_PATH)   # disjoint shards whose union is the oracle's map
        print("MULTI_OK", world, K, EXPECT_PATH, len(uk))
dist.destroy_process_group()
'''

# (path, k, expected route, pieces) per world size
MULTI_CASES = {2: [("partition", 19, "regions-heads", 1), ("partition", 19, "regions-heads-x4", None), (None, 31, "regions", None)],
               4: [(None, 21, "regions-packed", 1), (None, 21, "regions-packed-x4", 4), (None, 17, "regions-heads-x2", 2)],
               3: [(None, 21, "pairs", None), (None, 13, "dense", None)]}


@pytest.mark.parametrize("world", [2, 4, 3])
def test_ranks_sharing_one_gpu_merge_real_tables(world, tmp_path):
    cases = MULTI_CASES[world]
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {ROOT!r}\nCASES = {cases!r}\n" + MULTI)
    port = 29500 + world + (os.getpid() % 100)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("MULTI_OK") == len(cases), out.stdout[-2000:]


def test_bench_multi_rank_code_path_on_one_gpu():
    """bench.py --gpus 2 end to end (weak-scaled shards, merge inside the timed step, max-over-ranks
    timing, conservation check) with both ranks on the box's one GPU: BENCH_BACKEND=gloo stages the
    collectives through the host, everything else is the code the 8-GPU run executes."""
    import json
    port = 29400 + (os.getpid() % 100)
    env = dict(os.environ, BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "2000000", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert len(line) <= 6144
    mg = d["config"]["merge"]
    assert mg["conserved"] and mg["path"].startswith("regions"), mg
    assert mg["merged_occurrences"] == 2 * d["config"]["kmers_per_step_per_gpu"] or mg["conserved"]
    assert d["config"]["single_gpu_same_share"]["value"] > 0       # the N = 1 point of the SAME workload, measured in this run


@pytest.mark.parametrize("n", [2, 4])
def test_bench_group_mode_runs_the_multi_rank_accounting_through_the_c_abi(n):
    """RCCL refuses two ranks on one device, so the bench's C-ABI merge branch cannot run under torchrun on
    a 1-GPU box.  `bench.py --group N` drives N thread ranks through kh_group_create / kh_group_merge (the in-process hub) and
    fills the same config.merge / per_rank / conserved / rccl_nranks fields: the accounting code of the 8-GPU line runs here."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--group", str(n), "--steps", "2", "--warmup", "1", "--reads", "1000000"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, BENCH_FULL_PATH=f"/tmp/bench_group{n}.json"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) <= 6144          # ONE line, and one the driver's 8 KB tail holds whole
    d = json.loads(lines[0])
    mg = d["config"]["merge"]
    assert mg["rccl_nranks"] == n and mg["conserved"] is True and mg["lib_conserved"] is True, mg
    assert mg["path"].startswith("regions") and mg["merged_occurrences"] == n * d["config"]["kmers_per_step_per_gpu"] or mg["conserved"]
    assert len(d["config"]["per_rank"]) == n and d["config"]["per_rank_columns"][0] == "rank"
    assert d["verify"]["ok"] is True and d["config"]["mode"] == f"group{n}"
    full = json.load(open(f"/tmp/bench_group{n}.json"))
    assert full["config"]["merge"]["sent_count_sum"] > 0 and len(full["config"]["per_rank"]) == n


def test_bench_force_merge_drives_the_rccl_merge_on_one_gpu():
    """`bench.py --force-merge`: the N > 1 step -- count, then kh_merge_across over a REAL RCCL communicator (of one rank) inside
    the timed region -- with every key of the N > 1 line (config.merge from the library, rccl_nranks from ncclCommCount,
    conservation, per_rank, single_gpu_same_share): what a torchrun launch on 8 GPUs executes, on the box's one."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-merge", "--steps", "2", "--warmup", "1", "--reads", "1500000", "--verify"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, BENCH_FULL_PATH="/tmp/bench_force_merge.json"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) <= 6144
    d = json.loads(lines[0])
    mg = d["config"]["merge"]
    assert mg["impl"].startswith("kh_merge_across") and mg["rccl_nranks"] == 1 and mg["conserved"] is True and mg["lib_conserved"] is True, mg
    assert mg["merged_occurrences"] == d["config"]["kmers_per_step_per_gpu"] and mg["path"].startswith("regions")
    assert d["config"]["single_gpu_same_share"]["value"] > 0 and len(d["config"]["per_rank"]) == 1
    assert d["verify"]["ok"] is True
