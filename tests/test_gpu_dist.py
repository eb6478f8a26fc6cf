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
m = O.OracleMap(); m.scan_flat(bases, 21, nthreads=4)
with krust_amd.DeviceCounter(21) as dc:
    dc.push(bases)
    info = merge_across_ranks(dc)
    keys, cnts = dc.result()
ok, oc = m.arrays()
assert np.array_equal(keys, ok) and np.array_equal(cnts, oc)
assert info["sent_pairs"] == 0 and info["recv_pairs"] == len(m) == info["owned_distinct"]
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
