#!/usr/bin/env python3
"""Wall-clock anatomy of one bench step with the multi-GPU merge leg, on ONE GPU at S100M size
(RCCL world 1: the all-to-all is a device-local copy, everything else is what an 8-GPU rank executes).
usage: python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/merge_step_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import krust_amd
from krust_amd import distributed as D

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
reads, rl = int(os.environ.get("READS", 100_000_000)), 150
tb = torch.empty(reads * (rl + 1), dtype=torch.uint8, device="cuda")
krust_amd.synth_reads_device(tb.data_ptr(), None, 20260130, 1 << 27, rl, 0, reads)
hint = int((1 << 27) * 1.05 + reads * 11.9)
dc = krust_amd.DeviceCounter(21, capacity_hint=hint, stream=torch.cuda.current_stream().cuda_stream)

def sync():
    torch.cuda.synchronize()

for it in range(3):
    sync(); t0 = time.perf_counter()
    dc.reset(); dc.push_device(tb.data_ptr(), None, tb.numel()); st = dc.finish()
    sync(); t1 = time.perf_counter()
    info = D.merge_across_ranks(dc)
    sync(); t2 = time.perf_counter()
    print(f"it {it}: count {1e3*(t1-t0):.1f} ms, merge leg {1e3*(t2-t1):.1f} ms ({info['path']}, {info['recv_pairs']} units), "
          f"step {1e3*(t2-t0):.1f} ms", flush=True)
dist.destroy_process_group()
