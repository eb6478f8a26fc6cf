#!/usr/bin/env python3
"""One skewed case of tools/skew_probe.py (10 % homopolymer / (AC)n reads), partitioned path only: for profiling."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import krust_amd
reads, rl = 10_000_000, 150
stride = rl + 1
tb = torch.empty(reads * stride, dtype=torch.uint8, device="cuda")
krust_amd.synth_reads_device(tb.data_ptr(), None, 20260130, 1 << 27, rl, 0, reads)
torch.cuda.synchronize()
v = tb.view(reads, stride)
n = reads // 10
v[:n, :rl] = ord("A")
v[n // 2:n, 1:rl:2] = ord("C")
torch.cuda.synchronize()
with krust_amd.DeviceCounter(21, capacity_hint=300_000_000, path="partition") as dc:
    for it in range(2):
        dc.reset()
        t0 = time.perf_counter()
        dc.push_device(tb.data_ptr(), None, tb.numel())
        st = dc.finish()
        dt = time.perf_counter() - t0
    print(f"{dt*1e3:8.1f} ms distinct={st['distinct']} stages={ {k: round(x,1) for k,x in st['stage_ms'].items() if x>0.05} }")
