#!/usr/bin/env python3
"""Robustness probe: how the two insert paths behave on heavily skewed input (a fraction of the
reads replaced by homopolymer / short-period repeats, as in real genomes' poly-A and satellites)."""
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
for frac, label in ((0.0, "uniform"), (0.1, "10% poly-A"), (0.5, "50% poly-A + (AC)n")):
    if frac:
        n = int(reads * frac)
        v[:n, :rl] = ord("A")
        v[n // 2:n, 1:rl:2] = ord("C")
    torch.cuda.synchronize()
    for path in ("partition", "direct"):
        with krust_amd.DeviceCounter(21, capacity_hint=300_000_000, path=path) as dc:
            for it in range(2):
                dc.reset()
                t0 = time.perf_counter()
                dc.push_device(tb.data_ptr(), None, tb.numel())
                st = dc.finish()
                dt = time.perf_counter() - t0
            top = max(c for c, f in dc.histogram()[-3:])
            print(f"{label:22s} {path:9s} {dt*1e3:8.1f} ms  {st['kmers']/dt/1e9:6.1f} G/s  distinct={st['distinct']} max_count={top} stages={ {k: round(x,1) for k,x in st['stage_ms'].items() if x>0.05} }")
