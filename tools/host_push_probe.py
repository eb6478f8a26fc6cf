#!/usr/bin/env python3
"""PCIe-inclusive rate of kh_push (host buffers -> pinned staging -> HBM -> count)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import krust_amd
reads, rl = int(os.environ.get("READS", 10_000_000)), 150
tb = torch.empty(reads * (rl + 1), dtype=torch.uint8, device="cuda")
krust_amd.synth_reads_device(tb.data_ptr(), None, 20260130, 1 << 27, rl, 0, reads)
host = tb.cpu().numpy()
for path in (None, "direct"):
    with krust_amd.DeviceCounter(21, capacity_hint=int(3e8), path=path) as dc:
        for it in range(2):
            dc.reset()
            t0 = time.perf_counter(); dc.push(host); st = dc.finish(); dt = time.perf_counter() - t0
        print(f"path={path}: kh_push {host.size/1e9:.2f} GB in {dt*1e3:.0f} ms = {host.size/dt/1e9:.2f} GB/s, {st['kmers']/dt/1e9:.2f} G k-mers/s; h2d_ms={st['h2d_ms']:.0f} kernel_ms={st['count_kernel_ms']:.0f} batches={st['part_batches']}")
    with krust_amd.DeviceCounter(21, capacity_hint=int(3e8), path=path) as dc:
        dc.reset(); t0 = time.perf_counter(); dc.push_device(tb.data_ptr(), None, tb.numel()); st = dc.finish(); dt = time.perf_counter() - t0
        print(f"          device-resident: {dt*1e3:.0f} ms = {st['kmers']/dt/1e9:.2f} G k-mers/s")
