#!/usr/bin/env python3
"""End-to-end rates with HOST buffers on both sides (never the bench `value`): kh_push from pageable
memory (pinned staging -> HBM -> count), then kh_result_copy of every (key, count) pair back to host
arrays, and the device-resident rate beside them.  READS=40000000 python tools/host_push_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import krust_amd
reads, rl = int(os.environ.get("READS", 10_000_000)), 150
tb = torch.empty(reads * (rl + 1), dtype=torch.uint8, device="cuda")
krust_amd.synth_reads_device(tb.data_ptr(), None, 20260130, 1 << 27, rl, 0, reads)
host = tb.cpu().numpy()
hint = int((1 << 27) * 1.05 + reads * 11.9)
with krust_amd.DeviceCounter(21, capacity_hint=hint) as dc:
    for it in range(2):
        dc.reset()
        t0 = time.perf_counter(); dc.push(host); st = dc.finish(); t1 = time.perf_counter()
        keys, counts = dc.result(sort=False); t2 = time.perf_counter()
        dt, dr = t1 - t0, t2 - t1
        print(f"it {it}: kh_push {host.size/1e9:.2f} GB in {dt*1e3:.0f} ms = {host.size/dt/1e9:.2f} GB/s, {st['kmers']/dt/1e9:.2f} G k-mers/s "
              f"(h2d {st['h2d_ms']:.0f} ms, kernels {st['count_kernel_ms']:.0f} ms, {st['part_batches']} batches); "
              f"kh_result_copy {keys.size} pairs ({keys.size*16/1e9:.2f} GB) in {dr*1e3:.0f} ms = {keys.size*16/dr/1e9:.2f} GB/s; "
              f"end to end {st['kmers']/(dt+dr)/1e9:.2f} G k-mers/s", flush=True)
    hist_t0 = time.perf_counter(); h = dc.histogram(); hist_dt = time.perf_counter() - hist_t0
    print(f"kh_histogram (device, {len(h)} lines): {hist_dt*1e3:.1f} ms")
    dc.reset(); t0 = time.perf_counter(); dc.push_device(tb.data_ptr(), None, tb.numel()); st = dc.finish(); dt = time.perf_counter() - t0
    print(f"device-resident: {dt*1e3:.0f} ms = {st['kmers']/dt/1e9:.2f} G k-mers/s")
