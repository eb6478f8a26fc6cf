#!/usr/bin/env python3
"""Cost of a region-window export (kh_set_region_window) against the whole-range export, S100M table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import krust_amd

reads, rl, N = int(os.environ.get("READS", 100_000_000)), 150, int(os.environ.get("NPARTS", 8))
tb = torch.empty(reads * (rl + 1), dtype=torch.uint8, device="cuda")
krust_amd.synth_reads_device(tb.data_ptr(), None, 20260130, 1 << 27, rl, 0, reads)
hint = int((1 << 27) * 1.05 + reads * 11.9)
dc = krust_amd.DeviceCounter(21, capacity_hint=hint)
dc.push_device(tb.data_ptr(), None, tb.numel())
st = dc.finish()
del tb
n, R = st["distinct"], st["table_slots"] // 4096
keys = torch.empty(n, dtype=torch.int64, device="cuda")
rc = torch.empty(R, dtype=torch.int32, device="cuda")

def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3

for rep in range(2):
    _, t = timed(lambda: dc.export_regions_heads_device(N, keys.data_ptr(), 2 * n, rc.data_ptr(), R))
    print(f"whole range: {t:.2f} ms")
    for npieces in (2, 4, 8):
        ts = []
        for piece in range(npieces):
            dc.set_region_window(piece, npieces)
            _, t = timed(lambda: dc.export_regions_heads_device(N, keys.data_ptr(), 2 * n, rc.data_ptr(), R))
            ts.append(t)
        dc.set_region_window(0, 1)
        print(f"{npieces} pieces: " + " ".join(f"{t:.2f}" for t in ts) + f"  sum {sum(ts):.2f} ms")
