#!/usr/bin/env python3
"""kh_result_copy of S100M (1.09 G pairs, 17.4 GB) into pinned arrays: wall time and rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, krust_amd as K
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
tb = torch.empty(reads * 151, dtype=torch.uint8, device="cuda")
K.synth_reads_device(tb.data_ptr(), None, 20260130, 1 << 30, 150, 0, reads)
torch.cuda.synchronize()
with K.DeviceCounter(21, capacity_hint=int(1.09e9 * reads / 1e8)) as dc:
    dc.push_device(tb.data_ptr(), None, tb.numel()); st = dc.finish()
    n = dc.result_size()
    with K.PinnedArray(n, np.uint64) as pk, K.PinnedArray(n, np.uint64) as pc:
        for rep in range(3):
            t0 = time.perf_counter(); k, c = dc.result(sort=False, out=(pk.array, pc.array)); dt = time.perf_counter() - t0
            print(f"pinned result copy {rep}: {n} pairs in {dt:.3f} s = {16 * n / dt / 1e9:.1f} GB/s; sum of counts {int(c.sum())} vs kmers {st['kmers']}")
    t0 = time.perf_counter(); k, c = dc.result(sort=False); dt = time.perf_counter() - t0
    print(f"pageable result copy: {dt:.3f} s = {16 * n / dt / 1e9:.1f} GB/s")
