#!/usr/bin/env python3
"""kh_push of S100M from pinned memory (kh_host_alloc) -> kh_finish, wall time; KMERHIP_ACC_MAX_MB varies the accumulation buffers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, krust_amd as K
reads = int(os.environ.get("READS", 100_000_000))
tb = torch.empty(reads * 151, dtype=torch.uint8, device="cuda")
K.synth_reads_device(tb.data_ptr(), None, 20260130, 1 << 27, 150, 0, reads)
torch.cuda.synchronize()
with K.PinnedArray(tb.numel()) as pin:
    pin.array[:] = tb.cpu().numpy()
    for mb in sys.argv[1:] or ["0"]:
        if mb == "0": os.environ.pop("KMERHIP_ACC_MAX_MB", None)
        else: os.environ["KMERHIP_ACC_MAX_MB"] = mb
        with K.DeviceCounter(21, capacity_hint=int((1 << 27) * 1.05 + reads * 9.3)) as dc:
            for it in range(2):
                dc.reset()
                t0 = time.perf_counter(); dc.push(pin.array); st = dc.finish(); dt = time.perf_counter() - t0
            print(f"acc_max_mb={mb}: {pin.array.size/1e9:.1f} GB in {dt*1e3:.0f} ms = {pin.array.size/dt/1e9:.1f} GB/s (h2d {st['h2d_ms']:.0f} ms, kernels {st['count_kernel_ms']:.0f} ms, {st['part_batches']} batches)", flush=True)
