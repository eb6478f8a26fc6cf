#!/usr/bin/env python3
"""Timing experiment: level-1 scatter with its write-out (1) or its staging + write-out (2) skipped.
Step 1 runs normally so that the pool holds valid payloads for the later stages."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, krust_amd
reads, rl = 100_000_000, 150
tb = torch.empty(reads * (rl + 1), dtype=torch.uint8, device="cuda")
krust_amd.synth_reads_device(tb.data_ptr(), None, 20260130, 1 << 27, rl, 0, reads)
dc = krust_amd.DeviceCounter(21, capacity_hint=int((1 << 27) * 1.05 + reads * 11.9))
for dbg in ("0", "0", "1", "2", "0"):
    os.environ["KMERHIP_DEBUG"] = dbg
    dc.reset(); dc.push_device(tb.data_ptr(), None, tb.numel()); st = dc.finish()
    print("dbg", dbg, {k: round(v, 1) for k, v in st["stage_ms"].items() if v > 0.5})
