#!/usr/bin/env python3
"""PCIe rates on the box: one pinned copy at a time vs two side by side on two streams, each direction."""
import time, torch
n = 4 << 30
d = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(2)]
h = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(2)]
s = [torch.cuda.Stream() for _ in range(2)]
def run(pairs, label):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for (dst, src, st) in pairs:
        with torch.cuda.stream(st):
            dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label}: {len(pairs) * n / dt / 1e9:.1f} GB/s")
for rep in range(2):
    run([(h[0], d[0], s[0])], "D2H one stream ")
    run([(h[0], d[0], s[0]), (h[1], d[1], s[1])], "D2H two streams")
    run([(h[0][: n // 2], d[0][: n // 2], s[0]), (h[0][n // 2:], d[0][n // 2:], s[1])], "D2H one buffer, halves on two streams")
    run([(d[0], h[0], s[0])], "H2D one stream ")
    run([(d[0], h[0], s[0]), (d[1], h[1], s[1])], "H2D two streams")
    run([(d[0], h[0], s[0]), (h[1], d[1], s[1])], "H2D + D2H together")
