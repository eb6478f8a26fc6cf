#!/usr/bin/env python3
"""Times the multi-GPU merge legs on ONE GPU at headline size (S100M table): region-ordered export,
and the LDS shard merge fed by 8 logical senders (the same export stands in for every sender, so
the read volume and the 8x segment re-read match an 8-rank weak-scaling run; counts come out x8).
Also times the generic pairs path (export by owner + atomic merge) for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import krust_amd

reads, rl, N = int(os.environ.get("READS", 100_000_000)), 150, 8
tb = torch.empty(reads * (rl + 1), dtype=torch.uint8, device="cuda")
krust_amd.synth_reads_device(tb.data_ptr(), None, 20260130, 1 << 27, rl, 0, reads)
hint = int((1 << 27) * 1.05 + reads * 11.9)
dc = krust_amd.DeviceCounter(21, capacity_hint=hint)
dc.push_device(tb.data_ptr(), None, tb.numel())
st = dc.finish()
del tb
torch.cuda.empty_cache()
n, R = st["distinct"], st["table_slots"] // 4096
keys = torch.empty(n, dtype=torch.int64, device="cuda")
cnts = torch.empty(n, dtype=torch.int64, device="cuda")
rc = torch.empty(R, dtype=torch.int32, device="cuda")

def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3

(parts, _), t_exp = timed(lambda: dc.export_regions_device(N, keys.data_ptr(), cnts.data_ptr(), n, rc.data_ptr(), R))
_, t_exp2 = timed(lambda: dc.export_regions_device(N, keys.data_ptr(), cnts.data_ptr(), n, rc.data_ptr(), R))
offs = np.concatenate([[0], np.cumsum(parts)]).astype(np.int64)
print(f"table: {n} distinct, {R} regions; export_regions: {t_exp:.1f} / {t_exp2:.1f} ms ({n*16/1e9:.1f} GB of pairs)")
sh = krust_amd.DeviceCounter(21, capacity_hint=hint)
per = R // N
for rep in range(2):
    sh.reset(); sh.set_shard(0, N)
    _, t_m = timed(lambda: sh.merge_regions_device(R, [keys.data_ptr() + 8 * int(offs[0])] * N, [cnts.data_ptr() + 8 * int(offs[0])] * N,
                                                   [rc.data_ptr()] * N))
    s2 = sh.finish()
    print(f"merge_regions (8 senders x {int(parts[0])} pairs, 8x re-read): {t_m:.1f} ms -> {s2['distinct']} distinct")
# packed form (one u64 per pair)
(res), t_pe = timed(lambda: dc.export_regions_packed_device(N, keys.data_ptr(), n, rc.data_ptr(), R))
(res), t_pe2 = timed(lambda: dc.export_regions_packed_device(N, keys.data_ptr(), n, rc.data_ptr(), R))
if res is None:
    print("packed export: not representable")
else:
    pparts, _ = res
    poffs = np.concatenate([[0], np.cumsum(pparts)]).astype(np.int64)
    print(f"export_regions_packed: {t_pe:.1f} / {t_pe2:.1f} ms ({n*8/1e9:.1f} GB of pairs)")
    for rep in range(2):
        sh.reset(); sh.set_shard(0, N)
        _, t_m = timed(lambda: sh.merge_regions_packed_device(R, [keys.data_ptr() + 8 * int(poffs[0])] * N, [rc.data_ptr()] * N))
        s2 = sh.finish()
        print(f"merge_regions_packed (8 senders x {int(pparts[0])} pairs): {t_m:.1f} ms -> {s2['distinct']} distinct")
    # 32-bit heads
    (res), t_he = timed(lambda: dc.export_regions_heads_device(N, keys.data_ptr(), 2 * n, rc.data_ptr(), R))
    (res), t_he2 = timed(lambda: dc.export_regions_heads_device(N, keys.data_ptr(), 2 * n, rc.data_ptr(), R))
    if res is None:
        print("heads export: not representable")
    else:
        hparts, _ = res
        hoffs = np.concatenate([[0], np.cumsum(hparts)]).astype(np.int64)
        print(f"export_regions_heads: {t_he:.1f} / {t_he2:.1f} ms ({int(hparts.sum())} heads for {n} pairs, {int(hparts.sum())*4/1e9:.1f} GB)")
        for rep in range(2):
            sh.reset(); sh.set_shard(0, N)
            _, t_m = timed(lambda: sh.merge_regions_heads_device(R, [keys.data_ptr() + 4 * int(hoffs[0])] * N, [rc.data_ptr()] * N))
            s2 = sh.finish()
            print(f"merge_regions_heads (8 senders x {int(hparts[0])} heads): {t_m:.1f} ms -> {s2['distinct']} distinct")
    if os.environ.get("SKIP_GENERIC"):
        sys.exit(0)
    # restore the wide export for the generic comparison below
# generic path for comparison
(parts2), t_e3 = timed(lambda: dc.export_by_owner_device(N, keys.data_ptr(), cnts.data_ptr(), n))
sh.reset()
_, t_m2 = timed(lambda: (sh.merge_pairs_device(keys.data_ptr(), cnts.data_ptr(), n), sh.finish()))
print(f"generic: export_by_owner {t_e3:.1f} ms; atomic merge of {n} pairs {t_m2:.1f} ms")
_, t_r = timed(lambda: (sh.reset(), sh.finish()))
print(f"reset: {t_r:.1f} ms")
