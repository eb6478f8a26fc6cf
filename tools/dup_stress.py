#!/usr/bin/env python3
"""Stress for a duplicate-key race: many unhinted multi-batch builds of the same input; distinct must
never vary.  usage: python tools/dup_stress.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import krust_amd
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
reads, rl, k = 16_000_000, 150, 21
tb = torch.empty(reads * (rl + 1), dtype=torch.uint8, device="cuda")
krust_amd.synth_reads_device(tb.data_ptr(), None, int(os.environ.get("SEED", 20260130)), 1 << 28, rl, 0, reads)
torch.cuda.synchronize()
with krust_amd.DeviceCounter(k, capacity_hint=600_000_000) as dc:
    dc.push_device(tb.data_ptr(), None, tb.numel())
    ref = dc.finish()
    if os.environ.get("DUMP"):
        rkeys, rcnts = dc.result(sort=True)
print("reference (one hinted push):", ref["kmers"], ref["distinct"], flush=True)
nsl = int(os.environ.get("SLICES", 19))
per = (reads // nsl) * (rl + 1)
bad = 0
for rep in range(reps):
    dc = krust_amd.DeviceCounter(k, path=os.environ.get("PATHMODE") or None, capacity_hint=int(os.environ.get("HINT", 0)))
    for s_ in range(nsl):
        lo = s_ * per
        n = tb.numel() - lo if s_ == nsl - 1 else per
        dc.push_device(tb.data_ptr() + lo, None, n)
    st = dc.finish()
    rs = dc.result_size()
    ok = st["kmers"] == ref["kmers"] and st["distinct"] == ref["distinct"] == rs
    bad += not ok
    if not ok or rep % 10 == 0:
        print(f"rep {rep}: kmers {st['kmers']} distinct {st['distinct']} result_size {rs} grows {st['grows']} {'OK' if ok else 'MISMATCH'}", flush=True)
    if not ok and os.environ.get("DUMP"):
        keys, cnts = dc.result(sort=False)
        order = np.argsort(keys, kind="stable")
        ks, cs = keys[order], cnts[order]
        dup = np.flatnonzero(ks[1:] == ks[:-1])
        slots = st["table_slots"]
        rb = (slots // 4096).bit_length() - 1
        print(f"  total counts {int(cs.sum())} (k-mers {st['kmers']}); duplicate keys: {len(dup)}; table 2^{rb} regions")
        for i in dup[:12]:
            key = int(ks[i])
            pos = krust_amd.owner(key, k, 1 << min(rb + 12, 32)) if rb + 12 <= 32 else -1
            print(f"    key {key:#x} counts {int(cs[i])}+{int(cs[i+1])} region {pos >> 12} start {pos & 4095}")
        extra = np.setdiff1d(ks, rkeys, assume_unique=True)
        missing = np.setdiff1d(rkeys, ks, assume_unique=True)
        print(f"  keys not in the reference: {len(extra)}; reference keys missing: {len(missing)}")
        common = np.intersect1d(ks, rkeys, assume_unique=True)
        ca = cs[np.searchsorted(ks, common)]; cb = rcnts[np.searchsorted(rkeys, common)]
        diff = np.flatnonzero(ca != cb)
        print(f"  common keys with a different count: {len(diff)}")
        rbq = rb
        def where(key):
            pos = krust_amd.owner(int(key), k, 1 << min(rbq + 12, 32))
            return f"region {pos >> 12} (p1 {pos >> 12 >> (rbq - 10)}) start {pos & 4095}"
        for e in extra[:12]:
            print(f"    extra   {krust_amd.unpack(int(e), k)} count {int(cs[np.searchsorted(ks, e)])} {where(e)}")
        for e in missing[:12]:
            print(f"    missing {krust_amd.unpack(int(e), k)} ref count {int(rcnts[np.searchsorted(rkeys, e)])} {where(e)}")
        for i in diff[:12]:
            print(f"    differs {krust_amd.unpack(int(common[i]), k)} {int(ca[i])} vs ref {int(cb[i])} {where(common[i])}")
        dc.close()
        break
    dc.close()
print("mismatches:", bad, "of", reps)
