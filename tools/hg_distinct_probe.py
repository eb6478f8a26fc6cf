#!/usr/bin/env python3
"""Locates a distinct-count deficit seen on the hg-like assembly (tests/test_gpu_scale.py): counts the same
records through several routes of the library and reports where they disagree.

  A  one push_device of the whole flat buffer, capacity hint (one FRESH partitioned pass, no growth)
  B  the same through the direct path (device atomics)
  C  what the CLI does: no hint, whole records pushed in ~512 MB pieces (table grows, later pieces are
     non-fresh passes or direct inserts)
For every pair of routes: distinct, total, and -- via kh_lookup of one table's keys in the other -- the keys
whose counts differ, with their table-hash coordinates.

usage: hg_distinct_probe.py [scale_percent=100] [routes=ABC]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import krust_amd as K  # noqa: E402
import oracle_lib as O  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
routes = sys.argv[2] if len(sys.argv) > 2 else "ABC"
NCPU = 16
lens = np.maximum(1, (np.array(O.HG38_LENGTHS, dtype=np.float64) * scale / 100.0)).astype(np.uint64)
flat = O.synth_hg(38, lens, nthreads=NCPU)
print(f"scale {scale}%: {int(lens.sum())} bases", flush=True)
t0 = time.time()
if os.environ.get("WANT"):  # "total,distinct" known from an earlier run of the oracle
    want_total, want_distinct = (int(x) for x in os.environ["WANT"].split(","))
else:
    want_total, want_distinct, _, _ = O.hist_flat_radix(flat, 21, nthreads=NCPU, npasses=8, min_count=1)
print(f"oracle: total {want_total} distinct {want_distinct} ({time.time() - t0:.1f} s)", flush=True)

tb = torch.from_numpy(flat).cuda()
torch.cuda.synchronize()
tables = {}


def report(name, dc):
    st = dc.finish()
    print(f"route {name}: kmers {st['kmers']} distinct {st['distinct']} slots {st['table_slots']} grows {st['grows']} "
          f"batches {st['part_batches']} launches {st['launches']}  d_total {st['kmers'] - want_total} d_distinct {st['distinct'] - want_distinct}",
          flush=True)
    tables[name] = dc


if "A" in routes:
    dc = K.DeviceCounter(21, capacity_hint=int(want_distinct * 1.02), path="partition")
    dc.push_device(tb.data_ptr(), None, tb.numel())
    report("A", dc)
if "B" in routes:
    dc = K.DeviceCounter(21, capacity_hint=int(want_distinct * 1.02), path="direct")
    dc.push_device(tb.data_ptr(), None, tb.numel())
    report("B", dc)
if "C" in routes:
    dc = K.DeviceCounter(21)
    ends = np.cumsum(lens + np.uint64(1)).astype(np.int64)
    lo = 0
    piece = 512 << 20
    while lo < flat.size:
        # whole records: the largest record end within `piece`, else the next record end
        cand = ends[(ends > lo) & (ends <= lo + piece)]
        hi = int(cand[-1]) if cand.size else int(ends[ends > lo][0])
        dc.push_device(tb.data_ptr() + lo, None, hi - lo)
        lo = hi
    report("C", dc)

names = sorted(tables)
for i, a in enumerate(names):
    for b in names[i + 1:]:
        ka, ca = tables[a].result(sort=False)
        bad = 0
        for off in range(0, ka.size, 1 << 27):
            kk, cc = ka[off:off + (1 << 27)], ca[off:off + (1 << 27)]
            got = tables[b].lookup(kk)
            d = np.flatnonzero(got != cc)
            bad += d.size
            for j in d[:20]:
                key = int(kk[j])
                print(f"  {a} vs {b}: key {key:#x} {K.unpack(key, 21)}  {a}={int(cc[j])} {b}={int(got[j])}", flush=True)
        print(f"{a} vs {b}: {bad} keys of {a} differ in {b}", flush=True)
        del ka, ca
