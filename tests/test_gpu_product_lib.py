"""The PRODUCT library (krust_amd/lib/libkmerhip.so) as it ships: the rest of the GPU suite loads the test build
(libkmerhip_testing.so, tests/conftest.py), which is the same code plus the environment switches that force kernel variants,
table geometries and failures.  Here a fresh process loads libkmerhip.so and (1) counts reads through both insert paths and
the text path against the oracle, (2) shows that the test switches are NOT in it: with KMERHIP_L2_ARENA=0, KMERHIP_NARROW=0,
KMERHIP_TABLE_REGIONS and KMERHIP_FAULT set, the batch still takes the arena kernel, the table still is the 8-byte image's
size chosen by the library, and a merge does not fail."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import krust_amd
from krust_amd import native
import oracle_lib as O
assert native.LIB_PATH.endswith("libkmerhip.so"), native.LIB_PATH
bases, qual = O.synth_reads(20260130, 1 << 20, 150, 0, 120_000)
out = {}
for k, minq in ((21, None), (31, 20)):
    m = O.OracleMap()
    total = m.scan_flat(bases, k, qual=qual if minq is not None else None, min_quality=minq, nthreads=4)
    ok_, oc_ = m.arrays()
    for path in ("direct", "partition"):
        with krust_amd.DeviceCounter(k, min_quality=minq, path=path, capacity_hint=50_000_000) as dc:  # (2^15 regions: 1024 partitions x 32 buckets)
            dc.push(bases, qual if minq is not None else None)
            st = dc.finish()
            keys, cnts = dc.result()
        assert st["kmers"] == total and np.array_equal(keys, ok_) and np.array_equal(cnts, oc_), (k, path)
        out[f"k{k}-{path}"] = {"slots": st["table_slots"], "level2_count_ms": st["stage_ms"]["level2_count"], "level2_ms": st["stage_ms"]["level2"]}
# a merge of two ranks sharing the device: KMERHIP_FAULT must not make it fail
with krust_amd.DeviceGroup(21, [0, 0], capacity_hint=3_000_000) as g:
    g[0].push(bases[: 60_000 * 151]); g[1].push(bases[60_000 * 151:])
    infos = g.merge()
    out["merge_paths"] = [i["path"] for i in infos]
print("RESULT " + json.dumps(out))
'''


def test_product_library_counts_like_the_oracle_and_has_no_test_switches():
    env = dict(os.environ)
    env.pop("KMERHIP_LIB", None)  # the product library
    env.update(KMERHIP_L2_ARENA="0", KMERHIP_NARROW="0", KMERHIP_TABLE_REGIONS=str(1024 * 3), KMERHIP_FAULT="0:start", KMERHIP_P1_BINS="0")
    p = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + CHILD], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    # 120 k reads = 15.6 M windows into a table hinted to 1024 x 32 regions: level 2 through the arena
    # kernel (no counting pass) although KMERHIP_L2_ARENA=0 asks for the exact one, and never the 1024 x 3 regions of KMERHIP_TABLE_REGIONS
    part = res["k21-partition"]
    assert part["slots"] == 1 << 27 and part["level2_ms"] > 0 and part["level2_count_ms"] == 0, res
    assert all(pth.startswith("regions") for pth in res["merge_paths"]), res
