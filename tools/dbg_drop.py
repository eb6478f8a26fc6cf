import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["KMERHIP_LIB"] = "libkmerhip_testing.so"
import numpy as np, ctypes as C, threading
import krust_amd as K
import oracle_lib as O
world, k = 2, 19
n_reads = 40000
full_b, _ = O.synth_reads(20260130, 1 << 20, 150, 0, n_reads, with_qual=False)
per = n_reads // world
os.environ["KMERHIP_MERGE_TIMEOUT_S"] = "60"
os.environ["KMERHIP_TRACE"] = os.environ.get("DBG_TRACE", "0")
for bad in (1, 0):
    with K.DeviceGroup(k, [0] * world, capacity_hint=3_000_000) as g:
        for r, dc in enumerate(g.counters):
            lo, hi = r * per, (n_reads if r == world - 1 else (r + 1) * per)
            dc.push(full_b[lo * 151: hi * 151])
        os.environ["KMERHIP_FAULT"] = f"{bad}:drop_half"
        L = K.lib(); out = [None] * world
        def run(i):
            info = K.native.KhMergeInfo()
            rc = L.kh_merge_across(g[i]._h, C.byref(info))
            out[i] = (rc, L.kh_last_error(g[i]._h).decode(), K.native.merge_info_dict(info))
        th = [threading.Thread(target=run, args=(i,)) for i in range(world)]
        [t.start() for t in th]; [t.join() for t in th]
        for o in out: print("bad", bad, o[0], o[1][:200], o[2]["path"], o[2]["conserved"], o[2]["recv_units"], flush=True)
        del os.environ["KMERHIP_FAULT"]
