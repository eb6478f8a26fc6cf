"""A merge in a world of ONE rank must leave the table's content as it was (every key is the rank's own).
python tools/world1_merge_probe.py  (GPU box; KMERHIP_LIB=libkmerhip_testing.so for the KMERHIP_TABLE_REGIONS rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import krust_amd as K

cases = [(2_000_000, None, None), (2_000_000, str(1024 * 48), None), (2_000_000, str(1024 * 768), None), (100_000_000, None, None), (125_000_000, None, "200"), (125_000_000, None, None)]
for reads, regions, budget in cases:
    for name, val in (("KMERHIP_TABLE_REGIONS", regions), ("KMERHIP_PART_BUDGET_GB", budget)):
        if val is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = val
    n = reads * 151
    bases = torch.empty(n, dtype=torch.uint8, device="cuda")
    K.synth_reads_device(bases.data_ptr(), None, 20260130, 1 << 27, 150, 3 * reads, reads)
    torch.cuda.synchronize()
    with K.DeviceCounter(21, capacity_hint=0) as dc:
        dc.comm_init(1, 0, K.comm_unique_id())
        dc.push_device(bases.data_ptr(), None, n)
        st = dc.finish()
        h0 = dc.histogram()
        info = dc.merge_across()
        st2 = dc.finish()
        h1 = dc.histogram()
        print(f"reads {reads} regions {regions} budget {budget}: batches {st['part_batches']} slots {st['table_slots']} (1024 x {st['table_slots'] // 4096 / 1024:g}) distinct {st['distinct']} -> {st2['distinct']} "
              f"{info['path']} pieces {info['pieces']} {'OK' if h0 == h1 and st['distinct'] == st2['distinct'] else 'KEYS LOST'}", flush=True)
    del bases
