"""Does a rank of a merge -- a context with a communicator attached -- still count configs[3]'s share (125 M reads) in ONE
partitioned batch?  python tools/comm_budget_probe.py [reads]  (GPU box; a world of one rank attaches the communicator)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import krust_amd as K

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
n = reads * 151
bases = torch.empty(n, dtype=torch.uint8, device="cuda")
K.synth_reads_device(bases.data_ptr(), None, 20260130, 1 << 27, 150, 3 * reads, reads)
torch.cuda.synchronize()
for with_comm in (False, True):
    with K.DeviceCounter(21, capacity_hint=0, trace=True) as dc:
        if with_comm:
            dc.comm_init(1, 0, K.comm_unique_id())
        dc.push_device(bases.data_ptr(), None, n)
        st = dc.finish()
        print(f"comm={with_comm}: batches {st['part_batches']} kernel {st['count_kernel_ms']:.1f} ms distinct {st['distinct']} free {torch.cuda.mem_get_info()[0] / 1e9:.1f} GB", flush=True)
        if with_comm:
            for rnd in range(3):  # (the first merge of a process may wait seconds for the driver behind a large hipMalloc)
                if rnd:
                    dc.reset()
                    dc.push_device(bases.data_ptr(), None, n)
                    dc.finish()
                info = dc.merge_across()
                print(f"merge {rnd}:", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in info.items()}, flush=True)
