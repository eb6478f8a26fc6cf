import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, krust_amd
reads = 4_000_000
tb = torch.empty(reads * 151, dtype=torch.uint8, device="cuda")
krust_amd.synth_reads_device(tb.data_ptr(), None, 20260130, 1 << 27, 150, 0, reads)
torch.cuda.synchronize()
for trial in range(2):
    t0 = time.perf_counter()
    dc = krust_amd.DeviceCounter(21, capacity_hint=int(os.environ.get("HINT", 0)))
    t1 = time.perf_counter()
    dc.push_device(tb.data_ptr(), None, tb.numel()); st = dc.finish()
    t2 = time.perf_counter()
    dc.reset(); dc.push_device(tb.data_ptr(), None, tb.numel()); st = dc.finish()
    t3 = time.perf_counter()
    print(f"trial {trial}: create {1e3*(t1-t0):.1f} ms, first count {1e3*(t2-t1):.1f} ms (kernels {st['count_kernel_ms']:.1f}), second {1e3*(t3-t2):.1f} ms")
    dc.close()
for gb in (1, 4, 16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x = torch.empty(gb << 30, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    t1 = time.perf_counter()
    x.fill_(1); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"torch.empty {gb} GiB: {1e3*(t1-t0):.1f} ms; first touch fill {1e3*(t2-t1):.1f} ms")
    del x; torch.cuda.empty_cache()
