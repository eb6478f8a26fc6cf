"""Where does the wall time of a large text push go?  (round 4: the hg-shaped sub-result took 4.3 s for 40 ms of kernels.)
python tools/text_stall_probe.py  -- on the GPU box; KMERHIP_TRACE=1 prints the library's own allocation / count walls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import krust_amd
import bench

dev = torch.device("cuda", 0)
text, flat = bench.hg_like_fasta_device(torch, dev)
del flat
torch.cuda.synchronize()
dc = krust_amd.DeviceCounter(21, capacity_hint=0, device=0)
for rep in range(4):
    t0 = time.perf_counter(); dc.reset(); torch.cuda.synchronize()
    t1 = time.perf_counter(); dc.push_text_device(text.data_ptr(), text.numel(), "fasta")
    t2 = time.perf_counter(); st = dc.finish()
    t3 = time.perf_counter(); h = dc.histogram()
    t4 = time.perf_counter()
    print(f"rep {rep}: reset {1e3*(t1-t0):.1f} push_text_device {1e3*(t2-t1):.1f} finish {1e3*(t3-t2):.1f} histogram {1e3*(t4-t3):.1f} ms; kernels {st['count_kernel_ms']:.1f} scan {st['text_scan_ms']:.1f}", flush=True)
dc.close()
