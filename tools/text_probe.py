"""Throughput of the device record scanner (kh_push_text) and of the CLI end to end, device scan vs
host line parser.  usage: python tools/text_probe.py [n_reads]"""
import os
import subprocess
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from krust_amd import native  # noqa: E402
import oracle_lib as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
bases, qual = O.synth_reads(20260130, 1 << 28, 150, 0, n)
b = bases.reshape(n, 151)[:, :150]
q = qual.reshape(n, 151)[:, :150]
digits = (np.arange(n)[:, None] // 10 ** np.arange(9, -1, -1)[None, :]) % 10
rec2 = np.empty((n, 13 + 151 + 2 + 151), dtype=np.uint8)  # "@r%010d\n" seq "\n+\n" qual "\n"
rec2[:, 0] = ord("@")
rec2[:, 1] = ord("r")
rec2[:, 2:12] = digits + 48
rec2[:, 12] = 10
rec2[:, 13:163] = b
rec2[:, 163] = 10
rec2[:, 164] = ord("+")
rec2[:, 165] = 10
rec2[:, 166:316] = q
rec2[:, 316] = 10
text = rec2.reshape(-1)
print(f"text: {text.size / 1e9:.3f} GB, {n} reads", flush=True)

for minq in (None, 20):
    with native.DeviceCounter(21, min_quality=minq, capacity_hint=int(n * 12)) as dc:
        for it in range(3):
            dc.reset()
            t0 = time.perf_counter()
            dc.push_text(text, "fastq")
            st = dc.finish()
            dt = time.perf_counter() - t0
            print(f"push_text minq={minq} it={it}: {dt * 1e3:.1f} ms wall  ({text.size / dt / 1e9:.2f} GB/s text, "
                  f"{st['kmers'] / dt / 1e9:.2f} G k-mers/s)  scan={st['text_scan_ms']:.2f} ms "
                  f"({text.size / st['text_scan_ms'] / 1e6:.1f} GB/s)  count={st['count_kernel_ms']:.2f} ms h2d={st['h2d_ms']:.1f} ms", flush=True)
        d = torch.frombuffer(memoryview(text), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        for it in range(2):
            dc.reset()
            t0 = time.perf_counter()
            dc.push_text_device(d.data_ptr(), text.size, "fastq")
            st = dc.finish()
            dt = time.perf_counter() - t0
            print(f"push_text_device minq={minq}: {dt * 1e3:.1f} ms wall, scan={st['text_scan_ms']:.2f} ms count={st['count_kernel_ms']:.2f} ms", flush=True)
        del d

path = os.environ.get("PROBE_FILE", "/tmp/text_probe.fq")
text.tofile(path)
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "krust_amd", "host", "kmerust")
for label, env in (("device scan", {}), ("host parser", {"KMERUST_HOST_PARSE": "1"})):
    for args in (["21", path, "--format", "histogram", "--quiet"], ["21", path, "-Q", "20", "--format", "histogram", "--quiet"]):
        t0 = time.perf_counter()
        r = subprocess.run([BIN, *args], capture_output=True, env={**os.environ, **env})
        dt = time.perf_counter() - t0
        print(f"CLI {label} {' '.join(args[2:4]) if '-Q' in args else '':8s}: {dt:.2f} s  rc={r.returncode} lines={len(r.stdout.splitlines())} "
              f"({text.size / dt / 1e9:.2f} GB/s of text)", flush=True)
os.remove(path)
