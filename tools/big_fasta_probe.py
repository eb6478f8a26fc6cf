#!/usr/bin/env python3
"""hg38-shaped CLI check (BASELINE configs[4] without the real file): a FASTA with a 600 Mbp record --
larger than the 256 MiB text chunk, so the chunk buffer has to grow -- through `kmerust 21 <file>
--format histogram`, total against the oracle's streaming scan.  Run on the GPU box (uses /dev/shm)."""
import os, sys, time, subprocess
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle_lib as O
rng = np.random.default_rng(1)
path = "/dev/shm/big.fa"
lens = [600_000_000, 1000, 90_000_000, 37]
seqs = []
with open(path, "wb") as f:
    for i, n in enumerate(lens):
        s = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)[rng.choice(9, size=n, p=[.24, .24, .24, .24, .01, .01, .005, .005, .01])]
        seqs.append(s)
        f.write(b">chr%d\n" % i)
        full = n // 60
        body = np.empty((full, 61), dtype=np.uint8); body[:, :60] = s[: full * 60].reshape(full, 60); body[:, 60] = 10
        f.write(body.tobytes())
        if n % 60: f.write(s[full * 60:].tobytes() + b"\n")
print("file", os.path.getsize(path) / 1e9, "GB", flush=True)
m = O.OracleMap()
tot = 0
for s in seqs:
    tot += m.scan_flat(s, 21, sample_mask=1023, nthreads=16)
hist_sum = tot
t0 = time.perf_counter()
r = subprocess.run([os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "krust_amd", "host", "kmerust"), "21", path, "--format", "histogram", "--quiet"], capture_output=True)
dt = time.perf_counter() - t0
got = [tuple(map(int, l.split(b"\t"))) for l in r.stdout.splitlines()]
print("rc", r.returncode, r.stderr[-300:], "time %.2f s" % dt, "lines", len(got))
print("sum(c*f) == total windows:", sum(c * f for c, f in got) == tot, sum(c * f for c, f in got), tot)
os.remove(path)
