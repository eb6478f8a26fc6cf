#!/bin/bash
# Where the CLI's wall time goes on a 5 GB FASTQ (16 M reads) in /dev/shm: library trace line + wall, next to `cat`.
python - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib as O
n = 16_000_000
bases, qual = O.synth_reads(20260130, 1 << 28, 150, 0, n)
b = bases.reshape(n, 151)[:, :150]; q = qual.reshape(n, 151)[:, :150]
digits = (np.arange(n)[:, None] // 10 ** np.arange(9, -1, -1)[None, :]) % 10
rec = np.empty((n, 317), dtype=np.uint8)
rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 2:12] = digits + 48; rec[:, 12] = 10
rec[:, 13:163] = b; rec[:, 163] = 10; rec[:, 164] = ord("+"); rec[:, 165] = 10; rec[:, 166:316] = q; rec[:, 316] = 10
rec.reshape(-1).tofile("/dev/shm/p.fq")
PY
python - <<'PY'
import subprocess, time, os
for kb in ("",) * 12:
    env = dict(os.environ, KMERHIP_TRACE="1")
    if kb: env["KMERUST_TEXT_CHUNK_KB"] = kb
    for i in range(1):
        t0 = time.perf_counter()
        r = subprocess.run(["krust_amd/host/kmerust", "21", "/dev/shm/p.fq", "--format", "histogram", "--quiet"], capture_output=True, env=env)
        dt = time.perf_counter() - t0
    hist = [tuple(map(int, l.split())) for l in r.stdout.decode().splitlines()]
    print(f"chunk_kb={kb or 'default'}: CLI wall {dt:.2f} s; histogram: distinct {sum(f for _, f in hist)}, k-mers {sum(c * f for c, f in hist)};",
          [w for w in r.stderr.decode().split() if w.startswith("distinct=") or w.startswith("launches=")],
          [l for l in r.stderr.decode().splitlines() if "overflow" in l or "grown" in l])
t0 = time.perf_counter(); open("/dev/shm/p.fq", "rb").read(); print(f"read() of the file: {time.perf_counter() - t0:.2f} s")
PY
rm -f /dev/shm/p.fq
