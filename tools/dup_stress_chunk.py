#!/usr/bin/env python3
"""One text chunk, many times: is kh_push_text_device deterministic?  And the same bases as a flat buffer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import krust_amd
import oracle_lib as O
n = int(os.environ.get("NREADS", 846_000))
bases, qual = O.synth_reads(20260130, 1 << 28, 150, 0, n)
b = bases.reshape(n, 151)[:, :150]; q = qual.reshape(n, 151)[:, :150]
digits = (np.arange(n)[:, None] // 10 ** np.arange(9, -1, -1)[None, :]) % 10
rec = np.empty((n, 317), dtype=np.uint8)
rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 2:12] = digits + 48; rec[:, 12] = 10
rec[:, 13:163] = b; rec[:, 163] = 10; rec[:, 164] = ord("+"); rec[:, 165] = 10; rec[:, 166:316] = q; rec[:, 316] = 10
text = rec.reshape(-1)
dtext = torch.from_numpy(text).cuda()
dflat = torch.from_numpy(bases).cuda()
torch.cuda.synchronize()
m = O.OracleMap(); tot = m.scan_flat(bases, 21, nthreads=8); truth = len(m)
print("oracle:", tot, truth, flush=True)
for name, fn in (("flat", lambda dc: dc.push_device(dflat.data_ptr(), None, dflat.numel())),
                 ("text", lambda dc: dc.push_text_device(dtext.data_ptr(), text.size, "fastq"))):
    for path in (None, "direct"):
        bad = 0
        for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
            with krust_amd.DeviceCounter(21, path=path) as dc:
                fn(dc)
                st = dc.finish()
            if st["distinct"] != truth or st["kmers"] != tot:
                bad += 1
                print(f"  {name}/{path} rep {rep}: kmers {st['kmers']} distinct {st['distinct']} (truth {truth})", flush=True)
        print(f"{name} path={path}: mismatches {bad}", flush=True)
