#!/usr/bin/env python3
"""As dup_stress.py, through kh_push_text (FASTQ text from host memory, CLI-sized chunks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import krust_amd
import oracle_lib as O
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 16_000_000
bases, qual = O.synth_reads(20260130, 1 << 28, 150, 0, n)
b = bases.reshape(n, 151)[:, :150]; q = qual.reshape(n, 151)[:, :150]
digits = (np.arange(n)[:, None] // 10 ** np.arange(9, -1, -1)[None, :]) % 10
rec = np.empty((n, 317), dtype=np.uint8)
rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 2:12] = digits + 48; rec[:, 12] = 10
rec[:, 13:163] = b; rec[:, 163] = 10; rec[:, 164] = ord("+"); rec[:, 165] = 10; rec[:, 166:316] = q; rec[:, 316] = 10
text = rec.reshape(-1)
per = ((256 << 20) // (317 * 16)) * 317 * 16  # whole records, 16-byte aligned chunk starts
expect = int(os.environ.get("EXPECT", 428182064))
bad = 0
mode = os.environ.get("MODE", "host")
if mode == "device":
    import torch
    dtext = torch.from_numpy(text).cuda()
    torch.cuda.synchronize()
for rep in range(reps):
    with krust_amd.DeviceCounter(21, path=os.environ.get('PATHMODE') or None, capacity_hint=int(os.environ.get('HINT', 0))) as dc:
        for lo in range(0, text.size, per):
            if mode == "device":
                assert (dtext.data_ptr() + lo) % 16 == 0 or True
                dc.push_text_device(dtext.data_ptr() + lo, min(per, text.size - lo), "fastq")
            else:
                dc.push_text(text[lo:lo + per], "fastq")
        st = dc.finish()
        rs = dc.result_size()
    ok = st["distinct"] == expect == rs
    bad += not ok
    if not ok or rep == 0: print(f"rep {rep}: kmers {st['kmers']} distinct {st['distinct']} result_size {rs} grows {st['grows']} {'OK' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad, "of", reps)
