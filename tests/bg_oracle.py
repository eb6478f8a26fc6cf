"""The suite's three long CPU-oracle computations, started when the session starts and collected by the tests that need them.

Round 4's GPU suite took 783 s of the driver's 1200; 150 s of it were three full CPU counts that the GPU sat idle for:
the hg38-sized histogram of tests/test_gpu_scale.py (76-106 s on 16 threads) and the two whole-map digests of
tests/test_gpu_parity.py::test_full_map_digest_10M_reads.  They need nothing from the GPU -- inputs come from the oracle's own
generators -- so tests/conftest.py runs this file as a child process on half of the cores while the first hundreds of tests
use the device, and `collect(name)` hands a test its result (or computes it on the spot if there is no child: a test run on
its own).  Same oracle functions, same arguments, same checks; only the waiting moved.

usage (conftest): python tests/bg_oracle.py OUTDIR NTHREADS job [job ...]      jobs: hg, digest_k21, digest_k31q20"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
SEED = 20260130
JOBS = ("hg", "digest_k21", "digest_k31q20")


def compute(job, nthreads):
    import oracle_lib as O
    t0 = time.time()
    if job == "hg":
        lens = np.array(O.HG38_LENGTHS, dtype=np.uint64)
        flat = O.synth_hg(38, lens, nthreads=nthreads)
        total, distinct, _, hist = O.hist_flat_radix(flat, 21, nthreads=nthreads, npasses=8)
        out = {"total": total, "distinct": distinct, "hist": hist}
    else:
        k, minq = (21, None) if job == "digest_k21" else (31, 20)
        bases, qual = O.synth_reads(SEED, 1 << 27, 150, 0, 10_000_000, with_qual=minq is not None)
        total, distinct, digest = O.count_flat_radix(bases, k, qual=qual, min_quality=minq, nthreads=nthreads)
        out = {"total": total, "distinct": distinct, "digest": digest}
    out["cpu_seconds"] = time.time() - t0
    out["threads"] = nthreads
    return out


def _path(outdir, job):
    return os.path.join(outdir, f"bg_oracle_{job}.json")


def collect(job, nthreads, timeout=1500.0):
    """The result of `job`: the child's if conftest started one (waits for it), else computed here."""
    outdir = os.environ.get("KMERHIP_BG_ORACLE_DIR")
    if outdir:
        t_end = time.time() + timeout
        while time.time() < t_end:
            if os.path.exists(_path(outdir, job)):
                with open(_path(outdir, job)) as f:
                    return json.load(f)
            if os.path.exists(os.path.join(outdir, "bg_oracle_FAILED")) or os.path.exists(os.path.join(outdir, "bg_oracle_DONE")):
                break  # the child is gone without this result
            time.sleep(0.5)
    return compute(job, nthreads)


if __name__ == "__main__":
    outdir, nthreads, jobs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
    try:
        for job in jobs:
            res = compute(job, nthreads)
            tmp = _path(outdir, job) + ".tmp"
            with open(tmp, "w") as f:
                json.dump(res, f)
            os.replace(tmp, _path(outdir, job))
        open(os.path.join(outdir, "bg_oracle_DONE"), "w").close()
    except BaseException:
        open(os.path.join(outdir, "bg_oracle_FAILED"), "w").close()
        raise
