#!/usr/bin/env python3
"""profiles/<tag>_summary.json (tools/summarize_prof.py) -> profiles/hbm_traffic.json, which bench.py reports as
roofline.traffic: HBM bytes per bench step, summed over every kernel of one step, from the separate rocprofv3 --pmc
passes of tools/profiles_run.sh.  FETCH_SIZE and WRITE_SIZE count KiB; FETCH_SIZE is doubled (gfx950 counts wide
coalesced streaming reads at half: MI355X_MICROARCH.md, HBM section -- every read of this pipeline is a 16-byte or
4-byte-per-lane coalesced stream).  The profiled run makes `launches` steps (warm-up + timed): per-step = sum / launches.

Other configurations profiled the same way (tools/profiles_run.sh <tag>_<name> with BENCH_ARGS) go under "configs":
--also NAME=TAG:reads:k:min_quality (min_quality -1 = none; reads 0 = the hg-shaped text), e.g.
  python tools/hbm_traffic.py r05c --also k31q20=r05c_k31q20:100000000:31:20 --also s125=r05c_s125:125000000:21:-1 --also hg=r05c_hg:0:21:-1
bench.py attaches them to the sub-results they match (roofline.traffic / traffic_frac).

usage: python tools/hbm_traffic.py <tag> [reads_per_gpu k] [--also NAME=TAG:reads:k:min_quality ...]"""
import json
import os
import sys

argv = list(sys.argv[1:])
also = []
while "--also" in argv:
    i = argv.index("--also")
    also.append(argv[i + 1])
    del argv[i:i + 2]
tag = argv[0]
reads = int(argv[1]) if len(argv) > 1 else 100_000_000
k = int(argv[2]) if len(argv) > 2 else 21
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


NOT_COUNTING = ("fasta_", "fastq_", "raw_", "scan_", "hist", "lookup")  # record scanning and output kernels: outside the count's roofline


def step_bytes(t, counting_only=True):
    """(2 x FETCH + WRITE, FETCH + WRITE) per step over the library's kernels of profiles/<t>_summary.json"""
    dd = json.load(open(os.path.join(root, "profiles", f"{t}_summary.json")))
    nsteps = max(kk["calls"] for kk in dd["kernels"] if "region_count" in kk["name"])
    ff = ww = 0.0
    for nm, ctrs in dd["pmc"].items():
        if "kh::" not in nm or "synth_reads" in nm or "table_init" in nm:  # (torch's kernels build the input; not part of a step)
            continue
        if counting_only and any(x in nm for x in NOT_COUNTING):
            continue
        ff += ctrs.get("FETCH_SIZE", {}).get("sum", 0.0) * 1024 / nsteps
        ww += ctrs.get("WRITE_SIZE", {}).get("sum", 0.0) * 1024 / nsteps
    return int(2 * ff + ww), int(ff + ww)

d = json.load(open(os.path.join(root, "profiles", f"{tag}_summary.json")))
steps = max(kk["calls"] for kk in d["kernels"] if "region_count" in kk["name"])
fetch = write = 0.0
per_kernel = {}
for name, ctrs in d["pmc"].items():
    f = ctrs.get("FETCH_SIZE", {}).get("sum", 0.0) * 1024 / steps
    w = ctrs.get("WRITE_SIZE", {}).get("sum", 0.0) * 1024 / steps
    if "synth_reads" in name or "table_init" in name:  # input generation / one-off table clearing: not part of a step
        continue
    fetch += f
    write += w
    if f + w > 1e8:
        per_kernel[name.replace("void kh::", "").replace("kh::", "")] = {"read_GB": round(2 * f / 1e9, 2), "written_GB": round(w / 1e9, 2)}
out = {
    "_comment": "HBM bytes per bench step (S100M, k=21), summed over the counting kernels of one step from separate rocprofv3 "
                f"--pmc passes (profiles/{tag}_summary.json, tools/hbm_traffic.py): FETCH_SIZE x 2 (gfx950 counts wide coalesced "
                "streaming reads at half, MI355X_MICROARCH.md HBM section; every read of this pipeline is a 16-B or 4-B/lane "
                "coalesced stream) + WRITE_SIZE.",
    "reads_per_gpu": reads, "k": k,
    "bytes_per_step": int(2 * fetch + write), "bytes_per_step_raw_fetch": int(fetch + write),
    "per_kernel": per_kernel, "tag": tag,
}
if also:
    out["configs"] = {}
    for spec in also:
        name, rest = spec.split("=", 1)
        t, r_, k_, q_ = rest.split(":")
        b2, braw = step_bytes(t)
        out["configs"][name] = {"tag": t, "reads": int(r_), "k": int(k_), "min_quality": None if int(q_) < 0 else int(q_),
                                "bytes_per_step": b2, "bytes_per_step_raw_fetch": braw}  # (the counting kernels: tools/summarize_prof.py keeps those)
json.dump(out, open(os.path.join(root, "profiles", "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
