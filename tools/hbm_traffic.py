#!/usr/bin/env python3
"""profiles/<tag>_summary.json (tools/summarize_prof.py) -> profiles/hbm_traffic.json, which bench.py reports as
roofline.traffic: HBM bytes per bench step, summed over every kernel of one step, from the separate rocprofv3 --pmc
passes of tools/profiles_run.sh.  FETCH_SIZE and WRITE_SIZE count KiB; FETCH_SIZE is doubled (gfx950 counts wide
coalesced streaming reads at half: MI355X_MICROARCH.md, HBM section -- every read of this pipeline is a 16-byte or
4-byte-per-lane coalesced stream).  The profiled run makes `launches` steps (warm-up + timed): per-step = sum / launches.

usage: python tools/hbm_traffic.py <tag> [reads_per_gpu k]"""
import json
import os
import sys

tag = sys.argv[1]
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 21
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
json.dump(out, open(os.path.join(root, "profiles", "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
