#!/usr/bin/env python3
"""Summarises a profiles_run.sh output directory (gpurun_out/prof_<tag>) into
profiles/<tag>_kernel_stats.csv + profiles/<tag>_summary.json (tracked, small).

usage: python tools/summarize_prof.py <tag> [kernel-substring]
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "count_"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
summary = {"tag": tag, "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline",
           "kernels": [], "pmc": {}}
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    for r in csv.DictReader(open(stats[0])):
        summary["kernels"].append({"name": r["Name"].split("(")[0], "calls": int(r["Calls"]),
                                   "avg_ms": float(r["AverageNs"]) / 1e6, "total_ms": float(r["TotalDurationNs"]) / 1e6,
                                   "pct": float(r["Percentage"])})
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for name, ctrs in acc.items():
            if not any(t in name for t in (kernel, "part1", "part2", "region_count", "scan_", "table_init")):
                continue
            for c, v in ctrs.items():
                summary["pmc"].setdefault(name, {})[c] = {"launches": len(v), "mean_per_launch": sum(v) / len(v),
                                                          "sum": sum(v)}
for name, ctrs in summary["pmc"].items():
    if "FETCH_SIZE" in ctrs and "WRITE_SIZE" in ctrs:
        # FETCH_SIZE/WRITE_SIZE are in KiB.  gfx950: FETCH_SIZE under-reports wide (16 B/lane) coalesced
        # streaming reads by 2x (MI355X_MICROARCH.md section HBM); narrow random reads are tallied per
        # 64-B request.  Both the raw and the doubled figure are kept.
        f = ctrs["FETCH_SIZE"]["mean_per_launch"] * 1024
        w = ctrs["WRITE_SIZE"]["mean_per_launch"] * 1024
        ctrs["hbm_bytes_per_launch_raw"] = f + w
        ctrs["hbm_bytes_per_launch_fetch_x2"] = 2 * f + w
    if "TCC_HIT_sum" in ctrs and "TCC_MISS_sum" in ctrs:
        h, m = ctrs["TCC_HIT_sum"]["sum"], ctrs["TCC_MISS_sum"]["sum"]
        ctrs["l2_hit_rate"] = h / (h + m) if h + m else None
# HBM bytes of one bench step = sum over the counting kernels (the profiled command runs warmup 1 + steps 1,
# i.e. every per-step kernel is launched twice).  Wide streaming reads: FETCH_SIZE x 2 (gfx950).
steps_profiled = 2
tot_raw = tot_x2 = 0.0
for name, ctrs in summary["pmc"].items():
    if "FETCH_SIZE" in ctrs and "WRITE_SIZE" in ctrs and "table_init" not in name:
        tot_raw += (ctrs["FETCH_SIZE"]["sum"] + ctrs["WRITE_SIZE"]["sum"]) * 1024 / steps_profiled
        tot_x2 += (2 * ctrs["FETCH_SIZE"]["sum"] + ctrs["WRITE_SIZE"]["sum"]) * 1024 / steps_profiled
summary["hbm_bytes_per_step_raw"] = tot_raw
summary["hbm_bytes_per_step_fetch_x2"] = tot_x2
with open(os.path.join(dst, f"{tag}_summary.json"), "w") as f:
    json.dump(summary, f, indent=1)
print(json.dumps(summary, indent=1)[:3000])
