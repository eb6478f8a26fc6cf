#!/usr/bin/env python3
"""gpurun_out/<log of tools/sq_probe.sh> -> profiles/<tag>_sq_counters.json (the hot kernels' SQ counters of one step).
usage: python tools/sq_to_json.py <tag> <label>=<log> [<label>=<log> ...] [--comment TEXT]"""
import ast
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
comment = ""
out = {}
args = sys.argv[2:]
if "--comment" in args:
    i = args.index("--comment")
    comment = args[i + 1]
    del args[i:i + 2]
for a in args:
    label, path = a.rsplit("=", 1)
    d = {}
    for line in open(os.path.join(root, path)):
        m = re.match(r"^(.*?)\s+(\{'SQ_.*\})\s*$", line)
        if not m:
            continue
        name = m.group(1).strip()
        if not any(t in name for t in ("part1", "part2_arena", "part2_count", "part2_scatter", "region_count", "hot_buckets", "ntable_hist", "fasta_compact")):
            continue
        vals = {k: float(v) for k, v in ast.literal_eval(m.group(2)).items()}
        d.setdefault(name, {}).update(vals)
    for name, c in d.items():
        if "SQ_INSTS_VALU" in c:
            c["valu_issue_ms_at_4_cycles"] = c["SQ_INSTS_VALU"] * 4 / 1024 / 2.4e9 * 1e3
        if c.get("SQ_LDS_IDX_ACTIVE"):
            c["lds_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
        if c.get("SQ_INSTS_VALU"):
            c["salu_per_valu"] = c.get("SQ_INSTS_SALU", 0.0) / c["SQ_INSTS_VALU"]
    out[label] = d
out = {"_comment": comment or f"tools/sq_probe.sh {tag}: two rocprofv3 --pmc passes of bench.py --steps 1 --warmup 0 --no-extras --no-verify; SQ counters of the hot kernels, summed over the launches of one step", **out}
with open(os.path.join(root, "profiles", f"{tag}_sq_counters.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1)[:1500])
