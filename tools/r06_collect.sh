#!/bin/bash
# after `gpurun -- bash tools/r06_final.sh`: what the run left under gpurun_out/ -> profiles/r06_* (run here, in the container).
# usage: bash tools/r06_collect.sh "<YYYY-mm-dd HH:MM:SS>"   (UTC; files of earlier runs older than that are dropped first)
set -e
cd "$(dirname "$0")/.."
CUT=${1:?the time the run started}
find gpurun_out/prof_r06 gpurun_out/prof_r06_s125 gpurun_out/prof_r06_hg gpurun_out/prof_r06_k31q20 gpurun_out/sq_r06 gpurun_out/sq_r06_k31q20 gpurun_out/sq_r06_fm -type f ! -newermt "$CUT" -delete
for t in r06 r06_s125 r06_hg r06_k31q20; do python tools/summarize_prof.py $t > /dev/null; done
python tools/hbm_traffic.py r06 --also k31q20=r06_k31q20:100000000:31:20 --also s125=r06_s125:125000000:21:-1 --also hg=r06_hg:0:21:-1 > /dev/null
python tools/sq_to_json.py r06_k31q20 "k=31 -Q 20=gpurun_out/r06z/sq_k31q20.log" --comment "tools/sq_probe.sh r06_k31q20 (BENCH_ARGS=--k 31 --min-quality 20 --no-hint): SQ counters of configs[2]'s hot kernels, one step" > /dev/null
cp gpurun_out/sq_r06_fm/summary.txt profiles/r06_fm_sq_counters.txt
cp gpurun_out/r06z/fm_kernel_stats.csv profiles/r06_forcemerge_kernel_stats.csv
cp gpurun_out/r06z/host_push_probe.txt profiles/r06_host_push_probe.txt
python tools/sq_to_json.py r06 "k=21 (default build)=gpurun_out/r06z/sq.log" --comment "tools/sq_probe.sh r06 (two rocprofv3 --pmc passes of bench.py --steps 1 --warmup 0 --no-extras --no-verify, S100M, k = 21): SQ counters of the hot kernels of one step, the round's last binary" > /dev/null
cp gpurun_out/r06z/bench.json profiles/r06_bench.json
cp gpurun_out/r06z/bench_full.json profiles/r06_bench_full.json
cp gpurun_out/r06z/forcemerge.json profiles/r06_forcemerge_bench.json
cp gpurun_out/r06z/group4.json profiles/r06_group4_bench.json
cp gpurun_out/r06z/skew_probe.txt profiles/r06_skew_probe.txt
python - <<'PY'
import json
d = json.load(open("profiles/r06_bench.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("stages_ms"), d["roofline"]["kernel_ms_per_step"])
print(json.dumps(d.get("unhinted")), json.dumps(d.get("cli")))
for c in d.get("configs", []):
    print(c["workload"][:60], round(c["value"] / 1e9, 1), c["ms_per_step"], c["frac"], c["verify_ok"], c.get("traffic_frac"))
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["optimised_value"], d["verify"]["ok"], json.dumps(d.get("end_to_end")))
t = json.load(open("profiles/hbm_traffic.json"))
print(t["bytes_per_step"], {k: v["bytes_per_step"] for k, v in t["configs"].items()})
f = json.load(open("profiles/r06_bench_full.json"))
for c in f["configs"]:
    if "hg" in c.get("workload", ""):
        print({k: round(c[k], 2) for k in ("ms_per_step", "count_ms", "histogram_ms", "text_scan_ms")}, c["roofline"]["stages_ms"])
fm = json.load(open("profiles/r06_forcemerge_bench.json"))
print("force-merge", fm["value"], fm["ms_per_step"], fm["config"]["merge"]["phase_ms"])
PY
grep -v "^\[W\|amdgpu" profiles/r06_skew_probe.txt | tail -6 | cut -c1-110
