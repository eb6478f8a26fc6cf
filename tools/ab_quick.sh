#!/bin/bash
# quick look at the stage times of the headline, configs[3]'s share and the hg-shaped step (run through gpurun)
python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/q_s100.json 2> gpurun_out/q_s100.err
python bench.py --no-extras --no-cpu-baseline --reads 125000000 --steps 3 --warmup 1 > gpurun_out/q_s125.json 2> gpurun_out/q_s125.err
python bench.py --hg --steps 3 --warmup 1 > gpurun_out/q_hg.json 2> gpurun_out/q_hg.err
python bench.py --no-extras --no-cpu-baseline --k 31 --min-quality 20 --no-hint --steps 3 --warmup 1 > gpurun_out/q_k31.json 2> gpurun_out/q_k31.err
python - <<PY
import json
for n in ("s100","s125","hg","k31"):
    try:
        d=json.load(open(f"gpurun_out/q_{n}.json"))
        print(n, round(d["value"]/1e9,1), d["ms_per_step"], d["roofline"]["stages_ms"], d.get("verify",{}).get("ok"))
    except Exception as e: print(n, "failed", e)
PY
