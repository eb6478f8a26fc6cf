#!/usr/bin/env python3
"""bench.py's `cli` leg on its own: usage python tools/cli_leg.py [reads ...]  (environment knobs of the binary pass through)."""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
import torch
import krust_amd
for reads in [int(x) for x in sys.argv[1:]] or [10_000_000]:
    r = bench.cli_leg(krust_amd, torch, torch.device("cuda:0"), 0, reads=reads)
    print(json.dumps({"reads": reads, "wall_s": r.get("wall_s"), "text_GBps": r.get("text_GBps"), "ok": r.get("ok"),
                      "runs": [(round(x["wall_s"], 3), x["phases"]) for x in r.get("runs", [])]}))
