#!/usr/bin/env python3
"""bench.py's hg-shaped sub-result (configs[4]: hg38's record lengths as FASTA text resident in HBM -> histogram) on its own."""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
import torch
import krust_amd
r = bench.sub_config_hg(krust_amd, torch, torch.device("cuda:0"), 0)
print(json.dumps({k: r[k] for k in r if k != "roofline"})[:1200])
print(json.dumps(r["roofline"]["stages_ms"]))
