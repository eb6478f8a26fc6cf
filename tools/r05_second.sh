#!/bin/bash
# round 5, second GPU call: exchange tests (all), counter evidence for configs[2] (k = 31 -Q 20, no hint) and configs[4] (hg-shaped)
mkdir -p gpurun_out/r05b
timeout 900 python -m pytest tests/test_gpu_exchange.py -x -q -m gpu --durations=8 > gpurun_out/r05b/t_exchange.log 2>&1
echo "exchange rc=$?"; tail -3 gpurun_out/r05b/t_exchange.log
BENCH_ARGS="--k 31 --min-quality 20 --no-hint" bash tools/profiles_run.sh r05a_k31q20 > gpurun_out/r05b/prof_k31.log 2>&1
BENCH_ARGS="--k 31 --min-quality 20 --no-hint" bash tools/sq_probe.sh r05a_k31q20 > gpurun_out/r05b/sq_k31.log 2>&1
BENCH_ARGS="--hg" bash tools/profiles_run.sh r05a_hg > gpurun_out/r05b/prof_hg.log 2>&1
BENCH_ARGS="--hg" bash tools/sq_probe.sh r05a_hg > gpurun_out/r05b/sq_hg.log 2>&1
tail -5 gpurun_out/r05b/sq_k31.log gpurun_out/r05b/sq_hg.log
tail -c 600 gpurun_out/prof_r05a_hg/bench_trace.log
