#!/bin/bash
# round 5, final measurements: profiles of the headline command, a driver-style bench run, the count-and-merge pre-flight
mkdir -p gpurun_out/r05z
bash tools/profiles_run.sh r05c > gpurun_out/r05z/prof.log 2>&1
bash tools/sq_probe.sh r05c > gpurun_out/r05z/sq.log 2>&1
BENCH_ARGS="--reads 125000000" bash tools/profiles_run.sh r05c_s125 > gpurun_out/r05z/prof_s125.log 2>&1
BENCH_ARGS="--hg" bash tools/profiles_run.sh r05c_hg > gpurun_out/r05z/prof_hg.log 2>&1
BENCH_ARGS="--k 31 --min-quality 20 --no-hint" bash tools/profiles_run.sh r05c_k31q20 > gpurun_out/r05z/prof_k31.log 2>&1
BENCH_FULL_PATH=gpurun_out/r05z/bench_full.json python bench.py > gpurun_out/r05z/bench.json 2> gpurun_out/r05z/bench.err
echo "bench rc=$?"; wc -c gpurun_out/r05z/bench.json
BENCH_FULL_PATH=gpurun_out/r05z/forcemerge_full.json python bench.py --force-merge --steps 5 --warmup 1 > gpurun_out/r05z/forcemerge.json 2> gpurun_out/r05z/forcemerge.err
BENCH_FULL_PATH=gpurun_out/r05z/group4_full.json python bench.py --group 4 --reads 25000000 --steps 3 --warmup 1 > gpurun_out/r05z/group4.json 2> gpurun_out/r05z/group4.err
head -c 900 gpurun_out/r05z/bench.json; echo; tail -c 600 gpurun_out/r05z/forcemerge.json; echo; tail -c 900 gpurun_out/r05z/group4.json
python tools/skew_probe.py > gpurun_out/r05z/skew_probe.txt 2>&1; tail -7 gpurun_out/r05z/skew_probe.txt
