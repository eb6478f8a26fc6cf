#!/bin/bash
# round 6, final measurements: profiles of the headline command and of the other configurations, a driver-style bench run,
# the count-and-merge pre-flights, SQ counters (headline, configs[2], the merge leg)
mkdir -p gpurun_out/r06z
bash tools/profiles_run.sh r06 > gpurun_out/r06z/prof.log 2>&1
bash tools/sq_probe.sh r06 > gpurun_out/r06z/sq.log 2>&1
BENCH_ARGS="--k 31 --min-quality 20 --no-hint" bash tools/sq_probe.sh r06_k31q20 > gpurun_out/r06z/sq_k31q20.log 2>&1
bash tools/sq_probe2.sh r06_fm "shard_merge|region_compact_image" --force-merge --steps 1 --warmup 0 --no-extras --no-cpu-baseline --no-verify > gpurun_out/r06z/sq_fm.log 2>&1
BENCH_ARGS="--reads 125000000" bash tools/profiles_run.sh r06_s125 > gpurun_out/r06z/prof_s125.log 2>&1
BENCH_ARGS="--hg" bash tools/profiles_run.sh r06_hg > gpurun_out/r06z/prof_hg.log 2>&1
BENCH_ARGS="--k 31 --min-quality 20 --no-hint" bash tools/profiles_run.sh r06_k31q20 > gpurun_out/r06z/prof_k31.log 2>&1
export TMPDIR=/tmp
rm -rf gpurun_out/r06z/prof_fm
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06z/prof_fm -- python3 bench.py --force-merge --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-verify > gpurun_out/r06z/prof_fm.json 2> gpurun_out/r06z/prof_fm.err
find gpurun_out/r06z/prof_fm -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r06z/fm_kernel_stats.csv
rm -rf gpurun_out/r06z/prof_fm
BENCH_FULL_PATH=gpurun_out/r06z/bench_full.json python bench.py > gpurun_out/r06z/bench.json 2> gpurun_out/r06z/bench.err
echo "bench rc=$?"; wc -c gpurun_out/r06z/bench.json
BENCH_FULL_PATH=gpurun_out/r06z/forcemerge_full.json python bench.py --force-merge --steps 5 --warmup 1 > gpurun_out/r06z/forcemerge.json 2> gpurun_out/r06z/forcemerge.err
BENCH_FULL_PATH=gpurun_out/r06z/group4_full.json python bench.py --group 4 --reads 25000000 --steps 3 --warmup 1 > gpurun_out/r06z/group4.json 2> gpurun_out/r06z/group4.err
head -c 900 gpurun_out/r06z/bench.json; echo; tail -c 600 gpurun_out/r06z/forcemerge.json; echo; tail -c 900 gpurun_out/r06z/group4.json
python tools/skew_probe.py > gpurun_out/r06z/skew_probe.txt 2>&1; tail -7 gpurun_out/r06z/skew_probe.txt
python tools/host_push_probe.py > gpurun_out/r06z/host_push_probe.txt 2>&1; tail -5 gpurun_out/r06z/host_push_probe.txt
