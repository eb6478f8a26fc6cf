#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun).
# usage: bash tools/profiles_run.sh <tag>     -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-verify ${BENCH_ARGS:-}"   # BENCH_ARGS: e.g. "--k 31 --min-quality 20"
# 1) per-kernel time
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
# 2) counters, each in its own pass (TCC has 4 slots: FETCH_SIZE=3, WRITE_SIZE=2)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 bench.py $ARGS > $OUT/bench_l2.log 2>&1
rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/pmc_ea -- python3 bench.py $ARGS > $OUT/bench_ea.log 2>&1
rocprofv3 -L 2>/dev/null | grep -E "TCC_(EA0_)?(ATOMIC|RDREQ|WRREQ|HIT|MISS|REQ|READ|WRITE)" | head -60 > $OUT/tcc_counters.txt
# drop the bulky per-dispatch traces bigger than 8 MiB, keep stats
find $OUT -type f -size +8M -delete
ls -R $OUT | head -80
