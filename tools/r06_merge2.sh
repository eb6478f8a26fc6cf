#!/bin/bash
O=gpurun_out/r06b; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_dist.py -x -q -m gpu > $O/exch.log 2>&1; echo "exchange rc=$?" >> $O/exch.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "merge or shard or export or region" > $O/parity_merge.log 2>&1; echo "parity rc=$?" >> $O/parity_merge.log
BENCH_FULL_PATH=$O/forcemerge_full.json timeout 900 python bench.py --force-merge --steps 5 --warmup 1 --no-extras --no-cpu-baseline > $O/forcemerge.json 2> $O/forcemerge.err
BENCH_FULL_PATH=$O/group4_full.json timeout 900 python bench.py --group 4 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/group4.json 2> $O/group4.err
rm -rf $O/prof_fm
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fm -- python3 bench.py --force-merge --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-verify > $O/prof_fm.log 2>&1
find $O/prof_fm -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/fm_kernel_stats.csv
find $O/prof_fm -type f -size +4M -delete
tail -n 3 $O/exch.log; tail -n 3 $O/parity_merge.log
head -12 $O/fm_kernel_stats.csv | cut -c1-100,200-330
