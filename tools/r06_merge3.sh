#!/bin/bash
O=gpurun_out/r06f; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_dist.py tests/test_gpu_product_lib.py -x -q -m gpu > $O/exch.log 2>&1; echo "exchange rc=$?" >> $O/exch.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "merge or shard or export or region" > $O/parity_merge.log 2>&1; echo "parity rc=$?" >> $O/parity_merge.log
BENCH_FULL_PATH=$O/forcemerge_full.json timeout 900 python bench.py --force-merge --steps 5 --warmup 1 --no-extras --no-cpu-baseline > $O/forcemerge.json 2> $O/forcemerge.err
BENCH_FULL_PATH=$O/group4_full.json timeout 900 python bench.py --group 4 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/group4.json 2> $O/group4.err
tail -n 3 $O/exch.log; tail -n 3 $O/parity_merge.log
