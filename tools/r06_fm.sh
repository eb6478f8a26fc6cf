#!/bin/bash
# quick look at the merge leg: bench.py --force-merge under the kernel trace; $1 = output tag
O=gpurun_out/r06_$1; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_exchange.py -x -q -m gpu -k "world1 or sharing or lost or conserved" > $O/exch.log 2>&1; echo "exchange rc=$?" >> $O/exch.log
rm -rf $O/prof_fm
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fm -- python3 bench.py --force-merge --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-verify > $O/prof_fm.json 2> $O/prof_fm.err
find $O/prof_fm -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/fm_kernel_stats.csv
rm -rf $O/prof_fm
tail -n 3 $O/exch.log
grep -E "shard_merge|region_compact|unit_digest|digest_reduce|rccl|region_count_kernel32|head_count" $O/fm_kernel_stats.csv | cut -c1-60,180-330
tail -c 900 $O/prof_fm.json
