#!/bin/bash
# kernel trace of the N > 1 step's one-GPU pre-flight (bench.py --force-merge): what the merge leg's kernels take
O=$PWD/gpurun_out/r06a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_product_lib.py -x -q -m gpu > $O/prod.log 2>&1; echo "rc=$?" >> $O/prod.log; tail -n 4 $O/prod.log
rm -rf $O/prof_fm
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fm -- python3 bench.py --force-merge --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-verify > $O/prof_fm.log 2>&1
find $O/prof_fm -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/fm_kernel_stats.csv
find $O/prof_fm -type f -size +8M -delete
head -30 $O/fm_kernel_stats.csv | cut -c1-220
