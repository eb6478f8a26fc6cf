#!/bin/bash
# which kernels of the k = 25 step changed with the new hash?
O=gpurun_out/r06hash; mkdir -p $O
export TMPDIR=/tmp
for lib in libkmerhip.so libkmerhip_oldhash.so; do
  KMERHIP_LIB=$lib python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-verify --k 25 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['config'], d['roofline'].get('stages_ms'), {k:v for k,v in d.items() if k in ('stats','overflow','ovf')})" | tee -a $O/k25_cfg.txt
  export KMERHIP_LIB=$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$lib -- python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-verify --k 25 > $O/prof_$lib.log 2>&1
  f=$(ls $O/prof_$lib/*/*kernel_stats.csv | head -1); head -8 $f | cut -d, -f1-4 | cut -c1-200 | tee -a $O/k25_cfg.txt
done
find $O -name "*.csv" -size +1M -delete
