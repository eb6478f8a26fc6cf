#!/bin/bash
O=gpurun_out/r06i; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_gpu_text.py -x -q -m gpu > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -n 3 $O/t.log
rm -f gpurun_out/ab_libs.txt
bash tools/ab_libs.sh "libkmerhip.so" "--k 25|--k 22|--k 22 --min-quality 20|--k 31 --min-quality 20|--k 21" > /dev/null 2>&1
KMERHIP_LIB=libkmerhip_testing.so KMERHIP_L2_NARROW=0 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-verify --k 25 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-narrow k25', round(d['ms_per_step'],2), d['roofline']['stages_ms'])" >> gpurun_out/ab_libs.txt
python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --k 25 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('verify k25', d['verify'])" >> gpurun_out/ab_libs.txt
cp gpurun_out/ab_libs.txt $O/ab_libs.txt; cat $O/ab_libs.txt
