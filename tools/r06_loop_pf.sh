#!/bin/bash
# round 6: the probing loop reads its next item while the current one is probed (KH_REGION_LOOP_PREFETCH) -- parity, then same-box A/B
O=gpurun_out/r06pf; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/parity.log
grep -E "passed|failed" $O/parity.log | tail -2
rm -f gpurun_out/ab_libs.txt
for rep in 1 2; do
bash tools/ab_libs.sh "libkmerhip.so libkmerhip_nopf.so" "--k 21|--reads 125000000|--k 25|--reads 10000000" > /dev/null
done
cp gpurun_out/ab_libs.txt $O/ab_libs.txt; cat $O/ab_libs.txt
