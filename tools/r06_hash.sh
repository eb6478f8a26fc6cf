#!/bin/bash
# round 6: the two-instruction Feistel rounds (new table hash) -- parity first, then same-box A/B against the old-hash build
O=gpurun_out/r06hash; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_product_lib.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/parity.log
grep -E "passed|failed" $O/parity.log | tail -2
rm -f gpurun_out/ab_libs.txt
for rep in 1 2; do
bash tools/ab_libs.sh "libkmerhip.so libkmerhip_oldhash.so" "--k 21|--k 31 --min-quality 20 --no-hint|--k 25|--k 17" > /dev/null
done
cp gpurun_out/ab_libs.txt $O/ab_libs.txt; cat $O/ab_libs.txt
