#!/bin/bash
# round 6: the written-out first-probe step of the region pass (KH_REGION_FP_ASM) -- parity first, then same-box A/B against the C++ step
# (the step itself is tools/region_fp_asm.patch: `git apply` it, `make -C krust_amd/csrc`, then `make -C krust_amd/csrc VARIANT=_nofpasm EXTRA=-DKH_REGION_FP_ASM=0`; measured and not kept)
O=gpurun_out/r06fp; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_product_lib.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/parity.log
grep -E "passed|failed" $O/parity.log | tail -2
rm -f gpurun_out/ab_libs.txt
for rep in 1 2; do
bash tools/ab_libs.sh "libkmerhip.so libkmerhip_nofpasm.so" "--k 21|--reads 125000000|--k 25|--reads 10000000" > /dev/null
done
cp gpurun_out/ab_libs.txt $O/ab_libs.txt; cat $O/ab_libs.txt
