#!/bin/bash
# round 6: the region pass's loop variants, same box: default (pulling loop for whole rounds), _nopull (-DKH_REGION_PULL=0), _cp (-DKH_REGION_CLAIM_PARTIAL=1)
LIBS="libkmerhip.so libkmerhip_nopull.so libkmerhip_cp.so"
rm -f gpurun_out/ab_libs.txt gpurun_out/pull_hg.txt
bash tools/ab_libs.sh "$LIBS" "--k 21|--reads 10000000" > /dev/null
for lib in $LIBS $LIBS; do
  KMERHIP_LIB=$lib python bench.py --hg --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib hg', d['ms_per_step'], d['roofline']['stages_ms'], d.get('verify'))" >> gpurun_out/pull_hg.txt
done
cat gpurun_out/ab_libs.txt gpurun_out/pull_hg.txt
