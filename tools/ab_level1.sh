#!/bin/bash
# A/B of the level-1 bin layout on the GPU box: default build (16-byte units of bin p kept in the order u ^ (p & 7)) against
# libkmerhip_nosw.so (make -C krust_amd/csrc VARIANT=_nosw EXTRA=-DKH_L1_SWIZZLE=0), headline and configs[2].
# Writes gpurun_out/ab_level1.txt.
out=gpurun_out/ab_level1.txt
: > $out
for lib in libkmerhip.so libkmerhip_nosw.so; do
  for cfg in "--k 21" "--k 31 --min-quality 20" "--k 19" "--k 25"; do
    KMERHIP_LIB=$lib python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-verify $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', '$cfg', 'ms_per_step', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['roofline']['stages_ms'].items()})" >> $out
  done
done
cat $out
