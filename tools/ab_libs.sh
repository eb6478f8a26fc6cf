#!/bin/bash
# A/B of builds of the library on the GPU box: usage tools/ab_libs.sh "<lib> <lib> ..." "<bench args>|<bench args>|..."
# (libs from `make -C krust_amd/csrc VARIANT=_x EXTRA=-D...`).  Appends to gpurun_out/ab_libs.txt.
out=gpurun_out/ab_libs.txt
IFS='|' read -ra CFGS <<< "${2:---k 21}"
for lib in $1; do
  for cfg in "${CFGS[@]}"; do
    KMERHIP_LIB=$lib python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-verify $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', '$cfg', 'ms_per_step', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['roofline']['stages_ms'].items()})" >> $out
  done
done
cat $out
