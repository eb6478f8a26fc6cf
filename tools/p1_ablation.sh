#!/bin/bash
# Where does part1_scatter_chunked_kernel (the level-1 kernel of 64-bit payloads; 32-bit ones take part1_bins_kernel) spend its time?  Builds the library with ablation bits (KH_ABL: the
# kernel then produces garbage, so the pipeline stops after level 1) and prints the level-1 stage time of
# bench.py for each.  Run the build part here (no GPU needed), the timing part through gpurun:
#   bash tools/p1_ablation.sh build     -> krust_amd/lib/libkmerhip_abl<N>.so
#   bash tools/p1_ablation.sh run       -> gpurun_out/p1_ablation.txt
set -u
BITS="0 1 2 4 8 16 9 5 13"   # 1 no hash, 2 no global stores, 4 no write-out, 8 no extraction, 16 no rank atomics
if [ "${1:-build}" = build ]; then
  for b in $BITS; do make -s -C krust_amd/csrc VARIANT=_abl$b EXTRA="-DKH_ABL=$b" || exit 1; done
else
  mkdir -p gpurun_out; : > gpurun_out/p1_ablation.txt
  for b in $BITS; do
    [ $b = 0 ] || export KMERHIP_STOP_AFTER_P1=1
    KMERHIP_LIB=libkmerhip_abl$b.so python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('KH_ABL=$b', 'level1_ms', round(d['roofline']['stages_ms'].get('level1',0),2))" >> gpurun_out/p1_ablation.txt
  done
  cat gpurun_out/p1_ablation.txt
fi
