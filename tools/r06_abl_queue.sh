#!/bin/bash
# round 6, timing experiment (WRONG results, --no-verify): what does the first probe's queue store cost the region pass?
# libkmerhip_abl1.so: -DKH_ABLR=1 (no probing loop); libkmerhip_abl17.so: -DKH_ABLR=17 (no loop and no queue store)
rm -f gpurun_out/ab_libs.txt
for rep in 1 2; do
bash tools/ab_libs.sh "libkmerhip.so libkmerhip_abl1.so libkmerhip_abl17.so" "--k 21" > /dev/null
done
cat gpurun_out/ab_libs.txt
