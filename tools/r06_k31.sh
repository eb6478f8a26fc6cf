#!/bin/bash
# round 6: the 8-byte payload as the hashed remainder -- the whole GPU suite, then configs[2] / k = 25 / the headline, and the wave-skip A/B
O=gpurun_out/r06g; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc=$?" >> $O/gpu_suite.log
tail -n 4 $O/gpu_suite.log
rm -f gpurun_out/ab_libs.txt
bash tools/ab_libs.sh "libkmerhip.so libkmerhip_wskip.so libkmerhip.so libkmerhip_wskip.so" "--k 31 --min-quality 20" > /dev/null 2>&1
bash tools/ab_libs.sh "libkmerhip.so" "--k 25|--k 21|--k 22 --min-quality 20" > /dev/null 2>&1
cp gpurun_out/ab_libs.txt $O/ab_libs.txt; cat $O/ab_libs.txt
