#!/bin/bash
O=gpurun_out/r06h; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q -m gpu -k "quality or window or k31 or q20 or stress or ragged or 1024_x or heavy or skew or hot or beyond" > $O/q.log 2>&1; echo "rc=$?" >> $O/q.log
tail -n 3 $O/q.log
rm -f gpurun_out/ab_libs.txt
bash tools/ab_libs.sh "libkmerhip.so libkmerhip.so" "--k 31 --min-quality 20|--k 22 --min-quality 20|--k 25 --min-quality 30" > /dev/null 2>&1
cp gpurun_out/ab_libs.txt $O/ab_libs.txt; cat $O/ab_libs.txt
