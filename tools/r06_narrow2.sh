#!/bin/bash
O=gpurun_out/r06i; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_gpu_text.py tests/test_gpu_scale.py -x -q -m gpu > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -n 3 $O/t.log
