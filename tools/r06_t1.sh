#!/bin/bash
O=gpurun_out/r06a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_product_lib.py -x -q -m gpu > $O/exch2.log 2>&1; echo "rc=$?" >> $O/exch2.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "merge or shard or export or region" > $O/parity2.log 2>&1; echo "rc=$?" >> $O/parity2.log
tail -n 15 $O/exch2.log; tail -n 15 $O/parity2.log
