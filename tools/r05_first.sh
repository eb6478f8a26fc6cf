#!/bin/bash
# round 5, first GPU call: the stand-alone RCCL probe (both RCCL builds on the image), the new tests, one driver-style bench
mkdir -p gpurun_out/r05a
TL=$(python -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
( cd tools/rccl && timeout 300 ./rccl_big_msg ) > gpurun_out/r05a/rccl_big_msg_rocm.txt 2>&1
echo "rc=$?" >> gpurun_out/r05a/rccl_big_msg_rocm.txt
( cd tools/rccl && LD_PRELOAD=$TL/librccl.so LD_LIBRARY_PATH=$TL timeout 300 ./rccl_big_msg ) > gpurun_out/r05a/rccl_big_msg_torch.txt 2>&1
echo "rc=$?" >> gpurun_out/r05a/rccl_big_msg_torch.txt
timeout 900 python -m pytest tests/test_gpu_exchange.py -x -q -m gpu > gpurun_out/r05a/t_exchange.log 2>&1
echo "rc=$?" >> gpurun_out/r05a/t_exchange.log
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_docs_drift.py "tests/test_gpu_parity.py::test_sample_sized_table_that_does_not_fit_the_room" -x -q -m gpu --durations=15 > gpurun_out/r05a/t_dist.log 2>&1
echo "rc=$?" >> gpurun_out/r05a/t_dist.log
BENCH_FULL_PATH=gpurun_out/r05a/bench_full.json timeout 900 python bench.py > gpurun_out/r05a/bench.json 2> gpurun_out/r05a/bench.err
echo "bench rc=$?"
tail -c 1500 gpurun_out/r05a/bench.json
tail -3 gpurun_out/r05a/t_exchange.log gpurun_out/r05a/t_dist.log
cat gpurun_out/r05a/rccl_big_msg_rocm.txt gpurun_out/r05a/rccl_big_msg_torch.txt
