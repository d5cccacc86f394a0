#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_transpose.py -x -q -m gpu 2>&1 | tail -8
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-sample 0
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --layout default
done
} > gpurun_out/quick.log 2>&1
cat gpurun_out/quick.log
