#!/bin/bash
# first GPU contact: smoke, kernel parity, single-rank transposes, a short bench
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
echo "== smoke"; timeout 600 python __graft_entry__.py smoke
echo "== kernels"; timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -15
echo "== transposes (single rank)"; timeout 900 python -m pytest tests/test_gpu_transpose.py -x -q -m gpu -k "single_rank" 2>&1 | tail -15
echo "== halo (single rank)"; timeout 600 python -m pytest tests/test_gpu_halo.py -x -q -m gpu -k "single_rank" 2>&1 | tail -15
echo "== bench"; timeout 900 python bench.py --steps 5 --warmup 2 --cpu-sample 256
echo "== bench default layout"; timeout 900 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --layout default
} > gpurun_out/first.log 2>&1
tail -60 gpurun_out/first.log
