#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
echo "== smoke (faulthandler)"; timeout 600 python -X faulthandler -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -40
echo "== kernels"; timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -15
echo "== multi-rank transposes"; timeout 1500 python -m pytest tests/test_gpu_transpose.py -x -q -m gpu -k "multi_rank or rccl or benchmark" 2>&1 | tail -40
echo "== multi-rank halos"; timeout 900 python -m pytest tests/test_gpu_halo.py -x -q -m gpu -k "multi_rank" 2>&1 | tail -40
} > gpurun_out/second.log 2>&1
tail -120 gpurun_out/second.log
