#!/bin/bash
# round 5: first GPU run of rows_dense_kernel -- its own tests, then the kernel tests that now also pass the whole-rows word
cd "$(dirname "$0")/.."
O=gpurun_out/r05_dense1; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_gpu_dense_rows.py -x -q -m gpu -v --durations=12 ) > $O/dense_tests.log 2>&1; tail -40 $O/dense_tests.log | cut -c1-400
( time timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu --durations=5 ) > $O/kernel_tests.log 2>&1; tail -12 $O/kernel_tests.log | cut -c1-300
