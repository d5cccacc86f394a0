#!/bin/bash
# Round 6, fourteenth call: the lines / rowlines / rotate tests in full, then the whole suite at HEAD.
cd "$(dirname "$0")/.."
O=gpurun_out/r06_fourteenth; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
( time timeout 600 python -m pytest tests/test_gpu_dense_rows.py tests/test_gpu_kernels.py -q -m gpu ) > $O/kernel_tests.log 2>&1; tail -5 $O/kernel_tests.log | cut -c1-300
grep -E "^(FAILED|ERROR)|^E  " $O/kernel_tests.log | head -12 | cut -c1-300
( time timeout 1500 python -m pytest tests -q -m gpu --durations=25 --junitxml=$O/junit.xml ) > $O/gpu_suite.log 2>&1; tail -6 $O/gpu_suite.log | cut -c1-300
grep -E "^(FAILED|ERROR)" $O/gpu_suite.log | head -20
