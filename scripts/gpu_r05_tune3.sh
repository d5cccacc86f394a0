#!/bin/bash
# round 5, call 3: the library with the batch-run walk (kernel tests + harness), code-size bisect on the library itself
cd "$(dirname "$0")/.."
O=gpurun_out/r05_tune3; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_transpose.py -x -q -m gpu ) > $O/kernel_tests.log 2>&1; tail -3 $O/kernel_tests.log
( timeout 300 scripts/tune/tune_fwd 8 10 3 ) > $O/tune_fwd_8_phase3.log 2>&1; head -8 $O/tune_fwd_8_phase3.log
( timeout 120 scripts/tune/tune_fwd 16 5 9 ) > $O/tune_fwd_16_lib.log 2>&1; head -8 $O/tune_fwd_16_lib.log | tail -7
( timeout 120 scripts/tune/tune_fwd 4 5 9 ) 2>&1 | head -8 | tail -7 > $O/tune_fwd_4_lib.log; cat $O/tune_fwd_4_lib.log
( timeout 900 bash scripts/probe/code_size_bisect.sh run ) > $O/code_size_bisect.log 2>&1; cat $O/code_size_bisect.log
