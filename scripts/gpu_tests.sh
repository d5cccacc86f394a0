#!/bin/bash
# the full -m gpu suite, as the driver runs it
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=40 ) > gpurun_out/gpu_tests.log 2>&1
tail -40 gpurun_out/gpu_tests.log
