#!/bin/bash
# third sample of the full -m gpu suite in the shipped configuration, then the default bench line
mkdir -p gpurun_out/final6
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/final6
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 -p no:cacheprovider ) > $O/gpu_tests.log 2>&1
tail -4 $O/gpu_tests.log | cut -c1-200
( time timeout 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ) 2>&1 | tail -3; cut -c1-300 $O/bench_n1.json
