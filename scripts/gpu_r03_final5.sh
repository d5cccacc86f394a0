#!/bin/bash
# the full -m gpu suite as the driver runs it (library defaults: 120-s peer timeout, runtime-default queues), then smoke
mkdir -p gpurun_out/final5
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/final5
( time timeout 2000 python -m pytest tests -x -q -m gpu --durations=8 -p no:cacheprovider ) > $O/gpu_tests.log 2>&1
tail -6 $O/gpu_tests.log | cut -c1-200
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-200
