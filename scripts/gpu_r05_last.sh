#!/bin/bash
# round 5, last call: smoke() as the driver runs it, then the -m gpu suite a second time at HEAD
cd "$(dirname "$0")/.."
O=gpurun_out/r05_last; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log | cut -c1-200
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/gpu_suite_2.log 2>&1; tail -6 $O/gpu_suite_2.log | cut -c1-200
