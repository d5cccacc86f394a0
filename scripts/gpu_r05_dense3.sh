#!/bin/bash
# round 5: what rows_dense_kernel gains (probe: library dispatch with / without the planner's word, and the public API)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_dense3; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD

( timeout 300 scripts/tune/partial_probe ) > $O/partial_probe_library.log 2>&1; grep -A5 "pitch 8208, dst +8 B, src +8" $O/partial_probe_library.log; grep library: $O/partial_probe_library.log
