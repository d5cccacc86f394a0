#!/bin/bash
# the full -m gpu suite (no early exit), then the stress with default queue settings
mkdir -p gpurun_out/suite
export HSA_ENABLE_IPC_MODE_LEGACY=0
export CUDECOMP_PEER_TIMEOUT=${CUDECOMP_PEER_TIMEOUT:-30}
O=gpurun_out/suite
( time timeout 3000 python -m pytest tests -q -m gpu --durations=15 -p no:cacheprovider ) > $O/gpu_tests.log 2>&1
tail -30 $O/gpu_tests.log | cut -c1-200
( time timeout 600 python scripts/probe/stress_eight_ranks.py mix 40 ) > $O/stress_mix.log 2>&1
tail -2 $O/stress_mix.log
