#!/bin/bash
# round 5, tuning call 1: forward-hop tile shapes / walks, isolated vs sustained launches, partial-unit probe, code-size probe
cd "$(dirname "$0")/.."
O=gpurun_out/r05_tune1; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 300 scripts/tune/partial_probe ) > $O/partial_probe.log 2>&1
( timeout 400 scripts/tune/tune_fwd 8 10 ) > $O/tune_fwd_8.log 2>&1
( timeout 300 scripts/tune/tune_fwd 16 10 ) > $O/tune_fwd_16.log 2>&1
( timeout 600 bash scripts/probe/code_size_probe.sh ) > $O/code_size_probe.log 2>&1
tail -12 $O/partial_probe.log; head -8 $O/tune_fwd_8.log
