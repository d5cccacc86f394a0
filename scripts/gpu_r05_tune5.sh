#!/bin/bash
# round 5, tuning call 5: forward hops, phase 5 of scripts/tune/tune_fwd.hip (store cache bits, persistent workgroups with
# prefetch, occupancy limits, longer destination segments under the run walk, far-pitch padding = DRAM channel aliasing probe)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_tune5; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 400 scripts/tune/tune_fwd 8 10 5 ) > $O/tune_fwd_fp64_phase5.log 2>&1
grep -c WRONG $O/tune_fwd_fp64_phase5.log
head -12 $O/tune_fwd_fp64_phase5.log | cut -c1-220
tail -48 $O/tune_fwd_fp64_phase5.log | cut -c1-200
