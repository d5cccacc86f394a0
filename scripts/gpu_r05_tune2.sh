#!/bin/bash
# round 5, tuning call 2: refined tile / walk scan per element size, more partial-probe cases, the r04 code-size A/B
cd "$(dirname "$0")/.."
O=gpurun_out/r05_tune2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 300 scripts/tune/partial_probe ) > $O/partial_probe.log 2>&1
( timeout 400 scripts/tune/tune_fwd 8 10 2 ) > $O/tune_fwd_8_phase2.log 2>&1
( timeout 300 scripts/tune/tune_fwd 16 10 2 ) > $O/tune_fwd_16_phase2.log 2>&1
( timeout 300 scripts/tune/tune_fwd 4 10 2 ) > $O/tune_fwd_4.log 2>&1
( timeout 600 python scripts/probe/ab_speed.py ) > $O/ab_speed_code_size.log 2>&1
tail -5 $O/ab_speed_code_size.log
