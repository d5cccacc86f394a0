#!/bin/bash
# round 5, VERDICT item 3: the device-memory flags on the reference's full 4-rank matrix (gated twins), twice, and on the 8-rank one
cd "$(dirname "$0")/.."
O=gpurun_out/r05_flags; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do
  ( time CUDECOMP_FLAGS_IN_DEVICE_MEMORY=1 timeout 1500 python tests/test_gpu_runner_cases.py --full ) > $O/matrix_4ranks_device_flags_run$rep.log 2>&1
  grep -c "passed in" $O/matrix_4ranks_device_flags_run$rep.log; tail -3 $O/matrix_4ranks_device_flags_run$rep.log | cut -c1-300
done
grep -h -E "DIAG|Input gate.*[1-9] stale|FAILED" $O/matrix_4ranks_device_flags_run*.log | head -20
