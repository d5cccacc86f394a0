#!/bin/bash
# round 5: the reference runner's full case matrices at HEAD (kernels in six code objects, gated twins) and the device-memory
# flags on the 4-rank matrix, twice
cd "$(dirname "$0")/.."
O=gpurun_out/r05_matrices; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python tests/test_gpu_runner_cases.py --full ) > $O/reference_sweep_full.log 2>&1; tail -2 $O/reference_sweep_full.log | cut -c1-200
( time timeout 1200 python tests/test_gpu_runner_cases.py --full --ngpu8 ) > $O/reference_sweep_full_ngpu8.log 2>&1; tail -2 $O/reference_sweep_full_ngpu8.log | cut -c1-200
for rep in 1 2; do
  ( time CUDECOMP_FLAGS_IN_DEVICE_MEMORY=1 timeout 1500 python tests/test_gpu_runner_cases.py --full ) > $O/flags_device_matrix_4ranks_run$rep.log 2>&1
  tail -2 $O/flags_device_matrix_4ranks_run$rep.log | cut -c1-200
done
( time timeout 1500 python tests/test_gpu_runner_cases.py --full-fortran ) > $O/reference_sweep_fortran_full.log 2>&1; tail -2 $O/reference_sweep_fortran_full.log | cut -c1-200
grep -h -E "DIAG|FAILED" $O/*.log | head -20
