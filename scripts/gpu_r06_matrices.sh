#!/bin/bash
# Round 6: the reference runner's complete case matrices at HEAD (C++ twins on 4 and 8 ranks, Fortran twins on 4), and the
# extended arms of the -m gpu suite (CUDECOMP_TEST_EXTENDED=1).
cd "$(dirname "$0")/.."
O=gpurun_out/r06_matrices; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
( time timeout 1500 python tests/test_gpu_runner_cases.py --full ) > $O/reference_sweep_full.log 2>&1; tail -4 $O/reference_sweep_full.log | cut -c1-200
( time timeout 1200 python tests/test_gpu_runner_cases.py --full --ngpu8 ) > $O/reference_sweep_full_ngpu8.log 2>&1; tail -4 $O/reference_sweep_full_ngpu8.log | cut -c1-200
( time timeout 1500 python tests/test_gpu_runner_cases.py --full-fortran ) > $O/reference_sweep_fortran_full.log 2>&1; tail -4 $O/reference_sweep_fortran_full.log | cut -c1-200
( time CUDECOMP_TEST_EXTENDED=1 timeout 900 python -m pytest tests -q -m "gpu and extended" -rA --durations=10 ) > $O/gpu_suite_extended.log 2>&1; tail -25 $O/gpu_suite_extended.log | cut -c1-250
( time CUDECOMP_TEST_EXTENDED=1 timeout 900 python -m pytest tests/test_gpu_native_sweep.py tests/test_gpu_runner_cases.py -q -m gpu -k "switches or eight_ranks" ) > $O/gpu_suite_extended_denser_slices.log 2>&1; tail -5 $O/gpu_suite_extended_denser_slices.log | cut -c1-250
