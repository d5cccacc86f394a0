#!/bin/bash
# Round 4, closing check at HEAD (after the last library change: the device epoch's initial value is synchronised): the
# test files that exercise epochs, graphs, halos and the native programs; then as much of the Fortran matrix as fits.
cd "$(dirname "$0")/.."
O=gpurun_out/r04_last
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(make -s -j16 -C cudecomp_amd && make -s -j16 -C cudecomp_amd MPI=1 && make -s -j8 -C tests/native all mpi && make -s -C tests/shim && (command -v amdflang > /dev/null && make -s -C fortran all tests || true)) > $O/build.log 2>&1 || { echo "build failed"; tail -20 $O/build.log; exit 1; }
( time timeout 500 python -m pytest tests/test_gpu_native.py tests/test_gpu_async.py tests/test_gpu_graphs.py tests/test_gpu_halo.py tests/test_gpu_self_exchange.py tests/test_gpu_relay.py tests/test_gpu_queue_census.py tests/test_gpu_failure_detection.py -x -q -m gpu ) > $O/subset.log 2>&1; tail -4 $O/subset.log | cut -c1-200
( time timeout 420 python tests/test_gpu_runner_cases.py --full-fortran ) > $O/reference_sweep_fortran_partial.log 2>&1; grep -c "cases passed" $O/reference_sweep_fortran_partial.log; tail -3 $O/reference_sweep_fortran_partial.log | cut -c1-200
