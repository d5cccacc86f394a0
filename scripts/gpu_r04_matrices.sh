#!/bin/bash
# Round 4: the reference runner's own case matrices at HEAD (library defaults: small exchanges fused, pooled workspaces):
# 26,454 cases on 4 ranks (C++), 20,460 on 8 ranks, 26,454 through the Fortran twins; and the transpose configurations once
# more with the two-hop relay switched on (it applies to the NVSHMEM enum on the 2 x 2 grids of the 4-rank matrix).
cd "$(dirname "$0")/.."
O=gpurun_out/r04_matrices
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(make -s -j16 -C cudecomp_amd && make -s -j8 -C tests/native all && make -s -C tests/shim && (command -v amdflang > /dev/null && make -s -C fortran all tests || true)) > $O/build.log 2>&1 || { echo "build failed"; tail -20 $O/build.log; exit 1; }
( time timeout 1200 python tests/test_gpu_runner_cases.py --full ) > $O/reference_sweep_full.log 2>&1; tail -5 $O/reference_sweep_full.log
( time timeout 1500 python tests/test_gpu_runner_cases.py --full --ngpu8 ) > $O/reference_sweep_full_ngpu8.log 2>&1; tail -5 $O/reference_sweep_full_ngpu8.log
( time CUDECOMP_TWO_HOP_RELAY=1 RUNNER_ONLY_TRANSPOSES=1 timeout 1200 python tests/test_gpu_runner_cases.py --full ) > $O/reference_sweep_transposes_relay_on.log 2>&1; tail -5 $O/reference_sweep_transposes_relay_on.log
( time timeout 2400 python tests/test_gpu_runner_cases.py --full-fortran ) > $O/reference_sweep_fortran_full.log 2>&1; tail -5 $O/reference_sweep_fortran_full.log
