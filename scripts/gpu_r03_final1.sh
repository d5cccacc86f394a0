#!/bin/bash
# A/B of the flag location on the stress (warm), the full -m gpu suite, then the reference's full case matrices
mkdir -p gpurun_out/final1
export HSA_ENABLE_IPC_MODE_LEGACY=0
export CUDECOMP_PEER_TIMEOUT=30
O=gpurun_out/final1
( time timeout 300 python scripts/probe/stress_eight_ranks.py mix 10 CUDECOMP_FLAGS_IN_HOST_MEMORY=1 ) > $O/stress_host_a.log 2>&1; grep "iterations failed" $O/stress_host_a.log | cut -c1-200
( time timeout 300 python scripts/probe/stress_eight_ranks.py mix 10 ) > $O/stress_dev_a.log 2>&1; grep "iterations failed" $O/stress_dev_a.log | cut -c1-200
( time timeout 300 python scripts/probe/stress_eight_ranks.py mix 10 CUDECOMP_FLAGS_IN_HOST_MEMORY=1 ) > $O/stress_host_b.log 2>&1; grep "iterations failed" $O/stress_host_b.log | cut -c1-200
( time timeout 2400 python -m pytest tests -q -m gpu --durations=12 -p no:cacheprovider ) > $O/gpu_tests.log 2>&1
tail -22 $O/gpu_tests.log | cut -c1-200
( time timeout 900 python tests/test_gpu_runner_cases.py --full ) > $O/reference_sweep_full.log 2>&1; tail -2 $O/reference_sweep_full.log
( time timeout 1200 python tests/test_gpu_runner_cases.py --full --ngpu8 ) > $O/reference_sweep_full_ngpu8.log 2>&1; tail -2 $O/reference_sweep_full_ngpu8.log
( time timeout 900 python tests/test_gpu_runner_cases.py --full-fortran ) > $O/reference_sweep_fortran_full.log 2>&1; tail -2 $O/reference_sweep_fortran_full.log
