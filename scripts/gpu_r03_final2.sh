#!/bin/bash
# tuning variants of the window kernel, the full -m gpu suite, smoke, then the reference's full case matrices
mkdir -p gpurun_out/final2
export HSA_ENABLE_IPC_MODE_LEGACY=0
export CUDECOMP_PEER_TIMEOUT=30
O=gpurun_out/final2
( TUNE_FROM=9 timeout 300 scripts/tune/tune_misaligned ) > $O/tune_misaligned.log 2>&1; grep -E "==|512thr|64B walk j-first|lib \(" $O/tune_misaligned.log | cut -c1-120
( time timeout 2400 python -m pytest tests -q -m gpu --durations=12 -p no:cacheprovider ) > $O/gpu_tests.log 2>&1
tail -8 $O/gpu_tests.log | cut -c1-200
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-200
( time timeout 1500 python tests/test_gpu_runner_cases.py --full --ngpu8 ) > $O/reference_sweep_full_ngpu8.log 2>&1; tail -2 $O/reference_sweep_full_ngpu8.log | cut -c1-200
( time timeout 1200 python tests/test_gpu_runner_cases.py --full ) > $O/reference_sweep_full.log 2>&1; tail -2 $O/reference_sweep_full.log | cut -c1-200
( time timeout 1200 python tests/test_gpu_runner_cases.py --full-fortran ) > $O/reference_sweep_fortran_full.log 2>&1; tail -2 $O/reference_sweep_fortran_full.log | cut -c1-200
