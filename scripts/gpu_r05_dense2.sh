#!/bin/bash
# round 5: rows_dense_kernel -- what it gains (probe), then the parts of the suite that move data onto halo-carrying pencils:
# the native sweeps (the "window_stores_any_size" entry sends every flagged row copy of the reference's matrices through the
# dense kernel), the transposes and the runner cases
cd "$(dirname "$0")/.."
O=gpurun_out/r05_dense2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 300 python scripts/probe/dense_rows_bench.py 10 ) > $O/dense_rows_bench.log 2>&1; tail -3 $O/dense_rows_bench.log | cut -c1-3000
( time timeout 1200 python -m pytest tests/test_gpu_dense_rows.py tests/test_gpu_native_sweep.py tests/test_gpu_transpose.py tests/test_gpu_runner_cases.py -x -q -m gpu --durations=8 ) > $O/tests.log 2>&1; tail -16 $O/tests.log | cut -c1-300
