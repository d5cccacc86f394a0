#!/bin/bash
# round 5: the library after the kernels were split over several code objects and the run walk went in
cd "$(dirname "$0")/.."
O=gpurun_out/r05_validate; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_transpose.py tests/test_gpu_native_sweep.py tests/test_gpu_halo.py -x -q -m gpu ) > $O/tests.log 2>&1; tail -4 $O/tests.log
for es in 8 16 4; do ( timeout 200 scripts/tune/tune_fwd $es 10 9 ) 2>&1 | head -7 > $O/tune_fwd_${es}_lib.log; cat $O/tune_fwd_${es}_lib.log | cut -c1-200; done
( timeout 600 python bench.py --steps 10 --warmup 3 ) > $O/bench.log 2>&1; grep -E '^\{' $O/bench.log | tail -1 | cut -c1-600
( timeout 300 python scripts/probe/ab_speed.py ) > $O/ab_speed.log 2>&1; grep "ms per case" $O/ab_speed.log
