#!/bin/bash
# Round 6, second call: the reworked suite (rank pool, explicit order) with every duration; the lines kernel A/B through the API.
cd "$(dirname "$0")/.."
O=gpurun_out/r06_second; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
T=$PWD/cudecomp_amd/lib_tuning/libcudecomp.so
( timeout 100 python scripts/probe/window_walk_ab.py 2>&1 | grep "^{" ) > $O/lines_ab.jsonl
( CUDECOMP_PRESERVE_OUTPUT_HALOS=1 timeout 100 python scripts/probe/window_walk_ab.py 2>&1 | grep "^{" ) >> $O/lines_ab.jsonl
( CUDECOMP_AMD_LIBRARY=$T CUDECOMP_LINES_UNIT=64 timeout 100 python scripts/probe/window_walk_ab.py 2>&1 | grep "^{" ) >> $O/lines_ab.jsonl
( CUDECOMP_AMD_LIBRARY=$T CUDECOMP_LINES_RUN_KIB=64 timeout 100 python scripts/probe/window_walk_ab.py 2>&1 | grep "^{" ) >> $O/lines_ab.jsonl
( CUDECOMP_AMD_LIBRARY=$T CUDECOMP_LINES_RUN_KIB=1024 timeout 100 python scripts/probe/window_walk_ab.py 2>&1 | grep "^{" ) >> $O/lines_ab.jsonl
( CUDECOMP_AMD_LIBRARY=$T CUDECOMP_LINES_RUN_KIB=100000 timeout 100 python scripts/probe/window_walk_ab.py 2>&1 | grep "^{" ) >> $O/lines_ab.jsonl
cut -c1-500 $O/lines_ab.jsonl
( time timeout 1500 python -m pytest tests -q -m gpu --durations=0 --junitxml=$O/junit.xml ) > $O/gpu_suite.log 2>&1; tail -8 $O/gpu_suite.log | cut -c1-300
grep -E "^(FAILED|ERROR)" $O/gpu_suite.log | head -20
