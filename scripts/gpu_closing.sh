#!/bin/bash
# The closing run of a round: the -m gpu suite as the driver runs it (-x), the opt-in arms, smoke(), the bench line.
# Everything lands under gpurun_out/closing/; copy what is to be judged into profiles/.
cd "$(dirname "$0")/.."
O=gpurun_out/closing; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=25 --junitxml=$O/junit.xml ) > $O/gpu_suite.log 2>&1; tail -6 $O/gpu_suite.log | cut -c1-300
grep -E "^(FAILED|ERROR)" $O/gpu_suite.log | head -20
( time CUDECOMP_TEST_EXTENDED=1 timeout 1200 python -m pytest tests -q -m "gpu and extended" -rA --durations=10 ) > $O/gpu_suite_extended.log 2>&1; tail -4 $O/gpu_suite_extended.log | cut -c1-250
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-300
( time timeout 400 python bench.py ) > $O/bench.log 2>&1; grep -E '^\{' $O/bench.log | tail -1 > $O/bench_n1.json
python -c "import json; r = json.load(open('$O/bench_n1.json')); print(r['ms_per_step'], r['roofline']['frac'], r['stats'].get('in_place_cycle_ms'))"
