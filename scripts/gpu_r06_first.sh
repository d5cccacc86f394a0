#!/bin/bash
# Round 6, first call: the -m gpu suite at HEAD with EVERY test's duration (input to the suite rework), then the bench line.
cd "$(dirname "$0")/.."
O=gpurun_out/r06_first; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
nproc > $O/nproc.txt; free -g >> $O/nproc.txt
( time timeout 1500 python -m pytest tests -q -m gpu --durations=0 --junitxml=$O/junit.xml ) > $O/gpu_suite.log 2>&1; tail -5 $O/gpu_suite.log | cut -c1-200
( time timeout 400 python bench.py ) > $O/bench.log 2>&1; grep -E '^\{' $O/bench.log | tail -1 > $O/bench_n1.json
python -c "import json; r = json.load(open('$O/bench_n1.json')); print(r['ms_per_step'], r['roofline']['frac'], json.dumps(r['extra'].get('halo_pencil_transposes'))[:800])"
