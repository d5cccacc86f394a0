#!/bin/bash
# Round 6, fifteenth call: big random sweeps of the two whole-line permutation kernels (fresh examples), the suite at HEAD, the
# bench line, the reference's 4- and 8-rank matrices at HEAD.
cd "$(dirname "$0")/.."
O=gpurun_out/r06_fifteenth; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
( time CUDECOMP_TEST_SWEEP_RANDOM=1 CUDECOMP_TEST_SWEEP_EXAMPLES=3000 timeout 900 python -m pytest tests/test_gpu_dense_rows.py -q -m gpu -k "random_sweep" ) > $O/random_sweeps.log 2>&1; tail -6 $O/random_sweeps.log | cut -c1-300
grep -E "^(FAILED|ERROR)|^E  " $O/random_sweeps.log | head -12 | cut -c1-300
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=25 --junitxml=$O/junit.xml ) > $O/gpu_suite.log 2>&1; tail -6 $O/gpu_suite.log | cut -c1-300
grep -E "^(FAILED|ERROR)" $O/gpu_suite.log | head -20
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-400
( time timeout 400 python bench.py ) > $O/bench.log 2>&1; grep -E '^\{' $O/bench.log | tail -1 > $O/bench_n1.json
python - <<'PY'
import json
r = json.load(open("gpurun_out/r06_fifteenth/bench_n1.json"))
print(r["ms_per_step"], r["roofline"]["frac"], r["stats"].get("in_place_cycle_ms"))
h = r["extra"]["halo_pencil_transposes"]
print({k: (v["ms"], v["frac"], v["kernel"][:26]) for k, v in h["per_layout"]["contiguous"].items()})
print({k: (v["ms"], v["frac"]) for k, v in h["config5_pencil_contiguous"]["per_op"].items()})
PY
( time timeout 1200 python tests/test_gpu_runner_cases.py --full ) > $O/reference_sweep_full.log 2>&1; tail -3 $O/reference_sweep_full.log | cut -c1-200; grep -E "all [0-9]+ cases|FAILED" $O/reference_sweep_full.log | head -3
( time timeout 1200 python tests/test_gpu_runner_cases.py --full --ngpu8 ) > $O/reference_sweep_full_ngpu8.log 2>&1; grep -E "all [0-9]+ cases|FAILED" $O/reference_sweep_full_ngpu8.log | head -3
