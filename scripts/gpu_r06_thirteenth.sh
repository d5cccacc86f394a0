#!/bin/bash
# Round 6, thirteenth call: transpose_rowlines_kernel (inverse hops onto halo pencils): tests, A/B against the window kernel on one
# box, bench line, whole suite.
cd "$(dirname "$0")/.."
O=gpurun_out/r06_thirteenth; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
T=$PWD/cudecomp_amd/lib_tuning/libcudecomp.so
( time timeout 400 python -m pytest tests/test_gpu_dense_rows.py tests/test_gpu_kernels.py -q -m gpu ) > $O/kernel_tests.log 2>&1; tail -5 $O/kernel_tests.log | cut -c1-300
grep -E "^(FAILED|ERROR)|^E  " $O/kernel_tests.log | head -12 | cut -c1-300
probe() { ( env "$@" timeout 150 python scripts/probe/window_walk_ab.py 2>&1 | grep "^{" ) >> $O/lines_ab.jsonl; }
probe X=1
probe CUDECOMP_PRESERVE_OUTPUT_HALOS=1
probe X=2
probe CUDECOMP_AMD_LIBRARY=$T CUDECOMP_LINES_MODE=0
python - <<'PY'
import json
for l in open("gpurun_out/r06_thirteenth/lines_ab.jsonl"):
    r = json.loads(l)
    print(r["switches"], {k: (v["XToY"], v["YToZ"], v["ZToY"], v["YToX"], v["kernel"][:28]) for k, v in r["cases"].items()})
PY
( time timeout 1500 python -m pytest tests -q -m gpu --durations=25 --junitxml=$O/junit.xml ) > $O/gpu_suite.log 2>&1; tail -6 $O/gpu_suite.log | cut -c1-300
grep -E "^(FAILED|ERROR)" $O/gpu_suite.log | head -20
( time timeout 400 python bench.py ) > $O/bench.log 2>&1; grep -E '^\{' $O/bench.log | tail -1 > $O/bench_n1.json
python - <<'PY'
import json
r = json.load(open("gpurun_out/r06_thirteenth/bench_n1.json"))
print(r["ms_per_step"], r["roofline"]["frac"], r["stats"].get("in_place_cycle_ms"))
h = r["extra"]["halo_pencil_transposes"]
print({k: (v["ms"], v["frac"], v["kernel"][:26]) for k, v in h["per_layout"]["contiguous"].items()})
print({k: (v["ms"], v["frac"]) for k, v in h["config5_pencil_contiguous"]["per_op"].items()})
PY
