#!/bin/bash
# Round 6, eleventh call: the lines kernel with its loads batched (default AND tuning build, one box), its tests, the bench
# extras, then the whole suite with the thinner default slices.
cd "$(dirname "$0")/.."
O=gpurun_out/r06_eleventh; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
T=$PWD/cudecomp_amd/lib_tuning/libcudecomp.so
probe() { ( env "$@" timeout 150 python scripts/probe/window_walk_ab.py 2>&1 | grep "^{" ) >> $O/lines_ab.jsonl; }
probe X=1
probe CUDECOMP_PRESERVE_OUTPUT_HALOS=1
probe CUDECOMP_AMD_LIBRARY=$T
probe CUDECOMP_AMD_LIBRARY=$T CUDECOMP_LINES_MODE=0
probe CUDECOMP_AMD_LIBRARY=$T CUDECOMP_LINES_RUN_KIB=0
probe CUDECOMP_AMD_LIBRARY=$T CUDECOMP_LINES_RUN_KIB=8
probe CUDECOMP_AMD_LIBRARY=$T CUDECOMP_LINES_RUN_KIB=32
probe WALK_AB_PARK_GIB=32
python - <<'PY'
import json
for l in open("gpurun_out/r06_eleventh/lines_ab.jsonl"):
    r = json.loads(l)
    print(r.get("park_gib"), r["switches"], {k: (v["XToY"], v["YToZ"], v["ZToY"]) for k, v in r["cases"].items()})
PY
( time timeout 300 python -m pytest tests/test_gpu_dense_rows.py tests/test_gpu_kernels.py -q -m gpu ) > $O/lines_tests.log 2>&1; tail -3 $O/lines_tests.log
( time timeout 1500 python -m pytest tests -q -m gpu --durations=30 --junitxml=$O/junit.xml ) > $O/gpu_suite.log 2>&1; tail -6 $O/gpu_suite.log | cut -c1-300
grep -E "^(FAILED|ERROR)" $O/gpu_suite.log | head -20
( time timeout 400 python bench.py ) > $O/bench.log 2>&1; grep -E '^\{' $O/bench.log | tail -1 > $O/bench_n1.json
python - <<'PY'
import json
r = json.load(open("gpurun_out/r06_eleventh/bench_n1.json"))
print(r["ms_per_step"], r["roofline"]["frac"], r["stats"].get("in_place_cycle_ms"))
h = r["extra"]["halo_pencil_transposes"]
print({k: (v["ms"], v["frac"]) for k, v in h["per_layout"]["contiguous"].items()}, {k: (v["ms"], v["frac"]) for k, v in h["config5_pencil_contiguous"]["per_op"].items()})
PY
