#!/bin/bash
# Round 6, ninth call: the test body's stream dependency fixed -- scenarios, stress, whole suite; then why the bench's halo-pencil
# extras were slow on one box: probe / bench / probe on ONE box.
cd "$(dirname "$0")/.."
O=gpurun_out/r06_ninth; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
run() { timeout 200 python scripts/probe/pool_scenarios.py "$@" 2>/dev/null | grep "^{" | tail -1 >> $O/scenarios.jsonl; }
for i in 1 2 3; do run "8:2x4:b6:cycle 4:2x2:b6 4:2x2:b7 4:2x2:b8 4:1x4:b6 4:4x1:b7 4:2x2:b1"; done
python - <<'PY'
import json
for l in open("gpurun_out/r06_ninth/scenarios.jsonl"):
    r = json.loads(l)
    print(r["scenario"], [(j["job"], j["failures"]) for j in r["results"]])
PY
( timeout 400 python scripts/probe/pool_sequence_stress.py 8 pool 2>/dev/null | grep "^{" | tail -1 ) > $O/pool_sequence_stress.jsonl; cut -c1-400 $O/pool_sequence_stress.jsonl
( time timeout 1500 python -m pytest tests -q -m gpu --durations=40 --junitxml=$O/junit.xml ) > $O/gpu_suite.log 2>&1; tail -6 $O/gpu_suite.log | cut -c1-300
grep -E "^(FAILED|ERROR)" $O/gpu_suite.log | head -20
T=$PWD/cudecomp_amd/lib_tuning/libcudecomp.so
probe() { for arm in "CUDECOMP_LINES_MODE=0" "CUDECOMP_LINES_GROUP=16"; do ( env $arm CUDECOMP_AMD_LIBRARY=$T timeout 100 python scripts/probe/window_walk_ab.py 2>&1 | grep "^{" ) >> $O/lines_ab_$1.jsonl; done; }
probe before
( time timeout 400 python bench.py ) > $O/bench.log 2>&1; grep -E '^\{' $O/bench.log | tail -1 > $O/bench_n1.json
probe after
python - <<'PY'
import json
for w in ("before", "after"):
    for l in open("gpurun_out/r06_ninth/lines_ab_%s.jsonl" % w):
        r = json.loads(l)
        print(w, r["switches"], {k: (v["XToY"], v["YToZ"]) for k, v in r["cases"].items()})
r = json.load(open("gpurun_out/r06_ninth/bench_n1.json"))
print(r["ms_per_step"], r["roofline"]["frac"], r["stats"].get("in_place_cycle_ms"))
h = r["extra"]["halo_pencil_transposes"]
print({k: (v["ms"], v["frac"]) for k, v in h["per_layout"]["contiguous"].items()}, {k: (v["ms"], v["frac"]) for k, v in h["config5_pencil_contiguous"]["per_op"].items()})
PY
