#!/bin/bash
# Round 6, fourth call: (1) the board-address arena against the pooled-suite failure (arms), (2) lines kernel walk: groups x runs,
# (3) the whole suite again (side-by-side native launches) with durations.
cd "$(dirname "$0")/.."
O=gpurun_out/r06_fourth; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
T=$PWD/cudecomp_amd/lib_tuning/libcudecomp.so
for arm in "" "CUDECOMP_BOARD_FRESH_ADDRESS=0" "CUDECOMP_BOARD_FRESH_ADDRESS=0 CUDECOMP_FLAGS_IN_DEVICE_MEMORY=1"; do
  ( timeout 300 python scripts/probe/pool_sequence_stress.py 8 pool $arm 2>/dev/null | grep "^{" | tail -1 ) >> $O/pool_sequence_stress_arms.jsonl
done
cut -c1-700 $O/pool_sequence_stress_arms.jsonl
for arm in "CUDECOMP_LINES_MODE=0" "CUDECOMP_LINES_GROUP=16" "CUDECOMP_LINES_GROUP=0" "CUDECOMP_LINES_GROUP=8" "CUDECOMP_LINES_GROUP=32" "CUDECOMP_LINES_GROUP=16 CUDECOMP_LINES_RUN_KIB=32" "CUDECOMP_LINES_GROUP=16 CUDECOMP_LINES_RUN_KIB=4" "CUDECOMP_LINES_GROUP=16 CUDECOMP_LINES_RUN_KIB=2"; do
  ( env $arm CUDECOMP_AMD_LIBRARY=$T timeout 100 python scripts/probe/window_walk_ab.py 2>&1 | grep "^{" ) >> $O/lines_ab.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r06_fourth/lines_ab.jsonl"):
    r = json.loads(l)
    print(r["switches"], {k: (v["XToY"], v["YToZ"]) for k, v in r["cases"].items()})
PY
( time timeout 1500 python -m pytest tests -q -m gpu --durations=40 --junitxml=$O/junit.xml ) > $O/gpu_suite.log 2>&1; tail -6 $O/gpu_suite.log | cut -c1-300
grep -E "^(FAILED|ERROR)" $O/gpu_suite.log | head -20
