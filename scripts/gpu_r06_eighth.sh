#!/bin/bash
# Round 6, eighth call: which feature of graph_cycle poisons the next job / fails in it (after an earlier library world).
cd "$(dirname "$0")/.."
O=gpurun_out/r06_eighth; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
run() { timeout 200 python scripts/probe/pool_scenarios.py "$@" 2>/dev/null | grep "^{" | tail -1 >> $O/scenarios.jsonl; }
W="8:2x4:b6:cycle"
run "$W 4:2x2:b6:graph-nc 4:1x4:b6"
run "$W 4:2x2:b6:graph-td 4:1x4:b6"
run "$W 4:2x2:b6:graph-ds 4:1x4:b6"
run "$W 4:2x2:b6 4:1x4:b6:graph-nc"
run "$W 4:2x2:b6 4:1x4:b6:graph-td"
run "$W 4:2x2:b6 4:1x4:b6:graph-ds"
run "$W 4:2x2:b6:graph-nc 4:1x4:b6:graph-nc"
run "$W 4:2x2:b6:cycle 4:1x4:b6"
run "$W 4:2x2:b6 4:1x4:b6" CUDECOMP_SKIP_LINK_PROBE=1
run "$W 4:2x2:b6 4:1x4:b6" CUDECOMP_PEER_COPY_ENGINE=sdma
python - <<'PY'
import json
for l in open("gpurun_out/r06_eighth/scenarios.jsonl"):
    r = json.loads(l)
    print(r["scenario"], r["env"], [(j["job"], j["failures"]) for j in r["results"]])
PY
# the lines kernel against the window kernel on one box, fresh process vs a process whose allocator has served large blocks before
T=$PWD/cudecomp_amd/lib_tuning/libcudecomp.so
for arm in "CUDECOMP_LINES_MODE=0" "CUDECOMP_LINES_GROUP=16"; do
  ( env $arm CUDECOMP_AMD_LIBRARY=$T timeout 100 python scripts/probe/window_walk_ab.py 2>&1 | grep "^{" ) >> $O/lines_ab.jsonl
  ( env $arm WALK_AB_PREALLOCATE=1 CUDECOMP_AMD_LIBRARY=$T timeout 100 python scripts/probe/window_walk_ab.py 2>&1 | grep "^{" ) >> $O/lines_ab.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r06_eighth/lines_ab.jsonl"):
    r = json.loads(l)
    print(r["switches"], r.get("preallocate"), {k: (v["XToY"], v["YToZ"]) for k, v in r["cases"].items()})
PY
for i in 1 2 3; do timeout 120 python -m pytest tests/test_gpu_c_example.py -q -m gpu -k sub_communicators 2>&1 | tail -2; done
