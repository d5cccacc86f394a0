#!/bin/bash
# Round 6, fifth call: bisect the deterministic pooled failure (graph_cycle 1x4 NVSHMEM after 2x2 jobs after an 8-rank world).
cd "$(dirname "$0")/.."
O=gpurun_out/r06_fifth; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
run() { timeout 200 python scripts/probe/pool_scenarios.py "$@" 2>/dev/null | grep "^{" | tail -1 >> $O/scenarios.jsonl; }
run "8:2x4:b6:cycle 4:2x2:b6 4:2x2:b7 4:2x2:b8 4:1x4:b6"
run "4:2x2:b6 4:2x2:b7 4:2x2:b8 4:1x4:b6"
run "8:2x4:b6:cycle 4:1x4:b6"
run "8:2x4:b6:cycle 4:2x2:b6 4:1x4:b6"
run "8:2x4:b6:cycle 4:2x2:b8 4:1x4:b6"
run "2:2x1:b6:cycle 4:2x2:b6 4:2x2:b7 4:2x2:b8 4:1x4:b6"
run "8:2x4:b1:cycle 4:2x2:b6 4:2x2:b7 4:2x2:b8 4:1x4:b6"
run "8:2x4:b6:cycle 4:2x2:b6:cycle 4:2x2:b7:cycle 4:2x2:b8:cycle 4:1x4:b6:cycle"
run "8:2x4:b6:cycle 4:2x2:b6 4:2x2:b7 4:2x2:b8 4:1x4:b6" CUDECOMP_WORKSPACE_POOL_MIB=0
run "8:2x4:b6:cycle 4:2x2:b6 4:2x2:b7 4:2x2:b8 4:1x4:b6" CUDECOMP_VERIFY_IPC_MAPPINGS=1 CUDECOMP_DEBUG_VERIFY_EXCHANGE=1
CUDECOMP_TEST_POOL_KEEP_LOGS=$PWD/$O/logs run "8:2x4:b6:cycle 4:2x2:b6 4:2x2:b7 4:2x2:b8 4:1x4:b6" CUDECOMP_DEBUG_PEER=1 CUDECOMP_VERBOSE=1
python - <<'PY'
import json
for l in open("gpurun_out/r06_fifth/scenarios.jsonl"):
    r = json.loads(l)
    print(r["scenario"], r["env"], [(j["job"], j["failures"]) for j in r["results"]])
    for j in r["results"]:
        if j["failures"]: print("     ", j["first"][:1])
PY
for f in $O/logs/*worker1.log; do tail -c 6000 $f > $f.tail; done; rm -f $O/logs/*.log; ls $O/logs | head
