#!/bin/bash
# Round 6, seventh call: the pooled failure once more, instrumented (the row's counters at the start of every call), and the
# question "idle processes or world rebuild": eight workers alive WITHOUT an earlier library world.
cd "$(dirname "$0")/.."
O=gpurun_out/r06_seventh; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
run() { timeout 200 python scripts/probe/pool_scenarios.py "$@" 2>/dev/null | grep "^{" | tail -1 >> $O/scenarios.jsonl; }
run "8:2x4:b6:probe 4:2x2:b6 4:1x4:b6"
run "8:2x4:b6:cycle 4:2x2:b6 4:1x4:b6"
run "8:2x4:b6:cycle 4:2x2:b6 4:1x4:b7"
run "8:2x4:b6:cycle 4:2x2:b6 4:1x4:b1"
run "8:2x4:b6:cycle 4:2x2:b6 4:1x4:b6:cycle"
run "8:2x4:b6:cycle 4:2x2:b7 4:1x4:b6"
run "8:2x4:b6:cycle 4:1x4:b6 4:2x2:b6 4:1x4:b6"
CUDECOMP_TEST_POOL_KEEP_LOGS=$PWD/$O/logs run "8:2x4:b6:cycle 4:2x2:b6 4:1x4:b6" CUDECOMP_DEBUG_PEER=1
python - <<'PY'
import json
for l in open("gpurun_out/r06_seventh/scenarios.jsonl"):
    r = json.loads(l)
    print(r["scenario"], r["env"], [(j["job"], j["failures"]) for j in r["results"]])
    for j in r["results"]:
        if j["failures"]: print("     ", j["first"][:1])
PY
for f in $O/logs/*worker[0-3].log; do grep -E "=== job|DEBUG rank . begin" $f | tail -40 > $f.txt; done; rm -f $O/logs/*.log; ls $O/logs
