#!/bin/bash
# Round 6, tenth call: why are the bench's halo-pencil extras slower than the stand-alone probe on the same box?  Probe arms that
# rebuild the bench's history one piece at a time.
cd "$(dirname "$0")/.."
O=gpurun_out/r06_tenth; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
probe() { ( env "$@" timeout 150 python scripts/probe/window_walk_ab.py 2>&1 | grep "^{" ) >> $O/lines_history_ab.jsonl; }
probe X=1
probe WALK_AB_DEFAULT_FIRST=1
probe WALK_AB_PARK_GIB=32
probe WALK_AB_PARK_GIB=32 WALK_AB_DEFAULT_FIRST=1
probe WALK_AB_PREALLOCATE=1 WALK_AB_PARK_GIB=32 WALK_AB_DEFAULT_FIRST=1
probe CUDECOMP_WORKSPACE_POOL_MIB=0 WALK_AB_PARK_GIB=32
python - <<'PY'
import json
for l in open("gpurun_out/r06_tenth/lines_history_ab.jsonl"):
    r = json.loads(l)
    print(r.get("preallocate"), r.get("park_gib"), r["switches"], {k: (v["XToY"], v["YToZ"], v["ZToY"]) for k, v in r["cases"].items()})
PY
( timeout 300 python bench.py --steps 3 --warmup 2 --cpu-sample 0 ) 2>/dev/null | grep -E '^\{' | tail -1 > $O/bench_n1.json
python - <<'PY'
import json
r = json.load(open("gpurun_out/r06_tenth/bench_n1.json"))
h = r["extra"]["halo_pencil_transposes"]
print("bench", {k: (v["ms"], v["frac"]) for k, v in h["per_layout"]["contiguous"].items()})
PY
