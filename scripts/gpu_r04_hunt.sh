#!/bin/bash
# Round 4, item 1 of VERDICT: regime map + shuffled / repeated hunt with diagnostics (scripts/probe/hunt_shared_gpu.py).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04_hunt
export HSA_ENABLE_IPC_MODE_LEGACY=0
# the snapshot may have been taken between an edit and a rebuild: make sure binaries match the sources (no-op otherwise)
(make -s -j16 -C cudecomp_amd && make -s -j8 -C tests/native) > gpurun_out/r04_hunt/build.log 2>&1 || { echo "build failed"; tail -20 gpurun_out/r04_hunt/build.log; exit 1; }
python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/r04_hunt/device.txt 2>&1
timeout 2300 python scripts/probe/hunt_shared_gpu.py gpurun_out/r04_hunt ${1:-150} > gpurun_out/r04_hunt/summary.jsonl 2> gpurun_out/r04_hunt/stderr.log
echo "rc $?" >> gpurun_out/r04_hunt/stderr.log
# keep the merge small: per-rank logs of clean arms are not interesting
python - <<'PY'
import json, os, glob
d = "gpurun_out/r04_hunt"
for line in open(os.path.join(d, "summary.jsonl")):
    try:
        rec = json.loads(line)
    except ValueError:
        continue
    if "arm" in rec and not rec["failed"] and not rec["timed_out"] and not rec["diag"]:
        for f in glob.glob(os.path.join(d, rec["arm"] + "_rank[1-9]*.log")) + glob.glob(os.path.join(d, rec["arm"] + "_cases.txt")):
            os.unlink(f)
PY
tail -c 3000 gpurun_out/r04_hunt/summary.jsonl
