#!/bin/bash
# round 5, first GPU call: partition probe, self-check of the input gate, per-case cost of the A/B state
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r05_first
bash scripts/probe/partition_probe.sh > gpurun_out/r05_first/partition_probe.out 2>&1
tail -5 gpurun_out/partition_probe/probe.log
# gate self-check: an injected stale input must trip the gate (2 ranks, one transpose case, one halo case)
for prog in transpose halo; do
  echo "--pr 1 --pc 2 --gx 64 --gy 60 --gz 68 --backend 1" > /tmp/one_case.txt
  for r in 0 1; do
    RANK=$r WORLD_SIZE=2 LOCAL_RANK=$r MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 CUDECOMP_BOOTSTRAP_PORT=29612 CUDECOMP_TEST_JOB=gate$prog \
      CUDECOMP_TEST_INJECT_STALE_INPUT=1 timeout 120 tests/native/build/${prog}_test_R64 --testfile /tmp/one_case.txt > gpurun_out/r05_first/gate_selfcheck_${prog}_rank$r.log 2>&1 &
  done
  wait
  grep -h -E "DIAG|Input gate|FAILED|PASSED" gpurun_out/r05_first/gate_selfcheck_${prog}_rank*.log | cut -c1-300
done
# per-case cost of the A/B in both library states (20 s per program)
timeout 400 python scripts/probe/upload_ab.py gpurun_out/r05_first/ab_tuning 80 --arms default --slices 1 > gpurun_out/r05_first/ab_cost_tuning_lib.jsonl 2>&1
timeout 400 python scripts/probe/upload_ab.py gpurun_out/r05_first/ab_default 80 --arms default --slices 1 --lib $PWD/cudecomp_amd/lib > gpurun_out/r05_first/ab_cost_default_lib.jsonl 2>&1
python - <<'PY'
import json
for f in ("tuning", "default"):
    for line in open("gpurun_out/r05_first/ab_cost_%s_lib.jsonl" % f):
        try:
            r = json.loads(line)
        except ValueError:
            print(line.strip()[:300]); continue
        if "arm" in r:
            print(f, r["arm"], r["completed"], "cases", r["ms_per_case"], "ms/case failed", r["failed"], "trips", r["gate_trips_input_stale"], r["kfd_queues_max"], r["gate_reports"][:1])
        else:
            print(f, json.dumps(r)[:300])
PY
