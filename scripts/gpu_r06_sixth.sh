#!/bin/bash
# Round 6, sixth call: the epoch cells (slab, uncached, atomics) against the pooled failure; whole suite; bench line; the plain
# `bench.py --gpus 4` command and the multi-GPU evidence script as dry runs on the one-GPU box.
cd "$(dirname "$0")/.."
O=gpurun_out/r06_sixth; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
run() { timeout 200 python scripts/probe/pool_scenarios.py "$@" 2>/dev/null | grep "^{" | tail -1 >> $O/scenarios.jsonl; }
for i in 1 2 3 4; do run "8:2x4:b6:cycle 4:2x2:b6 4:2x2:b7 4:2x2:b8 4:1x4:b6"; done
run "8:2x4:b6:cycle 4:2x2:b6 4:1x4:b6"
run "8:2x4:b1:cycle 4:2x2:b6 4:2x2:b7 4:2x2:b8 4:1x4:b6 4:4x1:b6 4:2x2:b6"
python - <<'PY'
import json
for l in open("gpurun_out/r06_sixth/scenarios.jsonl"):
    r = json.loads(l)
    print(r["scenario"], [(j["job"], j["failures"]) for j in r["results"]])
PY
( timeout 400 python scripts/probe/pool_sequence_stress.py 8 pool 2>/dev/null | grep "^{" | tail -1 ) > $O/pool_sequence_stress.jsonl; cut -c1-600 $O/pool_sequence_stress.jsonl
( time timeout 1500 python -m pytest tests -q -m gpu --durations=40 --junitxml=$O/junit.xml ) > $O/gpu_suite.log 2>&1; tail -6 $O/gpu_suite.log | cut -c1-300
grep -E "^(FAILED|ERROR)" $O/gpu_suite.log | head -20
( time timeout 400 python bench.py ) > $O/bench.log 2>&1; grep -E '^\{' $O/bench.log | tail -1 > $O/bench_n1.json
python -c "import json; r = json.load(open('$O/bench_n1.json')); print(r['ms_per_step'], r['roofline']['frac'], r['stats'].get('in_place_cycle_ms')); print(json.dumps(r['extra'].get('halo_pencil_transposes'))[:1800]); print(json.dumps(r['extra'].get('config5_halo'))[:600])"
( time timeout 900 python bench.py --gpus 4 --steps 2 --warmup 1 ) > $O/bench_gpus4_plain_command.log 2>&1; grep -E '^\{' $O/bench_gpus4_plain_command.log | tail -1 > $O/bench_gpus4_plain_command.json; cut -c1-600 $O/bench_gpus4_plain_command.json; tail -3 $O/bench_gpus4_plain_command.log | cut -c1-200
( time timeout 600 bash scripts/first_multi_gpu.sh --shared --quick --gpus 8 ) > $O/first_multi_gpu_dry_run.log 2>&1; tail -25 $O/first_multi_gpu_dry_run.log | cut -c1-200
cp gpurun_out/first_multi_gpu/summary.json $O/first_multi_gpu_dry_run_summary.json 2>/dev/null
