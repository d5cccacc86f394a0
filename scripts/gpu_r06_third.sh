#!/bin/bash
# Round 6, third call: (1) is the one wrong eager cycle of the pooled suite a handle-reuse bug?  (2) lines kernel walks A/B,
# (3) the in-place rotation kernel: tests + bench line, (4) smoke().
cd "$(dirname "$0")/.."
O=gpurun_out/r06_third; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
T=$PWD/cudecomp_amd/lib_tuning/libcudecomp.so
( timeout 600 python scripts/probe/pool_sequence_stress.py 8 ) > $O/pool_sequence_stress.log 2>&1; grep "^{" $O/pool_sequence_stress.log | tail -1 | cut -c1-1500
( time timeout 300 python -m pytest tests/test_gpu_transpose.py tests/test_gpu_dense_rows.py -q -m gpu -k "rotation or in_place or lines or preserve" ) > $O/rotate_tests.log 2>&1; tail -5 $O/rotate_tests.log | cut -c1-300
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-400
for arm in "CUDECOMP_LINES_MODE=0" "CUDECOMP_LINES_WALK=0" "CUDECOMP_LINES_WALK=1" "CUDECOMP_LINES_WALK=2" "CUDECOMP_LINES_RUN_KIB=8" "CUDECOMP_LINES_RUN_KIB=16" "CUDECOMP_LINES_RUN_KIB=32" "CUDECOMP_LINES_WALK=1 CUDECOMP_LINES_UNIT=64"; do
  ( env $arm CUDECOMP_AMD_LIBRARY=$T timeout 100 python scripts/probe/window_walk_ab.py 2>&1 | grep "^{" ) >> $O/lines_ab.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r06_third/lines_ab.jsonl"):
    r = json.loads(l)
    print(r["switches"], {k: (v["XToY"], v["YToZ"]) for k, v in r["cases"].items()})
PY
( time timeout 300 python bench.py --steps 5 --warmup 3 --cpu-sample 0 --no-extras ) > $O/bench.log 2>&1; grep -E '^\{' $O/bench.log | tail -1 > $O/bench_n1.json
python -c "import json; r = json.load(open('$O/bench_n1.json')); print(r['ms_per_step'], r['roofline']['frac'], r['stats'].get('in_place_cycle_ms'))"
( CUDECOMP_DISABLE_INPLACE_ROTATION=1 timeout 300 python bench.py --steps 5 --warmup 3 --cpu-sample 0 --no-extras ) 2>/dev/null | grep -E '^\{' | tail -1 | python -c "import json,sys; r = json.loads(sys.stdin.read()); print('staged in place:', r['stats'].get('in_place_cycle_ms'))"
