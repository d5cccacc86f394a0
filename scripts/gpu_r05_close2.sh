#!/bin/bash
# round 5, closing run after the dense row copy: bench line at HEAD (+ rocprofv3 kernel stats of the same command), smoke() as the
# driver runs it, then the -m gpu suite minus the three modules already run at this HEAD's kernels (native sweep, transposes,
# runner cases: scripts/gpu_r05_dense2.sh, 160 passed)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_close2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
REPO=$PWD
( time timeout 400 python bench.py ) > $O/bench.log 2>&1; grep -E '^\{' $O/bench.log | tail -1 > $O/bench_n1.json; python - <<'PY'
import json
r = json.load(open("gpurun_out/r05_close2/bench_n1.json"))
print(r["ms_per_step"], r["value"], r["config"]["per_op_ms"], r["config"].get("per_op_ms_sustained"))
print({k: v for k, v in r["roofline"].items() if k in ("frac", "avg_launch_ms", "kernel_sum_ms", "gap_ms", "kernel")})
print(json.dumps(r["extra"].get("halo_pencil_transposes"))[:1500])
PY
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace -o bench -- \
     python $REPO/bench.py --steps 5 --warmup 3 --cpu-sample 0 --no-extras > $REPO/$O/trace.log 2>&1 )
for f in $(find $O/trace -name "*kernel_stats.csv"); do echo "== $f"; head -5 $f | cut -c1-200; done
find $O/trace -name "*kernel_trace.csv" -size +1M -delete
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log | cut -c1-200
( time timeout 560 python -m pytest tests -x -q -m gpu --ignore=tests/test_gpu_native_sweep.py --ignore=tests/test_gpu_transpose.py --ignore=tests/test_gpu_runner_cases.py --durations=5 ) > $O/gpu_suite_rest.log 2>&1; tail -14 $O/gpu_suite_rest.log | cut -c1-250
