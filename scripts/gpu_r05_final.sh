#!/bin/bash
# round 5, closing run: the full -m gpu suite as the driver runs it, the reference's 8-rank matrix with the twins' kept data buffers,
# the bench line + rocprofv3 evidence, local phases at the multi-GPU shapes, one more harness table
cd "$(dirname "$0")/.."
O=gpurun_out/r05_final; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 ) > $O/gpu_suite.log 2>&1; tail -14 $O/gpu_suite.log | cut -c1-200
( time timeout 1200 python tests/test_gpu_runner_cases.py --full --ngpu8 ) > $O/reference_sweep_full_ngpu8.log 2>&1; tail -4 $O/reference_sweep_full_ngpu8.log | cut -c1-200
( timeout 900 python bench.py ) > $O/bench.log 2>&1; grep -E '^\{' $O/bench.log | tail -1 > $O/bench_n1.json
python - <<'PY'
import json
r = json.load(open("gpurun_out/r05_final/bench_n1.json"))
print(r["ms_per_step"], r["value"], r["config"]["per_op_ms"], r["config"].get("per_op_ms_sustained"))
print({k: v for k, v in r["roofline"].items() if k in ("frac", "avg_launch_ms", "kernel_sum_ms", "sustained_hop_sum_ms", "gap_ms", "kernel", "per_kernel")})
PY
bash scripts/gpu_r05_profile.sh > $O/profile.log 2>&1; tail -12 $O/profile.log | cut -c1-200
( timeout 300 python scripts/probe/local_phases.py ) > $O/local_phases.json 2> $O/local_phases.err; tail -3 $O/local_phases.err | cut -c1-200
( timeout 200 scripts/tune/tune_fwd 8 10 4 ) > $O/tune_fwd_8_phase4.log 2>&1; tail -10 $O/tune_fwd_8_phase4.log
