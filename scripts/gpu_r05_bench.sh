#!/bin/bash
# round 5: bench line (with the sustained per-hop split), ping-pong cycle in the harness, rocprofv3 evidence
cd "$(dirname "$0")/.."
O=gpurun_out/r05_bench; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 200 scripts/tune/tune_fwd 8 10 9 ) 2>&1 | head -9 > $O/tune_fwd_8_cycle.log; cat $O/tune_fwd_8_cycle.log | cut -c1-220
( timeout 900 python bench.py ) > $O/bench.log 2>&1; grep -E '^\{' $O/bench.log | tail -1 > $O/bench_n1.json; python - <<'PY'
import json
r = json.load(open("gpurun_out/r05_bench/bench_n1.json"))
print(r["ms_per_step"], r["value"], r["config"]["per_op_ms"], r["config"].get("per_op_ms_sustained"))
print({k: v for k, v in r["roofline"].items() if k in ("frac", "avg_launch_ms", "kernel_sum_ms", "sustained_hop_sum_ms", "gap_ms", "kernel")})
PY
bash scripts/gpu_r05_profile.sh > $O/profile.log 2>&1; tail -30 $O/profile.log | cut -c1-220
