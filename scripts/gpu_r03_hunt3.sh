#!/bin/bash
# round 3, hunt step 3: (a) probe variants: importers never close / owner naps between free and malloc / 2 processes with
# handle dump; (b) the library with the workspace pool + verified mappings (default), with the pool off (verification
# must catch the stale mappings), and the new tests
mkdir -p gpurun_out/hunt3
export HSA_ENABLE_IPC_MODE_LEGACY=0
export CUDECOMP_PEER_TIMEOUT=30
P=scripts/probe/ipc_remap_probe
O=gpurun_out/hunt3
( for args in "8 100 257 4096" "8 100 513 4096" "2 12 129 4096" "2 100 257 4096"; do
    echo "=== $P $args"; timeout 100 $P $args 2>&1 | head -40; echo "--- rc ${PIPESTATUS[0]}"
  done ) > $O/remap_probe.log 2>&1
grep -c "bad u64 in bytes" $O/remap_probe.log; grep RESULT $O/remap_probe.log
( time timeout 600 python -m pytest tests/test_gpu_workspace_pool.py tests/test_gpu_transpose.py -x -q -m gpu -k "pool or two_live or bench_workload or 2_31" ) > $O/new_tests.log 2>&1
tail -5 $O/new_tests.log
( time timeout 700 python scripts/probe/stress_eight_ranks.py mix 30 ) > $O/stress_mix_pool.log 2>&1
tail -2 $O/stress_mix_pool.log
( time timeout 500 python scripts/probe/stress_eight_ranks.py mix 20 CUDECOMP_WORKSPACE_POOL_MIB=0 ) > $O/stress_mix_nopool_verify.log 2>&1
tail -2 $O/stress_mix_nopool_verify.log
