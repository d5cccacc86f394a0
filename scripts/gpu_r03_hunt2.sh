#!/bin/bash
# round 3, hunt step 2: the IPC remap probe with variants (short, bounded), then the library stress with the exchange
# verification on
mkdir -p gpurun_out/hunt2
export HSA_ENABLE_IPC_MODE_LEGACY=0
export CUDECOMP_PEER_TIMEOUT=30
P=scripts/probe/ipc_remap_probe
O=gpurun_out/hunt2
( for args in "8 40 129 4096" "8 150 33 4096" "8 150 65 4096" "4 150 1 4096" "2 150 1 4096" "8 150 1 65536" "8 150 0 4096"; do
    echo "=== $P $args"; timeout 120 $P $args 2>&1 | head -60; echo "--- rc ${PIPESTATUS[0]}"
  done ) > $O/remap_probe.log 2>&1
grep -c "bad u64 in bytes" $O/remap_probe.log; grep RESULT $O/remap_probe.log
( time timeout 900 python scripts/probe/stress_eight_ranks.py mix 24 CUDECOMP_DEBUG_VERIFY_EXCHANGE=1 ) > $O/stress_mix_verify.log 2>&1
tail -3 $O/stress_mix_verify.log
grep -h "CUDECOMP:VERIFY" gpurun_out/native_failure_* 2>/dev/null | head -20
