#!/bin/bash
# round 3, hunt for the 8-rank stale slice, step 1: (a) platform probe without any library code, (b) the library's
# stress with the backends SEPARATED (6/7 skip the host rendezvous, 1/2/8 take it)
mkdir -p gpurun_out/hunt1
export HSA_ENABLE_IPC_MODE_LEGACY=0
export CUDECOMP_PEER_TIMEOUT=30
P=scripts/probe/ipc_remap_probe
O=gpurun_out/hunt1
( for args in "8 400 1 4096" "8 400 0 4096" "8 400 9 4096 3" "8 400 5 4096" "8 400 3 4096" "8 300 17 4096" "4 400 1 4096" "8 200 1 65536"; do
    echo "=== $P $args"; timeout 600 $P $args 2>&1 | tail -40
  done ) > $O/remap_probe.log 2>&1
tail -5 $O/remap_probe.log
for b in mix 6 7 2 8; do
  ( time timeout 900 python scripts/probe/stress_eight_ranks.py $b 12 ) > $O/stress_$b.log 2>&1
  tail -3 $O/stress_$b.log
done
