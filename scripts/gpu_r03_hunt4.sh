#!/bin/bash
# round 3, hunt step 4: is it hardware-queue oversubscription?  (a) stand-alone probe with local kernels only, 8 processes,
# default queues vs GPU_MAX_HW_QUEUES=2; (b) the library stress with 2 queues per process
mkdir -p gpurun_out/hunt4
export HSA_ENABLE_IPC_MODE_LEGACY=0
export CUDECOMP_PEER_TIMEOUT=30
P=scripts/probe/oversub_probe
O=gpurun_out/hunt4
( echo "=== default queues";          timeout 300 $P 8 1500 4 2048 2>&1 | tail -12
  echo "=== default queues, 7 streams"; timeout 300 $P 8 1500 7 2048 2>&1 | tail -12
  echo "=== GPU_MAX_HW_QUEUES=2";     GPU_MAX_HW_QUEUES=2 timeout 300 $P 8 1500 4 2048 2>&1 | tail -12
  echo "=== 4 processes, default";    timeout 300 $P 4 1500 4 2048 2>&1 | tail -12 ) > $O/oversub_probe.log 2>&1
grep RESULT $O/oversub_probe.log
( time timeout 900 python scripts/probe/stress_eight_ranks.py mix 60 GPU_MAX_HW_QUEUES=2 ) > $O/stress_mix_2queues.log 2>&1
tail -2 $O/stress_mix_2queues.log
