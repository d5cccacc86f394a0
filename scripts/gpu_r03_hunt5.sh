#!/bin/bash
# round 3, hunt step 5: (a) stand-alone probe, local kernels only, with SPINNING kernels holding the extra queues;
# (b) library stress with DEFAULT queue settings after the library stopped using a stream per peer on shared devices;
# (c) the full -m gpu suite
mkdir -p gpurun_out/hunt5
export HSA_ENABLE_IPC_MODE_LEGACY=0
export CUDECOMP_PEER_TIMEOUT=30
P=scripts/probe/oversub_probe
O=gpurun_out/hunt5
( echo "=== 8 procs, 3 spinning streams, default queues"; timeout 240 $P 8 800 3 2048 1 2>&1 | tail -12; echo "rc $?"
  echo "=== 8 procs, 1 spinning stream, GPU_MAX_HW_QUEUES=2"; GPU_MAX_HW_QUEUES=2 timeout 240 $P 8 800 1 2048 1 2>&1 | tail -12; echo "rc $?"
  echo "=== 4 procs, 3 spinning streams, default queues"; timeout 240 $P 4 800 3 2048 1 2>&1 | tail -12; echo "rc $?" ) > $O/oversub_probe_spin.log 2>&1
grep RESULT $O/oversub_probe_spin.log
( time timeout 900 python scripts/probe/stress_eight_ranks.py mix 60 ) > $O/stress_mix_default_queues.log 2>&1
tail -2 $O/stress_mix_default_queues.log
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 ) > $O/gpu_tests.log 2>&1
tail -25 $O/gpu_tests.log
