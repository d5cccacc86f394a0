#!/bin/bash
# How many hardware (KFD) queues do the eight ranks of the stress hold, per setting?  Samples
# /sys/class/kfd/kfd/proc/<pid>/queues while `stress_eight_ranks.py mix 4` runs; prints the maximum per process and in total,
# next to the time per iteration and the failures.  Usage (GPU box): bash scripts/probe/count_queues.sh
export HSA_ENABLE_IPC_MODE_LEGACY=0 CUDECOMP_PEER_TIMEOUT=30
sample() {  # $1 = pid of the stress driver
  local maxtot=0 maxper=0
  while kill -0 $1 2>/dev/null; do
    local tot=0
    for d in /sys/class/kfd/kfd/proc/*/queues; do
      [ -d "$d" ] || continue
      local n=$(ls "$d" 2>/dev/null | wc -l)
      tot=$((tot + n)); [ $n -gt $maxper ] && maxper=$n
    done
    [ $tot -gt $maxtot ] && maxtot=$tot
    sleep 0.2
  done
  echo "   KFD queues: max per process $maxper, max in total $maxtot"
}
# (tests/mp.py gives more than five ranks per device GPU_MAX_HW_QUEUES=2; the runtime's own default is 4)
for setting in "runtime-default-queues:GPU_MAX_HW_QUEUES=4" "runtime-default-queues+copy-engines(stream-per-peer):GPU_MAX_HW_QUEUES=4 CUDECOMP_PEER_COPY_ENGINE=sdma" "runtime-default-queues+device-flags:GPU_MAX_HW_QUEUES=4 CUDECOMP_FLAGS_IN_DEVICE_MEMORY=1" "2-queues+copy-engines:CUDECOMP_PEER_COPY_ENGINE=sdma GPU_MAX_HW_QUEUES=2" "2-queues(harness-default):GPU_MAX_HW_QUEUES=2"; do
  name=${setting%%:*}; envs=${setting#*:}
  echo "== $name ($envs)"
  python scripts/probe/stress_eight_ranks.py mix ${STRESS_ITERS:-6} $envs > /tmp/cq_$name.log 2>&1 &
  pid=$!
  sample $pid
  wait $pid
  grep "iterations failed" /tmp/cq_$name.log | cut -c1-200
  grep -c "FAILED" /tmp/cq_$name.log
done
ls /sys/kernel/debug/kfd 2>/dev/null | head -3
