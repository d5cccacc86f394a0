#!/bin/bash
# Runs the stand-alone code-size probe (code_size_probe.cpp) over its build matrix: 1 / 2 / 4 processes sharing the GPU, with and
# without an IPC copy, for code objects from 0.06 to 1.9 MB with few large or many small kernels.  Output: one line per process.
cd "$(dirname "$0")/csp" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
for p in probe_16_64 probe_256_32 probe_64_256 probe_96_256 probe_64_400 probe_256_64 probe_64_1024 probe_1024_16 probe_256_256; do
  size=$(python3 -c "
import subprocess
o=subprocess.run(['readelf','-S','-W','$p'],capture_output=True,text=True).stdout
print([int(l.split()[5],16) for l in o.splitlines() if ' .hip_fatbin ' in l][0])")
  echo "== $p: $size bytes of device code"
  for n in 1 2 4; do
    for mode in noipc ipc; do
      [ $n -eq 1 ] && [ $mode = ipc ] && continue
      rm -f /dev/shm/csp_$p$n$mode*
      for r in $(seq 0 $((n - 1))); do RANK=$r WORLD_SIZE=$n JOB=$p$n$mode timeout 120 ./$p $mode & done
      wait
    done
  done
done
