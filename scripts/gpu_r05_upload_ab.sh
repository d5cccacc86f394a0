#!/bin/bash
# round 5, VERDICT item 1: the three-arm upload A/B with the input-integrity gate.  State: ONE code object of 0.73 MB in the
# library (scripts/probe/ab_state: the round-5 bisect's "same_tu" build, the state in which round 4's two failures were seen),
# 8 ranks alone on the device.  17 minutes per arm (half halo mix, half transpose mix) or 150,000 cases per program; arms
# interleaved in two slices.  Stop rule and reading: docstring of scripts/probe/upload_ab.py.
# (First attempt of the round: 22 minutes per arm ran into gpurun's 60-minute limit and its 64 MiB return limit -- nothing came
# back.  This run keeps the bulky per-rank logs in /tmp and fits into 55 minutes.)
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r05_upload_ab
timeout 3400 python scripts/probe/upload_ab.py gpurun_out/r05_upload_ab ${1:-1020} --lib $PWD/scripts/probe/ab_state --slices 2 > gpurun_out/r05_upload_ab/results.jsonl 2> gpurun_out/r05_upload_ab/stderr.log
tail -2 gpurun_out/r05_upload_ab/results.jsonl | cut -c1-1500; du -sh gpurun_out
