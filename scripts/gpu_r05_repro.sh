#!/bin/bash
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r05_repro
timeout 1500 python scripts/probe/repro_8rank_r64.py ${1:-all} > gpurun_out/r05_repro/arms.log 2>&1; cat gpurun_out/r05_repro/arms.log
