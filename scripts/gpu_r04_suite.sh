#!/bin/bash
# Round 4: the full -m gpu suite exactly as the driver runs it (library defaults, no environment tweaks), run number $1;
# with "extras" as $2 also the dtype PMC passes and the tiny-transpose latency table.
cd "$(dirname "$0")/.."
N=${1:-1}
mkdir -p gpurun_out/r04_suite
export HSA_ENABLE_IPC_MODE_LEGACY=0
(make -s -j16 -C cudecomp_amd && make -s -j16 -C cudecomp_amd MPI=1 && make -s -j8 -C tests/native all mpi && make -s -C benchmark && make -s -C oracle cpu_mpi_cycle && make -s -C tests/shim) > gpurun_out/r04_suite/build_$N.log 2>&1 || { echo "build failed"; tail -20 gpurun_out/r04_suite/build_$N.log; exit 1; }
( time timeout 2700 python -m pytest tests -x -q -m gpu --durations=12 ) > gpurun_out/r04_suite/suite_$N.log 2>&1
tail -22 gpurun_out/r04_suite/suite_$N.log | cut -c1-300
if [ "$2" = "extras" ]; then
  ( time bash scripts/gpu_profile_dtypes.sh ) > gpurun_out/r04_suite/profile_dtypes.log 2>&1; tail -8 gpurun_out/r04_suite/profile_dtypes.log | cut -c1-260
  timeout 600 python scripts/probe/flag_latency.py > gpurun_out/r04_suite/flag_latency.json 2> gpurun_out/r04_suite/flag_latency.err; cat gpurun_out/r04_suite/flag_latency.json
fi
