#!/bin/bash
# Round 4: three consecutive full -m gpu runs at HEAD, library defaults, exactly the driver's command line.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04_suite
export HSA_ENABLE_IPC_MODE_LEGACY=0
(make -s -j16 -C cudecomp_amd && make -s -j16 -C cudecomp_amd MPI=1 && make -s -j8 -C tests/native all mpi && make -s -C benchmark && make -s -C oracle cpu_mpi_cycle && make -s -C tests/shim) > gpurun_out/r04_suite/build_3x.log 2>&1 || { echo "build failed"; tail -20 gpurun_out/r04_suite/build_3x.log; exit 1; }
for n in 3 4 5; do
  ( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 ) > gpurun_out/r04_suite/suite_$n.log 2>&1
  echo "== run $n: $(grep -E ' passed| failed| error' gpurun_out/r04_suite/suite_$n.log | tail -1)"
done
