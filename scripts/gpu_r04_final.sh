#!/bin/bash
# Round 4, last GPU session: the bench line and the rocprofv3 stats + PMC passes at HEAD, then the 8-rank reference matrix.
cd "$(dirname "$0")/.."
O=gpurun_out/r04_final
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(make -s -j16 -C cudecomp_amd && make -s -j8 -C tests/native all && make -s -C benchmark && make -s -C oracle cpu_mpi_cycle && make -s -C tests/shim) > $O/build.log 2>&1 || { echo "build failed"; tail -20 $O/build.log; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-200
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
( time bash scripts/gpu_profile.sh ) > $O/profile.log 2>&1; tail -2 $O/profile.log
( time bash scripts/gpu_profile_dtypes.sh ) > $O/profile_dtypes.log 2>&1; tail -2 $O/profile_dtypes.log
( time timeout 900 python tests/test_gpu_runner_cases.py --full --ngpu8 ) > $O/reference_sweep_full_ngpu8.log 2>&1; tail -5 $O/reference_sweep_full_ngpu8.log | cut -c1-200
