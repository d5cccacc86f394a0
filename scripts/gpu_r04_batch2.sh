#!/bin/bash
# Round 4, second GPU session: the new tests (every-cell halos incl. config 5, MPI flavour on the reference matrix, FFT
# shortcuts + R2C, fork isolation of in-process GPU tests), the bench line with its new records, and the dry run of
# scripts/first_multi_gpu.sh on 8 ranks sharing the GPU.
cd "$(dirname "$0")/.."
O=gpurun_out/r04_batch2
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(make -s -j16 -C cudecomp_amd && make -s -j16 -C cudecomp_amd MPI=1 && make -s -j8 -C tests/native all mpi && make -s -C benchmark && make -s -C oracle cpu_mpi_cycle) > $O/build.log 2>&1 || { echo "build failed"; tail -20 $O/build.log; exit 1; }
( time timeout 900 python -m pytest tests/test_inprocess_isolation.py tests/test_gpu_halo.py tests/test_gpu_workspace_pool.py -x -q -m "gpu or not gpu" ) > $O/a_halo_isolation.log 2>&1; tail -3 $O/a_halo_isolation.log
( time timeout 1200 python -m pytest tests/test_gpu_baseline_configs.py -x -q -m gpu -k config5 ) > $O/b_config5.log 2>&1; tail -3 $O/b_config5.log
( time timeout 1500 python -m pytest tests/test_gpu_mpi_flavour.py -x -q -m gpu --durations=5 ) > $O/c_mpi_flavour.log 2>&1; tail -4 $O/c_mpi_flavour.log
( time timeout 1500 python -m pytest tests/test_gpu_fft3d.py -x -q -m gpu --durations=5 ) > $O/d_fft.log 2>&1; tail -4 $O/d_fft.log
( time timeout 900 python bench.py ) > $O/e_bench.json 2> $O/e_bench.err; tail -c 600 $O/e_bench.json
( time timeout 2400 bash scripts/first_multi_gpu.sh --shared --quick --gpus 8 ) > $O/f_first_multi_gpu_dry_run.log 2>&1; tail -30 $O/f_first_multi_gpu_dry_run.log
# regime map once more, with the driver's queue census per arm (compute vs SDMA queues over all processes)
timeout 900 python scripts/probe/hunt_shared_gpu.py $O/regime 1 regime > $O/g_regime.jsonl 2> $O/g_regime.err; rm -f $O/regime/*_rank[1-9]*.log $O/regime/*_cases.txt
python - <<'PY'
import json
for line in open("gpurun_out/r04_batch2/g_regime.jsonl"):
    try:
        r = json.loads(line)
    except ValueError:
        continue
    if "arm" in r:
        print("%-40s ms/case %-7s failed %d queues %s" % (r["arm"], r["ms_per_case"], r["failed"], r.get("kfd_queues_max")))
PY
