#!/bin/bash
# Round 4, fourth GPU session: regime map with the (cheap, pid-namespace-safe) queue census, 200 stress iterations at HEAD,
# the two-hop relay tests, the changed sweeps, rocprofv3 stats + PMC of the bench kernels and of the fp32 / complex128 kernels.
cd "$(dirname "$0")/.."
O=gpurun_out/r04_batch4
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(make -s -j16 -C cudecomp_amd && make -s -j8 -C tests/native all) > $O/build.log 2>&1 || { echo "build failed"; tail -20 $O/build.log; exit 1; }
( time timeout 900 python -m pytest tests/test_gpu_relay.py -x -q -m gpu --durations=5 ) > $O/a_relay.log 2>&1; tail -4 $O/a_relay.log
( time timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_workspace_pool.py -x -q -m gpu ) > $O/b_kernels_pool.log 2>&1; tail -3 $O/b_kernels_pool.log
timeout 700 python scripts/probe/hunt_shared_gpu.py $O/regime 1 regime > $O/c_regime.jsonl 2> $O/c_regime.err
timeout 1200 python scripts/probe/hunt_shared_gpu.py $O/hunt 150 stress_200 > $O/d_stress.jsonl 2> $O/d_stress.err
rm -f $O/regime/*_rank[1-9]*.log $O/regime/*_cases.txt $O/hunt/*_cases.txt
python - <<'PY'
import json
for f in ("c_regime.jsonl", "d_stress.jsonl"):
    for line in open("gpurun_out/r04_batch4/" + f):
        try:
            r = json.loads(line)
        except ValueError:
            continue
        if "arm" in r:
            print("%-40s cases %5d failed %d timeout %s ms/case %-7s compute queues %s of %s slots" % (r["arm"], r["cases"], r["failed"], r["timed_out"],
                  r["ms_per_case"], r.get("compute_queues_on_this_gpu_max"), r.get("hardware_queue_slots")))
            for d in r["diag"][:6]:
                print("    ", d[:400])
PY
( time timeout 1500 python -m pytest tests/test_gpu_native_sweep.py -x -q -m gpu --durations=5 ) > $O/e_native_sweep.log 2>&1; tail -4 $O/e_native_sweep.log
( time bash scripts/gpu_profile.sh ) > $O/f_profile.log 2>&1; tail -3 $O/f_profile.log
( time bash scripts/gpu_profile_dtypes.sh ) > $O/g_profile_dtypes.log 2>&1; tail -12 $O/g_profile_dtypes.log | cut -c1-300
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/h_bench.json 2> $O/h_bench.err; tail -c 400 $O/h_bench.json
