#!/bin/bash
# Round 4, third GPU session: regime map with the per-GPU queue census, hunt arms with the copy engines (a stream per
# peer, the round-2 shape of the transport), 200 stress iterations at HEAD, the device-memory flags with at most four ranks
# per GPU (full 4-rank reference matrix, latency, config-5 halos), the fp32 tile-shape A/B, the kernel tests.
cd "$(dirname "$0")/.."
O=gpurun_out/r04_batch3
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(make -s -j16 -C cudecomp_amd && make -s -j8 -C tests/native all) > $O/build.log 2>&1 || { echo "build failed"; tail -20 $O/build.log; exit 1; }
for n in /sys/class/kfd/kfd/topology/nodes/*; do echo "== $n"; cat $n/gpu_id 2>/dev/null; grep -E "simd_count|num_cp_queues|num_sdma|max_waves|num_xcc|array_count" $n/properties 2>/dev/null; done > $O/topology.txt 2>&1
( time timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu ) > $O/a_kernels.log 2>&1; tail -3 $O/a_kernels.log
timeout 900 python scripts/probe/hunt_shared_gpu.py $O/regime 1 regime > $O/b_regime.jsonl 2> $O/b_regime.err
timeout 1500 python scripts/probe/hunt_shared_gpu.py $O/hunt 150 hunt_sdma stress_200 > $O/c_hunt.jsonl 2> $O/c_hunt.err
rm -f $O/regime/*_rank[1-9]*.log $O/regime/*_cases.txt
python - <<'PY'
import json
for f in ("b_regime.jsonl", "c_hunt.jsonl"):
    for line in open("gpurun_out/r04_batch3/" + f):
        try:
            r = json.loads(line)
        except ValueError:
            continue
        if "arm" in r:
            print("%-40s cases %5d failed %d ms/case %-7s compute queues %s of %s slots" % (r["arm"], r["cases"], r["failed"], r["ms_per_case"],
                  r.get("compute_queues_on_this_gpu_max"), r.get("hardware_queue_slots")))
            for d in r["diag"][:6]:
                print("    ", d[:400])
PY
# fp32 tile shapes
for shape in 0 1 2; do echo "== CUDECOMP_TILE_SHAPE=$shape"; CUDECOMP_TILE_SHAPE=$shape timeout 300 python scripts/probe/dtype_table.py fp32 2>/dev/null; done > $O/d_fp32_tile_shapes.log 2>&1; cat $O/d_fp32_tile_shapes.log | cut -c1-700
# device-memory flags, at most four ranks per GPU
( time CUDECOMP_FLAGS_IN_DEVICE_MEMORY=1 timeout 1500 python tests/test_gpu_runner_cases.py --full ) > $O/e_flags_device_reference_matrix_4ranks.log 2>&1; tail -4 $O/e_flags_device_reference_matrix_4ranks.log
for flags in 0 1; do
  CUDECOMP_FLAGS_IN_DEVICE_MEMORY=$flags HALO_BENCH_GRID=2x2 timeout 600 python scripts/probe/halo_bench_ranks.py > $O/f_halo_bench_4ranks_flags$flags.json 2> $O/f_halo_bench_4ranks_flags$flags.err
done
python - <<'PY'
import json
for flags in (0, 1):
    try:
        d = json.load(open("gpurun_out/r04_batch3/f_halo_bench_4ranks_flags%d.json" % flags))
        v = d["variants"]["nvshmem_overlapped"]
        print("flags", d["flags"], {ax: {k: v[ax][k]["ms"] for k in ("dim0", "dim1", "dim2")} for ax in "XYZ"})
    except Exception as e:
        print("halo bench flags", flags, "failed:", e)
PY
