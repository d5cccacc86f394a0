"""BASELINE config 5 with neighbours: cudecompUpdateHalos{X,Y,Z} at 2048 x 2048 x 1024 fp64 on a 2 x 4 grid, halo width 2,
periodic, eight ranks (sharing the visible GPUs), per pencil and dim: time per update and the library's pack / exchange /
unpack split, for the overlapped and the plain sequence of the one-sided transport (reference include/internal/halo.h:
200-260 is the plain one).  Ranks that share ONE GPU make the exchange a local copy: the numbers are a flow and kernel
measurement, not a link measurement.
    python scripts/probe/halo_bench_ranks.py > gpurun_out/halo_bench_ranks.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cudecomp_amd as cd  # noqa: E402
from tests.mp import run_ranks  # noqa: E402

args = {"gdims": (2048, 2048, 1024), "pdims": (2, 4), "kind": 1, "halo": (2, 2, 2), "periods": (1, 1, 1), "axes": [0, 1, 2]}
nranks = 8
if os.environ.get("HALO_BENCH_GRID") == "2x2":  # the same per-rank pencils on four ranks (at most four ranks per GPU)
    args["gdims"], args["pdims"], nranks = (2048, 2048, 512), (2, 2), 4
out = {"workload": "%dx%dx%d fp64, %dx%d grid, halo 2, periodic; %d ranks" % (tuple(args["gdims"]) + tuple(args["pdims"]) + (nranks,)),
       "flags": "device memory" if os.environ.get("CUDECOMP_FLAGS_IN_DEVICE_MEMORY") == "1" else "host-pinned board", "variants": {}}
for name, backend, env in (("nvshmem_overlapped", cd.HALO_COMM_NVSHMEM, {}),
                           ("nvshmem_plain", cd.HALO_COMM_NVSHMEM, {"CUDECOMP_DISABLE_HALO_OVERLAP": "1"}),
                           ("mpi_plain", cd.HALO_COMM_MPI, {"CUDECOMP_DISABLE_HALO_OVERLAP": "1"})):
    env = dict(env, CUDECOMP_ENABLE_PERFORMANCE_REPORT="1", CUDECOMP_PERFORMANCE_REPORT_WARMUP_SAMPLES="3",
               CUDECOMP_PERFORMANCE_REPORT_SAMPLES="10")
    res = run_ranks(nranks, "tests.gpu_bodies", "halo_timed", dict(args, halo_backend=backend), timeout=900, extra_env=env)
    # max over ranks per entry (the slowest rank is what an application sees)
    merged = {}
    for ax in "XYZ":
        merged[ax] = {"pencil_shape_rank0": res[0][ax]["pencil_shape"], "workspace_MiB": res[0][ax]["workspace_MiB"]}
        for dim in range(3):
            k = "dim%d" % dim
            merged[ax][k] = {f: max(r[ax][k][f] for r in res) for f in ("ms", "pack_ms", "exchange_ms", "unpack_ms", "wire_MiB")}
    out["variants"][name] = merged
print(json.dumps(out, indent=1))
