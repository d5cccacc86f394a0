"""Flag round trips of the one-sided exchanges, device-memory flags vs the host-pinned board: tiny transposes (16^3 fp64)
back to back on 2, 4 and 8 ranks sharing the visible GPUs; microseconds per transpose (max over ranks).
    python scripts/probe/flag_latency.py > gpurun_out/flag_latency.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cudecomp_amd as cd  # noqa: E402
from tests.mp import run_ranks  # noqa: E402

out = {}
for n, pd in ((2, (1, 2)), (4, (1, 4)), (8, (1, 8))):
    for backend, bname in ((cd.TRANSPOSE_COMM_NVSHMEM, "nvshmem"), (cd.TRANSPOSE_COMM_NVSHMEM_SM, "nvshmem_sm")):
        for flags, env in (("device", {"CUDECOMP_FLAGS_IN_DEVICE_MEMORY": "1"}), ("host", {})):
            args = {"gdims": (16, 16, 16), "pdims": pd, "kind": 1, "transpose_backend": backend, "cycles": 200}
            res = run_ranks(n, "tests.gpu_bodies", "small_cycle_latency", args, timeout=300, extra_env=env)
            out["%d ranks %s %s flags" % (n, bname, flags)] = round(max(r["us_per_transpose"] for r in res), 1)
            print("%d ranks %-10s %-6s flags: %8.1f us per transpose" % (n, bname, flags, out["%d ranks %s %s flags" % (n, bname, flags)]),
                  file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
