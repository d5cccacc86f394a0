"""bench.py's per-dtype table on its own (extra.dtypes), e.g. to A/B a tuning switch:
    CUDECOMP_TILE_SHAPE=1 python scripts/probe/dtype_table.py [fp32|complex64|complex128 ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402

import bench  # noqa: E402
import cudecomp_amd as cd  # noqa: E402

torch.zeros(1, device="cuda")
h = cd.cudecompInit()
want = sys.argv[1:]
t = bench.dtype_table(cd, torch, h, torch.cuda.current_stream().cuda_stream, only=want or None)
for row in t["rows"]:
    if want and row["dtype"] not in want:
        continue
    print(json.dumps({k: row.get(k) for k in ("dtype", "layout", "cycle_ms", "min_frac", "round_trip_ok", "error")}
                     | {"per_op": [(o["op"], o["ms"], o["ms_min"], o["ms_max"], o["frac"], o["kernel"]) for o in row.get("per_op", [])]}))
cd.cudecompFinalize(h)
