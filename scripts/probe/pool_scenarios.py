"""Scenarios for the rank pool (tests/mp.py): a list of jobs, each "ranks:PRxPC:bBACKEND[:body]", run in order on the pool; prints
which jobs fail.  Bisects what a pooled sequence needs to go wrong.
    python scripts/probe/pool_scenarios.py "8:2x4:b6:cycle 4:2x2:b6 4:2x2:b7 4:2x2:b8 4:1x4:b6" [K=V ...]
body: graph (default, gpu_bodies.graph_cycle) | cycle (cycle_exact) | chain (transpose_chain).  CUDECOMP_TEST_POOL_KEEP_LOGS=dir
keeps the workers' logs."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import mp  # noqa: E402


def main():
    jobs = sys.argv[1].split()
    env = dict(a.split("=", 1) for a in sys.argv[2:])
    out = {"scenario": sys.argv[1], "env": env, "results": []}
    for spec in jobs:
        parts = spec.split(":")
        n, (pr, pc), backend = int(parts[0]), map(int, parts[1].split("x")), int(parts[2][1:])
        body = parts[3] if len(parts) > 3 else "graph"
        args = {"gdims": (96, 80, 112), "pdims": (pr, pc), "kind": 1, "ac": (1, 1, 1), "transpose_backend": backend, "replays": 2}
        variant = None
        if body.startswith("graph-"):   # graph-nc (no capture), graph-td (torch data pencils), graph-ds (default stream)
            body, variant = "graph", body[6:]
            args[{"nc": "no_capture", "td": "torch_data", "ds": "default_stream"}[variant]] = True
        fn = {"graph": "graph_cycle", "cycle": "cycle_exact", "chain": "transpose_chain", "probe": "pool_probe"}[body]
        if body == "probe":  # processes with a GPU context, no library handle
            args = {"handle": False}
        try:
            res = mp.run_ranks(n, "tests.gpu_bodies", fn, args, timeout=120, extra_env=env)
            fails = [f[:150] for r in res for f in (r if isinstance(r, list) else r.get("failures", []))]
        except AssertionError as e:
            fails = ["launch failed: " + str(e)[-300:]]
        out["results"].append({"job": spec, "failures": len(fails), "first": fails[:2]})
    mp.pool_stop()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
