"""Does a library handle that serves SEVERAL grid descriptors in a row (different process grids, transports, user graphs) stay
correct?  The rank pool of tests/mp.py makes the GPU suite do exactly that; its first run saw one wrong eager cycle
(graph_cycle, 1 x 4 grid, after three 2 x 2 jobs on the same handle).  Arms: the same job sequence in FRESH processes through
the `many` runner (handle reuse only), and through the pool with eight workers alive (handle reuse + idle contexts + worlds
rebuilt inside living processes).  usage: python scripts/probe/pool_sequence_stress.py [repetitions] [pool|fresh|both] [K=V ...]
(K=V: CUDECOMP_* switches for the pooled jobs; what the failure turned out to be: profiles/r06_pooled_suite_failure.md)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cudecomp_amd as cd  # noqa: E402
from tests import mp  # noqa: E402


def sequence():
    jobs = []
    for pdims in ((2, 2), (1, 4)):
        for backend in (cd.TRANSPOSE_COMM_NVSHMEM, cd.TRANSPOSE_COMM_NVSHMEM_PL, cd.TRANSPOSE_COMM_NVSHMEM_SM):
            jobs.append({"fn": "graph_cycle_failures", "id": "P%dx%d_b%d" % (pdims[0], pdims[1], backend),
                         "args": {"gdims": (96, 80, 112), "pdims": pdims, "kind": 1, "ac": (1, 1, 1), "transpose_backend": backend, "replays": 3}})
    return jobs


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    which = sys.argv[2] if len(sys.argv) > 2 else "both"
    env = dict(a.split("=", 1) for a in sys.argv[3:])
    out = {"reps": reps, "env": env}
    if which in ("fresh", "both"):
        t0 = time.time()
        bad = []
        for r in range(reps):
            try:
                for fails in mp.run_ranks(4, "tests.gpu_bodies", "many", {"jobs": sequence()}, timeout=600, fresh=True, extra_env=env):
                    bad += fails
            except AssertionError as e:
                bad.append("launch failed: " + str(e)[-500:])
        out["fresh_many"] = {"failures": bad[:6], "n_failures": len(bad), "seconds": round(time.time() - t0, 1)}
    if which in ("pool", "both"):
        t0 = time.time()
        bad, jobs_run = [], 0
        big = {"gdims": (64, 48, 80), "pdims": (2, 4), "kind": 1, "ac": (1, 1, 1), "transpose_backend": cd.TRANSPOSE_COMM_NVSHMEM}
        for r in range(reps):
            try:
                mp.run_ranks(8, "tests.gpu_bodies", "cycle_exact", big, timeout=300, extra_env=env)   # eight workers alive, an 8-rank world before
                for job in sequence():
                    jobs_run += 1
                    for res in mp.run_ranks(4, "tests.gpu_bodies", "graph_cycle", job["args"], timeout=300, extra_env=env):
                        bad += ["%s: %s" % (job["id"], f[:160]) for f in res["failures"]]
            except AssertionError as e:
                bad.append("launch failed: " + str(e)[-500:])
        mp.pool_stop()
        out["pool"] = {"failures": bad[:6], "n_failures": len(bad), "jobs": jobs_run, "seconds": round(time.time() - t0, 1),
                       "stats": dict(mp.pool_stats)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
