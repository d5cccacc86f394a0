#!/usr/bin/env python3
"""Launch benchmark/fft3d_benchmark on N ranks (one process per rank, ranks map to GPUs round-robin) and
reduce the per-rank JSON lines: time = max over ranks, errors = max over ranks.

    python benchmark/run_fft3d.py --ranks 8 -- --gx 1024 --gy 1024 --gz 1024 --pr 2 --pc 4 -o
"""
import argparse
import json
import os
import socket
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run(nranks, prog_args, timeout=900):
    exe = os.path.join(HERE, "fft3d_benchmark")
    subprocess.check_call(["make", "-s", "-C", HERE, "fft3d_benchmark"])
    port = free_port()
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(nranks), MASTER_ADDR="127.0.0.1",
                   CUDECOMP_BOOTSTRAP_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([exe] + prog_args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    recs, logs = [], []
    for p in procs:
        out, err = p.communicate(timeout=timeout)
        logs.append(out.decode(errors="replace") + err.decode(errors="replace"))
        for line in out.decode(errors="replace").splitlines():
            if line.startswith("{"):
                recs.append(json.loads(line))
    if len(recs) != nranks or any(p.returncode != 0 for p in procs):
        raise RuntimeError("fft3d_benchmark failed:\n" + "\n".join(logs))
    r0 = dict(recs[0])
    r0.pop("rank")
    r0["ms_avg"] = max(r["ms_avg"] for r in recs)
    r0["ms_min"] = max(r["ms_min"] for r in recs)
    r0["ms_max"] = max(r["ms_max"] for r in recs)
    n = r0["gdims"][0] * r0["gdims"][1] * r0["gdims"][2]
    import math
    r0["gflops"] = round(5.0 * n * math.log2(n) * 1e-9 / (r0["ms_avg"] * 1e-3), 1)
    r0["roundtrip_max_abs_err"] = max(r["roundtrip_max_abs_err"] for r in recs)
    r0["spectrum_rel_err"] = max(r["spectrum_rel_err"] for r in recs)
    # the ranks' parts of the weighted spectrum checksum add up to a number that does not depend on the decomposition
    r0["spectrum_checksum"] = [sum(r["spectrum_checksum"][0] for r in recs), sum(r["spectrum_checksum"][1] for r in recs)]
    r0["spectrum_abs_sum"] = sum(r["spectrum_abs_sum"] for r in recs)
    r0["ok"] = all(r["ok"] for r in recs)
    return r0, logs


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=1)
    ap.add_argument("rest", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    rest = [x for x in a.rest if x != "--"]
    rec, _ = run(a.ranks, rest)
    print(json.dumps(rec))
    sys.exit(0 if rec["ok"] else 1)
