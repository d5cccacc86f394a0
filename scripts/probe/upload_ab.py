"""Round 5: the experiment that decides the "upload hypothesis" behind the rare wrong results of ranks sharing a GPU
(two sightings in round 4, both in builds with the oversized device code).

    python scripts/probe/upload_ab.py OUTDIR SECONDS_PER_ARM [--lib DIR] [--ranks 8] [--arms default,pinned,sync]

State reproduced: the `make TUNING_VARIANTS=1` library (cudecomp_amd/lib_tuning: 0.72 MB of device code), 8 ranks ALONE on
the device (24 of 24 compute-queue slots, no holder process), the reference's own 8-GPU case lists halo_test_mix_cc and
transpose_test_mix_cc (R32, the RCCL enums through the test stand-in, as the runs that failed), repeated in seeded shuffles.

Three arms, same case files, differing ONLY in how the test program uploads the pencil (CUDECOMP_TEST_UPLOAD):
    default  synchronous hipMemcpy from pageable memory (what the reference's programs do)
    pinned   pinned staging buffer + hipMemcpyAsync + hipStreamSynchronize on the null stream
    sync     default + hipDeviceSynchronize
Every case passes the input-integrity gate of tests/native/native_test.h: a checksum kernel on the library's stream reads the
uploaded pencil BEFORE the call; downloads are checked against the device's view AFTER it.

STOP RULE (fixed before the run): every arm gets SECONDS_PER_ARM of wall clock (half halos, half transposes) or 150,000
cases, whichever comes first; the programs are killed at the deadline and the completed cases counted.  No arm is extended,
no arm is repeated because of its result.  Reading, fixed in advance:
    A  a wrong result (or a gate trip) with "input stale before call"  -> the upload, not the library
    B  a wrong result with the gate passed ("the upload was intact")   -> the library's (or the platform's) data path: the
       DIAG lines name the hop / halo update
    C  nothing in any arm                                              -> recorded; the gate stays on
One JSON line per (arm, program) on stdout."""
import json
import os
import random
import re
import signal
import subprocess
import sys
import threading
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.mp import free_port, kfd_queue_census  # noqa: E402
from tests.test_gpu_runner_cases import load_cases  # noqa: E402

SHIM = os.path.join(ROOT, "tests", "shim", "libfake_rccl.so")
MAX_CASES = 150000
KEEP = [None]  # directory for the results and the excerpts (small); the bulky per-rank logs go to a scratch directory


def case_file(lines, n, seed):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        chunk = list(lines)
        rng.shuffle(chunk)
        out += chunk
    return out[:n]


def run(outdir, name, binary, nranks, lines, env_extra, seconds):
    path = os.path.join(outdir, name + "_cases.txt")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    port_a, port_b, job = free_port(), free_port(), uuid.uuid4().hex[:16]
    procs, logs = [], []
    t0 = time.time()
    for r in range(nranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "WORLD_SIZE": str(nranks), "LOCAL_RANK": str(r), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port_a), "CUDECOMP_BOOTSTRAP_PORT": str(port_b), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                    "CUDECOMP_TEST_JOB": job, "CUDECOMP_PIPELINE_MIN_STAGE_MIB": "0"})
        if os.path.exists(SHIM):
            env["LD_PRELOAD"] = SHIM
        env.update(env_extra)
        log = open(os.path.join(outdir, "%s_rank%d.log" % (name, r)), "w")
        logs.append(log)
        procs.append(subprocess.Popen([binary, "--testfile", path], env=env, cwd=ROOT, stdout=log, stderr=subprocess.STDOUT))
    census, stop = {}, threading.Event()

    def sampler():
        while not stop.is_set():
            for k, v in kfd_queue_census().items():
                census[k] = max(census.get(k, 0), v)
            stop.wait(1.0)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    killed = False
    for p in procs:
        try:
            p.wait(timeout=max(1.0, seconds - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            killed = True
            break
    if killed:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGKILL)
        for p in procs:
            p.wait()
    wall = time.time() - t0
    stop.set()
    th.join()
    for log in logs:
        log.close()
    text0 = open(os.path.join(outdir, name + "_rank0.log"), errors="replace").read()
    passed, failed = text0.count(" PASSED"), text0.count(" FAILED")
    diag, gate = [], []
    for r in range(nranks):
        for line in open(os.path.join(outdir, "%s_rank%d.log" % (name, r)), errors="replace"):
            if line.startswith("DIAG") or "cells differ" in line or "CUDECOMP:ERROR" in line:
                diag.append(line.strip()[:900])
            elif line.startswith("Input gate"):
                gate.append(line.strip())
    failing, cmd = [], None
    for line in text0.splitlines():
        if line.startswith("command:"):
            cmd = line.split("_test_R32 ", 1)[-1].strip()
        elif line.strip() == "FAILED" and cmd is not None:
            failing.append(cmd[-220:])
    stale = sum(1 for d in diag if "input stale before call" in d)
    rec = {"arm": name, "program": os.path.basename(binary), "ranks": nranks, "env": env_extra, "cases_in_file": len(lines),
           "completed": passed + failed, "passed": passed, "failed": failed, "gate_trips_input_stale": stale,
           "download_mismatches": sum(1 for d in diag if "download differs" in d),
           "interior_overwritten": sum(1 for d in diag if "interior overwritten" in d),
           "killed_at_deadline": killed, "wall_s": round(wall, 1),
           "ms_per_case": round(1000 * wall / max(1, passed + failed), 2), "kfd_queues_max": census, "failing": failing[:20],
           "diag": diag[:60], "gate_reports": gate[:8], "exit_codes": [p.returncode for p in procs]}
    print(json.dumps(rec), flush=True)
    # The per-rank logs are bulky (a line or two per case: 15-20 MB per program run) and gpurun copies back at most 64 MiB:
    # they live in a scratch directory; what says something (DIAG lines, failing cases with their neighbourhood, the gate's
    # report) is written as a small excerpt next to the results.
    if failed or stale or rec["download_mismatches"] or rec["interior_overwritten"]:
        with open(os.path.join(KEEP[0], name + "_excerpt.log"), "w") as f:
            for r in range(nranks):
                lines_r = open(os.path.join(outdir, "%s_rank%d.log" % (name, r)), errors="replace").read().splitlines()
                f.write("===== rank %d (%d lines)\n" % (r, len(lines_r)))
                want = set()
                for i, line in enumerate(lines_r):
                    if line.startswith("DIAG") or "cells differ" in line or "CUDECOMP:" in line or line.strip() == "FAILED" or line.startswith("Input gate"):
                        want.update(range(max(0, i - 3), min(len(lines_r), i + 2)))
                for i in sorted(want)[:400]:
                    f.write(lines_r[i][:1000] + "\n")
    for r in range(nranks):
        os.unlink(os.path.join(outdir, "%s_rank%d.log" % (name, r)))
    os.unlink(path)
    return rec


def main():
    keepdir, per_arm = sys.argv[1], float(sys.argv[2])
    opts = sys.argv[3:]
    import tempfile
    os.makedirs(keepdir, exist_ok=True)
    KEEP[0] = keepdir
    outdir = tempfile.mkdtemp(prefix="upload_ab_")

    def opt(name, dflt):
        return opts[opts.index(name) + 1] if name in opts else dflt

    libdir = opt("--lib", os.path.join(ROOT, "cudecomp_amd", "lib_tuning"))
    nranks = int(opt("--ranks", "8"))
    arms = opt("--arms", "default,pinned,sync").split(",")
    cases = load_cases(8 if nranks == 8 else 4)
    halo, trans = cases["halo_test_mix_cc"], cases["transpose_test_mix_cc"]
    size = subprocess.run("readelf -S -W %s/libcudecomp.so | grep ' .hip_fatbin '" % libdir, shell=True, capture_output=True, text=True).stdout.split()
    print(json.dumps({"library": libdir, "hip_fatbin_bytes": int(size[5], 16) if len(size) > 5 else None, "ranks": nranks,
                      "seconds_per_arm": per_arm, "max_cases_per_program": MAX_CASES, "halo_mix_cases": len(halo),
                      "transpose_mix_cases": len(trans)}), flush=True)
    base_env = {"LD_LIBRARY_PATH": libdir + os.pathsep + os.environ.get("LD_LIBRARY_PATH", "")}
    upload = {"default": {}, "pinned": {"CUDECOMP_TEST_UPLOAD": "pinned"}, "sync": {"CUDECOMP_TEST_UPLOAD": "sync"}}
    native = os.path.join(ROOT, "tests", "native", "build")
    totals = {}
    # the arms are interleaved in slices so that a drift of the box over the run hits all three alike
    slices = int(opt("--slices", "2"))
    for s in range(slices):
        for arm in arms:
            env = dict(base_env, **upload[arm])
            for prog, lines in (("halo_test_R32", halo), ("transpose_test_R32", trans)):
                rec = run(outdir, "%s_%s_slice%d" % (arm, prog.split("_")[0], s), os.path.join(native, prog), nranks,
                          case_file(lines, MAX_CASES // slices, seed=77 + s), env, per_arm / (2 * slices))
                t = totals.setdefault(arm, {"completed": 0, "failed": 0, "gate_trips_input_stale": 0, "download_mismatches": 0,
                                            "interior_overwritten": 0, "wall_s": 0.0})
                for k in t:
                    t[k] += rec[k]
    print(json.dumps({"totals": totals}), flush=True)


if __name__ == "__main__":
    main()
