"""Launch a test body on N ranks (one process per rank), the way torchrun would: RANK / WORLD_SIZE /
MASTER_ADDR=127.0.0.1 in the environment.  libcudecomp.so's TCP bootstrap and torch.distributed (gloo)
both rendezvous from these.

    results = run_ranks(4, "tests.bodies", "pencil_info_golden", {"variant": "row_major"})

Each rank runs  <module>.<func>(rank, nranks, args)  and must return a JSON-serialisable value; an
exception (or a non-zero exit) on any rank fails the launch.
"""
import json
import os
import socket
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_port_counter = [0]


def free_port():
    """A TCP port that is free now and is unlikely to be taken before the ranks bind it: picked BELOW the kernel's
    ephemeral range (so no other process is handed it for an outgoing connection in the meantime), walking a
    per-process sequence."""
    for _ in range(2000):
        _port_counter[0] += 1
        port = 10000 + (os.getpid() * 131 + _port_counter[0] * 17) % 20000
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", port))
        except OSError:
            continue
        finally:
            s.close()
        return port
    raise RuntimeError("no free TCP port found")


def gpus_on_this_host():
    """GPU agents the kernel driver lists (no runtime needed)."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(f) as fh:
                for line in fh:
                    if line.startswith("simd_count") and int(line.split()[1]) > 0:
                        n += 1
        except (OSError, ValueError):
            pass
    return n


def kfd_queue_census():
    """Hardware queues the kernel driver (KFD) holds right now, over ALL processes: {"processes", "compute", "sdma",
    "other"}.  Compute queues beyond the device's hardware queue slots put the driver's scheduler into time-slicing
    ("runlist oversubscribed"): the regime DESIGN.md section 9 is about.  Empty dict if the sysfs tree is not readable."""
    root = "/sys/class/kfd/kfd/proc"
    out = {"processes": 0, "compute": 0, "sdma": 0, "other": 0}
    try:
        pids = [d for d in os.listdir(root) if d.isdigit()]
    except OSError:
        return {}
    for pid in pids:
        qdir = os.path.join(root, pid, "queues")
        try:
            queues = os.listdir(qdir)
        except OSError:
            continue
        if queues:
            out["processes"] += 1
        for q in queues:
            try:
                with open(os.path.join(qdir, q, "type")) as f:
                    t = f.read().strip()
            except OSError:
                continue
            # KFD queue types: 0 compute, 1 SDMA, 2 HIQ, 3 DIQ, 4 SDMA over xGMI (strings on newer kernels)
            if t in ("0", "compute"):
                out["compute"] += 1
            elif t in ("1", "4", "sdma", "sdma_xgmi"):
                out["sdma"] += 1
            else:
                out["other"] += 1
    return out


def shared_device_env(nranks, env):
    """Hook for ranks that SHARE a GPU.  Measured in round 3: eight processes with the runtime's
    default of four hardware queues each can push the device into time-slicing its queues (4-5x slower, rarely a wrong
    result in a kernel of ANY kind); GPU_MAX_HW_QUEUES=2 made the 8-rank stress and the complete 8-rank reference matrix
    clean -- but one full-suite run with it hung in an 8-rank sweep, so it is NOT applied by default.  (Round 4: the census
    shows that the value 2 does not lower the processes' queue count at all -- 26 compute queues with and without it -- while
    1 does; the regime is entered above the device's 24 compute-queue slots, profiles/r04_tuning.md.)  The library keeps
    its own stream count at two per process on shared devices instead; set CUDECOMP_TEST_SHARED_GPU_QUEUES=N to try a
    queue limit for runs with more than five ranks per device."""
    want = os.environ.get("CUDECOMP_TEST_SHARED_GPU_QUEUES")
    if not want:
        return
    ngpu = max(gpus_on_this_host(), 1)
    if (nranks + ngpu - 1) // ngpu > 5:
        env.setdefault("GPU_MAX_HW_QUEUES", want)


def run_ranks(nranks, module, func, args=None, timeout=300, extra_env=None, per_rank_env=None):
    port_a, port_b = free_port(), free_port()
    outdir = tempfile.mkdtemp(prefix="cudecomp_mp_")
    procs = []
    for r in range(nranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "WORLD_SIZE": str(nranks), "LOCAL_RANK": str(r), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port_a), "CUDECOMP_BOOTSTRAP_PORT": str(port_b),
                    "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                    "PYTHONPATH": ROOT + os.pathsep + env.get("PYTHONPATH", "")})
        # the staged pipeline of the one-sided transports normally keeps stages above 8 MiB; the tests' small grids must
        # run it with several stages too
        env.setdefault("CUDECOMP_PIPELINE_MIN_STAGE_MIB", "0")
        shared_device_env(nranks, env)
        if extra_env:
            env.update(extra_env)
        if per_rank_env:
            env.update(per_rank_env[r])
        out = os.path.join(outdir, "rank%d.json" % r)
        cmd = [sys.executable, os.path.abspath(__file__), module, func, json.dumps(args or {}), out]
        procs.append((subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT), out))
    results, failures = [], []
    for r, (p, out) in enumerate(procs):
        try:
            log, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q, _ in procs:
                if q.poll() is None:
                    q.kill()
            log, _ = p.communicate()
            failures.append("rank %d timed out\n%s" % (r, log.decode(errors="replace")[-4000:]))
            continue
        if p.returncode != 0 or not os.path.exists(out):
            failures.append("rank %d exit %s\n%s" % (r, p.returncode, log.decode(errors="replace")[-4000:]))
        else:
            with open(out) as f:
                results.append(json.load(f))
    if failures:
        raise AssertionError("\n".join(failures))
    return results


if __name__ == "__main__":
    import importlib
    module, func, args, out = sys.argv[1:5]
    if os.environ.get("CUDECOMP_TEST_RCCL_SHIM"):
        # tests/shim: make the stand-in's nccl* symbols global BEFORE libcudecomp.so is loaded, after torch so that
        # its HIP dependency binds to the runtime already in the process (same order rule as cudecomp_amd.lib())
        import ctypes
        import torch  # noqa: F401
        ctypes.CDLL(os.environ["CUDECOMP_TEST_RCCL_SHIM"], mode=ctypes.RTLD_GLOBAL)
    rank, nranks = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    result = getattr(importlib.import_module(module), func)(rank, nranks, json.loads(args))
    with open(out, "w") as f:
        json.dump(result, f)


def run_binary_ranks(nranks, argv, timeout=300, extra_env=None):
    """Launch a native executable (C / Fortran test twin) on N ranks with the same launcher environment;
    returns the list of per-rank stdout+stderr texts, raising on any non-zero exit."""
    import uuid
    port_a, port_b = free_port(), free_port()
    job = uuid.uuid4().hex[:16]  # scratch-file namespace of this launch (tests/native, tests/fortran verdict files)
    procs = []
    for r in range(nranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "WORLD_SIZE": str(nranks), "LOCAL_RANK": str(r), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port_a), "CUDECOMP_BOOTSTRAP_PORT": str(port_b),
                    "HSA_ENABLE_IPC_MODE_LEGACY": "0", "CUDECOMP_TEST_JOB": job})
        env.setdefault("CUDECOMP_PIPELINE_MIN_STAGE_MIB", "0")  # small grids still run several pipeline stages
        shared_device_env(nranks, env)
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([str(a) for a in argv], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    logs, failures = [], []
    for r, p in enumerate(procs):
        try:
            log, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                if q.poll() is None:
                    q.kill()
            log, _ = p.communicate()
            failures.append("rank %d timed out\n%s" % (r, log.decode(errors="replace")[-4000:]))
            continue
        text = log.decode(errors="replace")
        logs.append(text)
        if p.returncode != 0:
            failures.append("rank %d exit %s\n%s" % (r, p.returncode, text[-4000:]))
    if failures:
        # keep what every rank said where gpurun merges it back: the lines that matter (DIAG, failing commands) are rarely in the tail
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "ranks_failure_%s_%d.log" % (os.path.basename(str(argv[0])), os.getpid())), "w") as f:
                for r, text in enumerate(logs):
                    keep, lines_r = set(), text.splitlines()
                    for i, line in enumerate(lines_r):
                        if line.startswith("DIAG") or "differ" in line or "CUDECOMP:" in line or line.strip() == "FAILED" or "Input gate" in line:
                            keep.update(range(max(0, i - 2), min(len(lines_r), i + 2)))
                    f.write("===== rank %d (%d lines)\n" % (r, len(lines_r)))
                    f.write("\n".join(lines_r[i][:1200] for i in sorted(keep)[:300]) + "\n")
        except OSError:
            pass
        raise AssertionError("\n".join(failures))
    return logs
