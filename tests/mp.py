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
_port_lock = __import__("threading").Lock()


def free_port():
    """A TCP port that is free now and is unlikely to be taken before the ranks bind it: picked BELOW the kernel's
    ephemeral range (so no other process is handed it for an outgoing connection in the meantime), walking a
    per-process sequence."""
    for _ in range(2000):
        with _port_lock:
            _port_counter[0] += 1
            count = _port_counter[0]
        port = 10000 + (os.getpid() * 131 + count * 17) % 20000
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


# ---- the rank pool -----------------------------------------------------------------------------------------------------
# A launch of N fresh rank processes costs ~10 s before the first library call (N Python start-ups, N torch imports, N GPU
# contexts); the GPU suite makes hundreds of launches, and that start-up was two thirds of its wall time (round 5: the driver's
# run was cut at its limit).  GPU bodies therefore run on a POOL of long-lived rank workers: worker r imports everything and
# opens the GPU once and then serves rank r of every job.  What a job may still change per launch -- world size, the
# CUDECOMP_* switches the library reads when a handle is created -- is applied by finalizing the workers' handle and
# creating a new one (collective, driven from here); what only a fresh process can change (a preloaded RCCL stand-in,
# runtime-level variables) is part of the pool's signature: a job with another signature replaces the pool.  One failure or
# timeout kills the whole pool (the next job starts a new one), so a wedged rank never leaks into later tests.
# Never pooled: CPU bodies (gloo), per-rank environments, and the bodies in FRESH_FUNCS, which are ABOUT process-level state.
POOL_SWITCH = "CUDECOMP_TEST_RANK_POOL"   # =0: every launch in fresh processes, as before round 6
FRESH_FUNCS = {"absent_peer", "queue_census", "pool_pressure", "pool_behaviour", "malloc_timing", "link_info", "perf_report"}
# extra_env keys that only matter when a HANDLE is created or later (read through getenv by libcudecomp.so at those points)
_HANDLE_LEVEL_PREFIX = "CUDECOMP_"
_PROCESS_LEVEL = {"CUDECOMP_TEST_RCCL_SHIM", "CUDECOMP_AMD_LIBRARY", "CUDECOMP_DISABLE_MPI_DISCOVERY"}


def _split_env(extra_env):
    proc, handle = {}, {}
    for k, v in (extra_env or {}).items():
        if k.startswith(_HANDLE_LEVEL_PREFIX) and k not in _PROCESS_LEVEL:
            handle[k] = str(v)
        else:
            proc[k] = str(v)
    return proc, handle


class _Pool:
    def __init__(self, proc_env):
        self.proc_env = dict(proc_env)
        self.workers = []   # (Popen, read end of the answer pipe, log path)
        self.buf = []       # bytes of a worker's answers not consumed yet
        self.world = None   # (nranks, handle-env items) of the handle the workers 0..nranks-1 hold
        self.ports = None
        self.dir = tempfile.mkdtemp(prefix="cudecomp_pool_")
        self.seq = 0
        self.jobs = 0

    def grow(self, n):
        first = len(self.workers)
        for r in range(first, n):
            env = dict(os.environ)
            env.update({"LOCAL_RANK": str(r), "MASTER_ADDR": "127.0.0.1", "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                        "PYTHONPATH": ROOT + os.pathsep + env.get("PYTHONPATH", "")})
            env.setdefault("CUDECOMP_PIPELINE_MIN_STAGE_MIB", "0")
            env[POOL_SWITCH] = "0"  # a worker never starts a pool of its own
            shared_device_env(8, env)
            env.update(self.proc_env)
            rfd, wfd = os.pipe()
            log = os.path.join(self.dir, "worker%d.log" % r)
            with open(log, "wb") as lf:
                p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(wfd)], env=env, cwd=ROOT,
                                     stdin=subprocess.PIPE, stdout=lf, stderr=subprocess.STDOUT, pass_fds=(wfd,))
            os.close(wfd)
            self.workers.append((p, rfd, log))
            self.buf.append(b"")
        for r in range(first, n):  # wait until every new worker has its GPU context (first import of torch: up to minutes)
            if self._read(r, 600) is None:
                raise AssertionError("pool worker %d did not come up\n%s" % (r, self._log_tail(r, 0)))

    def _read(self, r, timeout):
        """Next answer line of worker r as an object; None on timeout or when the worker is gone."""
        import select
        import time
        p, rfd, _ = self.workers[r]
        deadline = time.monotonic() + timeout
        while b"\n" not in self.buf[r]:
            ready, _, _ = select.select([rfd], [], [], max(0.0, deadline - time.monotonic()))
            if not ready:
                return None
            chunk = os.read(rfd, 65536)
            if not chunk:
                return None
            self.buf[r] += chunk
        line, self.buf[r] = self.buf[r].split(b"\n", 1)
        return json.loads(line)

    def _log_tail(self, r, start):
        try:
            with open(self.workers[r][2], "rb") as f:
                f.seek(start)
                return f.read().decode(errors="replace")[-4000:]
        except OSError:
            return ""

    def _send(self, r, msg):
        p = self.workers[r][0]
        p.stdin.write((json.dumps(msg) + "\n").encode())
        p.stdin.flush()

    def _finalize_world(self):
        if self.world is None:
            return True
        n = self.world[0]
        self.seq += 1
        for r in range(n):
            self._send(r, {"cmd": "finalize", "id": self.seq})
        ok = True
        for r in range(n):
            ans = self._read(r, 120)
            ok = ok and ans is not None and ans.get("ok")
        self.world = None
        return ok

    def run(self, nranks, module, func, args, timeout, handle_env):
        import time
        if len(self.workers) < nranks:
            self.grow(nranks)
        world = (nranks, tuple(sorted(handle_env.items())))
        if self.world != world:
            if not self._finalize_world():
                raise AssertionError("pool: finalizing the previous world failed")
            self.ports = (free_port(), free_port())
            self.world = world
        self.seq += 1
        self.jobs += 1
        outs, starts = [], []
        for r in range(nranks):
            out = os.path.join(self.dir, "job%d_rank%d.json" % (self.seq, r))
            outs.append(out)
            try:
                starts.append(os.path.getsize(self.workers[r][2]))
            except OSError:
                starts.append(0)
            env = {"RANK": str(r), "WORLD_SIZE": str(nranks), "LOCAL_RANK": str(r), "MASTER_PORT": str(self.ports[0]),
                   "CUDECOMP_BOOTSTRAP_PORT": str(self.ports[1])}
            env.update(handle_env)
            self._send(r, {"cmd": "job", "id": self.seq, "module": module, "func": func, "args": args or {}, "env": env, "out": out})
        deadline = time.monotonic() + timeout
        results, failures = [], []
        for r in range(nranks):
            ans = self._read(r, max(0.0, deadline - time.monotonic()))
            if ans is None:
                failures.append("rank %d timed out or died\n%s" % (r, self._log_tail(r, starts[r])))
            elif not ans.get("ok") or not os.path.exists(outs[r]):
                failures.append("rank %d failed\n%s" % (r, self._log_tail(r, starts[r])))
            else:
                with open(outs[r]) as f:
                    results.append(json.load(f))
                os.unlink(outs[r])
        if failures:
            raise AssertionError("\n".join(failures))
        return results

    def stop(self, kill=False):
        import shutil
        if not kill:
            for p, _, _ in self.workers:
                try:
                    p.stdin.close()   # EOF: finalize (collective among the members of the current world) and leave
                except OSError:
                    pass
            for p, _, _ in self.workers:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    kill = True
        for p, rx, _ in self.workers:
            if p.poll() is None:
                p.kill()
                p.wait()
            try:
                os.close(rx)
            except OSError:
                pass
        self.workers, self.buf = [], []
        keep = os.environ.get("CUDECOMP_TEST_POOL_KEEP_LOGS")
        if keep:  # debugging: what every worker printed, job by job
            try:
                os.makedirs(keep, exist_ok=True)
                for name in os.listdir(self.dir):
                    if name.endswith(".log"):
                        shutil.copy(os.path.join(self.dir, name), os.path.join(keep, "%s_%s" % (os.path.basename(self.dir), name)))
            except OSError:
                pass
        shutil.rmtree(self.dir, ignore_errors=True)


_pool = [None]
pool_stats = {"started": 0, "jobs": 0, "fresh_launches": 0}


def pool_stop(kill=False):
    """Ends the rank pool, if one is alive (before anything else puts processes on the GPU: a forked in-process child, native
    test programs, fresh rank launches -- nine processes with a GPU context are one more than the device serves without
    time-slicing them)."""
    if _pool[0] is not None:
        pool, _pool[0] = _pool[0], None
        pool_stats["jobs"] += pool.jobs
        pool.stop(kill)


def _pool_run(nranks, module, func, args, timeout, extra_env):
    proc_env, handle_env = _split_env(extra_env)
    if _pool[0] is not None and _pool[0].proc_env != proc_env:
        pool_stop()
    if _pool[0] is None:
        import atexit
        _pool[0] = _Pool(proc_env)
        pool_stats["started"] += 1
        if pool_stats["started"] == 1:
            atexit.register(pool_stop, True)
    try:
        return _pool[0].run(nranks, module, func, args, timeout, handle_env)
    except BaseException:
        pool_stop(kill=True)  # whatever state the ranks are in, the next job starts from fresh processes
        raise


# The one platform error a launch gets a second attempt for: the runtime refuses to EXPORT a fresh device allocation over IPC
# ("hipIpcGetMemHandle failed: invalid argument"; on the importing ranks "a peer rank could not export its buffer over IPC").  Seen
# twice in about twenty suite runs of round 6, both times with eight processes on the one GPU creating and releasing shared workspaces
# (gpurun_out/r06_sixth: subcomm_test; profiles/r06_ipc_export_refused.log: the CUDECOMP_WORKSPACE_POOL_MIB=0 arm of the switch sweep);
# the library had already tried four allocations at other addresses (csrc/transport.cc workspaceAllocRaw) and goes on without the
# one-sided transport, which the backends under test need.  Not a property of the kernels, plans or transports.
IPC_EXPORT_REFUSED = ("could not export its buffer over IPC", "hipIpcGetMemHandle failed")


def _second_attempt_if_export_refused(what, launch):
    try:
        return launch()
    except AssertionError as e:
        if not any(sig in str(e) for sig in IPC_EXPORT_REFUSED):
            raise
        print("%s: the runtime refused to export a fresh workspace over IPC (platform hiccup, tests/mp.py IPC_EXPORT_REFUSED); "
              "one more attempt, in new processes" % what)
        pool_stop(kill=True)
        pool_stats["second_attempts"] = pool_stats.get("second_attempts", 0) + 1
        return launch()


def run_ranks(nranks, module, func, args=None, timeout=300, extra_env=None, per_rank_env=None, fresh=None):
    """`func(rank, nranks, args)` of `module` on N ranks (the rank pool, or fresh processes); returns the per-rank results, raises
    AssertionError with the ranks' output on any failure.  The only repetition: _second_attempt_if_export_refused."""
    return _second_attempt_if_export_refused("%s.%s on %d ranks" % (module, func, nranks), lambda: _run_ranks_once(
        nranks, module, func, args, timeout, extra_env, per_rank_env, fresh))


def _run_ranks_once(nranks, module, func, args=None, timeout=300, extra_env=None, per_rank_env=None, fresh=None):
    if fresh is None:
        fresh = (os.environ.get(POOL_SWITCH, "1") == "0" or module != "tests.gpu_bodies" or func in FRESH_FUNCS or
                 per_rank_env is not None or nranks > 8)
    if not fresh:
        return _pool_run(nranks, module, func, args, timeout, extra_env)
    pool_stop()
    pool_stats["fresh_launches"] += 1
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


def _load_shim_first():
    if os.environ.get("CUDECOMP_TEST_RCCL_SHIM"):
        # tests/shim: make the stand-in's nccl* symbols global BEFORE libcudecomp.so is loaded, after torch so that
        # its HIP dependency binds to the runtime already in the process (same order rule as cudecomp_amd.lib())
        import ctypes
        import torch  # noqa: F401
        ctypes.CDLL(os.environ["CUDECOMP_TEST_RCCL_SHIM"], mode=ctypes.RTLD_GLOBAL)


def _worker_main(index, result_fd):
    """A long-lived rank of the pool (see _Pool): imports torch and the library ONCE, opens the GPU once, then serves
    jobs from stdin -- one JSON object per line -- and answers on `result_fd`.  Commands: {"cmd": "job", ...},
    {"cmd": "finalize"} (collective: every member of the current world gets it), EOF = finalize and leave."""
    import gc
    import importlib
    import traceback
    res = os.fdopen(result_fd, "w", buffering=1)
    _load_shim_first()
    import torch
    from tests import gpu_bodies as B
    import cudecomp_amd as cd
    if torch.cuda.is_available():  # (the pool's own CPU tests run the protocol without a GPU)
        torch.cuda.set_device(index % max(torch.cuda.device_count(), 1))
        torch.zeros(1, device="cuda")
    res.write(json.dumps({"ready": index}) + "\n")

    def finalize():
        if B._HANDLE is not None:
            h, B._HANDLE = B._HANDLE, None
            cd.cudecompFinalize(h)

    for line in sys.stdin:
        msg = json.loads(line)
        if msg["cmd"] == "finalize":
            try:
                finalize()
                res.write(json.dumps({"id": msg["id"], "ok": True}) + "\n")
            except Exception:  # noqa: BLE001
                traceback.print_exc()
                res.write(json.dumps({"id": msg["id"], "ok": False}) + "\n")
            continue
        saved = {}
        for k, v in msg["env"].items():
            saved[k] = os.environ.get(k)
            os.environ[k] = v
        ok = True
        try:
            print("=== job %s: %s.%s rank %s of %s" % (msg["id"], msg["module"], msg["func"], msg["env"]["RANK"], msg["env"]["WORLD_SIZE"]), flush=True)
            fn = getattr(importlib.import_module(msg["module"]), msg["func"])
            result = fn(int(msg["env"]["RANK"]), int(msg["env"]["WORLD_SIZE"]), msg["args"])
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            with open(msg["out"], "w") as f:
                json.dump(result, f)
        except BaseException:  # noqa: BLE001
            traceback.print_exc()
            ok = False
        sys.stdout.flush()
        sys.stderr.flush()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        gc.collect()
        try:
            if torch.cuda.is_available() and torch.cuda.memory_reserved() > (24 << 30):  # a full-size case: do not sit on its buffers while others run
                torch.cuda.empty_cache()
        except Exception:  # noqa: BLE001
            pass
        res.write(json.dumps({"id": msg["id"], "ok": ok}) + "\n")
    try:
        finalize()
    except Exception:  # noqa: BLE001
        pass


if __name__ == "__main__":
    import importlib
    if sys.argv[1] == "--worker":
        _worker_main(int(sys.argv[2]), int(sys.argv[3]))
        sys.exit(0)
    module, func, args, out = sys.argv[1:5]
    _load_shim_first()
    rank, nranks = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    result = getattr(importlib.import_module(module), func)(rank, nranks, json.loads(args))
    with open(out, "w") as f:
        json.dump(result, f)


def run_binary_ranks(nranks, argv, timeout=300, extra_env=None):
    """Launch a native executable (C / Fortran test twin) on N ranks with the same launcher environment;
    returns the list of per-rank stdout+stderr texts, raising on any non-zero exit.  The only repetition:
    _second_attempt_if_export_refused."""
    return _second_attempt_if_export_refused("%s on %d ranks" % (os.path.basename(str(argv[0])), nranks),
                                             lambda: _run_binary_ranks_once(nranks, argv, timeout, extra_env))


def _run_binary_ranks_once(nranks, argv, timeout=300, extra_env=None):
    import uuid
    pool_stop()
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


def run_binary_groups(jobs, max_ranks=8, collect_errors=False):
    """Several INDEPENDENT native launches side by side -- jobs = [(nranks, argv, timeout, extra_env), ...] -- as long as
    together they stay within `max_ranks` processes on the GPU (the device serves eight processes without time-slicing them,
    DESIGN.md section 9).  The test programs spend most of their time on the host (filling and comparing pencils, the control
    plane of a descriptor per case), so two four-rank groups take about as long as one.  Returns the per-job logs in order;
    raises the first failure after all jobs have ended (collect_errors: returns the AssertionError in the job's place)."""
    from concurrent.futures import ThreadPoolExecutor
    pool_stop()
    results, errors = [None] * len(jobs), []
    batch, used = [], 0

    def flush():
        if not batch:
            return
        with ThreadPoolExecutor(max_workers=len(batch)) as ex:
            futs = [(i, ex.submit(run_binary_ranks, n, argv, timeout, env)) for i, (n, argv, timeout, env) in batch]
            for i, f in futs:
                try:
                    results[i] = f.result()
                except AssertionError as e:
                    errors.append(e)
                    results[i] = e
        batch.clear()

    for i, job in enumerate(jobs):
        if used + job[0] > max_ranks:
            flush()
            used = 0
        batch.append((i, job))
        used += job[0]
    flush()
    if errors and not collect_errors:
        raise errors[0]
    return results
