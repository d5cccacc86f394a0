"""Do the processes of a native four-rank list accumulate file descriptors when every case creates and releases IPC-shared
workspaces (CUDECOMP_WORKSPACE_POOL_MIB=0)?  Samples /proc/<pid>/fd of the running test programs once a second.
    python scripts/probe/ipc_fd_growth.py            (prints one JSON line per arm: pool off / pool on)"""
import json
import os
import resource
import sys
import tempfile
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import mp  # noqa: E402
from tests import test_gpu_native_sweep as S  # noqa: E402
from tests.test_gpu_native import _binary  # noqa: E402


def fd_counts(name):
    out = {}
    for pid in os.listdir("/proc"):
        if not pid.isdigit():
            continue
        try:
            if os.path.basename(os.readlink("/proc/%s/exe" % pid)) == name:
                fds = os.listdir("/proc/%s/fd" % pid)
                kinds = {}
                for fd in fds:
                    try:
                        t = os.readlink("/proc/%s/fd/%s" % (pid, fd))
                    except OSError:
                        continue
                    k = "dmabuf" if "dmabuf" in t else ("socket" if t.startswith("socket") else ("kfd/dri" if "/dev/" in t else "other"))
                    kinds[k] = kinds.get(k, 0) + 1
                out[int(pid)] = (len(fds), kinds)
        except OSError:
            pass
    return out


def main():
    import itertools
    lines = [S._tcase(pr, pc, b, extra=mo, oop=oop) for (pr, pc), b, mo, oop in
             itertools.product(S.PDIMS, [1, 2, 7, 8], S._mem_orders()[::9], (True, False))]
    reps = int(os.environ.get("FD_PROBE_REPS", "3"))
    lines = lines * reps
    for arm, env in (("pool_off", {"CUDECOMP_WORKSPACE_POOL_MIB": "0"}), ("pool_on", {})):
        with tempfile.NamedTemporaryFile("w", suffix="_cases.txt", delete=False) as f:
            f.write("\n".join(lines) + "\n")
        samples, stop = [], threading.Event()

        def sampler():
            while not stop.is_set():
                c = fd_counts("transpose_test_R64")
                if c:
                    samples.append((time.time(), c))
                time.sleep(1.0)

        t = threading.Thread(target=sampler)
        t.start()
        t0 = time.time()
        err = None
        try:
            logs = mp.run_binary_ranks(4, [_binary("transpose_test_R64"), "--testfile", f.name], 900,
                                       dict(env, CUDECOMP_TEST_STOP_AT_FIRST_FAILURE="1", CUDECOMP_TEST_VERDICT_TIMEOUT="60"))
            ok = "Passed all tests." in logs[0]
        except AssertionError as e:
            ok, err = False, str(e)[-1500:]
        stop.set()
        t.join()
        os.unlink(f.name)
        series = [[round(ts - t0, 1), sorted(n for n, _ in c.values())] for ts, c in samples]
        last_kinds = [k for _, k in samples[-1][1].values()] if samples else []
        print(json.dumps({"arm": arm, "cases": len(lines), "passed": ok, "seconds": round(time.time() - t0, 1),
                          "nofile_limit": resource.getrlimit(resource.RLIMIT_NOFILE), "fd_counts_over_time": series[:: max(1, len(series) // 12)],
                          "last_sample_kinds": last_kinds, "error": err}), flush=True)


if __name__ == "__main__":
    main()
