"""Repeats the 8-rank transposes of tests/test_gpu_native_sweep.py::test_sweep_eight_ranks (one backend at a time, or the
test's own mix) to catch rare failures:
    python scripts/probe/stress_eight_ranks.py <backend|mix> <iterations> [ENV=VALUE ...]
Prints, per iteration that fails, the failing cases and every "CUDECOMP:VERIFY" / "stale mapping" line of any rank; at the
end the failure count and how many stale IPC mappings the library detected (CUDECOMP_VERBOSE=1 is set for the ranks)."""
import itertools
import os
import re
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.mp import run_binary_ranks  # noqa: E402
from tests.test_gpu_native import _binary  # noqa: E402
from tests.test_gpu_native_sweep import _mem_orders, _tcase  # noqa: E402

if os.environ.get("STRESS_PARENT_CONTEXT"):
    # a NINTH process with a live GPU context, as the pytest process of the suite has once it ran a test in-process
    import torch
    _keep = torch.zeros(1 << 20, device="cuda")
    _keep += 1
    torch.cuda.synchronize()
backend, iters = (0 if sys.argv[1] == "mix" else int(sys.argv[1])), int(sys.argv[2])
env = dict(a.split("=", 1) for a in sys.argv[3:])
env.setdefault("CUDECOMP_VERBOSE", "1")
pd8 = [(1, 8), (2, 4), (4, 2), (8, 1)]
lines = [_tcase(pr, pc, backend, extra=mo, oop=oop) for (pr, pc), mo, oop in
         itertools.product(pd8, _mem_orders()[::6], (True, False))]
if backend == 0:  # the test's own list: backends 1, 2, 8 alternating
    lines = [_tcase(pr, pc, b, extra=mo, oop=oop) for (pr, pc), b, mo, oop in
             itertools.product(pd8, [1, 2, 8], _mem_orders()[::12], (True, False))]
bad = stale = 0
t0 = time.time()
for it in range(iters):
    with tempfile.NamedTemporaryFile("w", suffix="_cases.txt", delete=False) as f:
        f.write("\n".join(lines) + "\n")
        path = f.name
    text, failed = "", False
    try:
        logs = run_binary_ranks(8, [_binary("transpose_test_R64"), "--testfile", path], timeout=900, extra_env=dict(env))
        text = "\n".join(logs)
        failed = not (logs[0].count(" PASSED") == len(lines) and " FAILED" not in logs[0] and "Passed all tests." in logs[0])
    except AssertionError as e:
        text, failed = str(e), True
    finally:
        os.unlink(path)
    stale += len(re.findall(r"stale mapping", text))
    if failed:
        bad += 1
        keep = [l.strip()[-260:] for l in text.splitlines() if ("--pr" in l and "transpose_test" in l and "command:" not in l)
                or "differ" in l or "CUDECOMP:VERIFY" in l or "CUDECOMP:ERROR" in l]
        print("iteration %d FAILED:\n  %s" % (it, "\n  ".join(keep[:40])), flush=True)
    else:
        for l in text.splitlines():
            if "CUDECOMP:VERIFY" in l:
                print("iteration %d (passed) %s" % (it, l.strip()[-260:]), flush=True)
print("parent holds a GPU context: %s; backend %d env %s: %d of %d iterations failed (%d cases each, %.0f s); stale IPC mappings detected and replaced: %d"
      % (bool(os.environ.get("STRESS_PARENT_CONTEXT")), backend, env, bad, iters, len(lines), time.time() - t0, stale), flush=True)
