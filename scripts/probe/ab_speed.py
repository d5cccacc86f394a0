"""Per-case speed of the native test programs: the round-3 build (scripts/probe/ab_r03, built from commit cc346f8) against
HEAD on the SAME box, alternating, 4 ranks, a slice of the reference's transpose_test_cc case list (all eight backends need
the RCCL stand-in, so only the one-sided enums 1, 2, 3, 6, 7, 8 are taken).  Prints seconds per launch and ms per case."""
import os
import re
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.mp import run_binary_ranks  # noqa: E402
from tests.test_gpu_runner_cases import load_cases  # noqa: E402

lines = [l for l in load_cases()["transpose_test_cc"] if re.search(r"--backend [123678] ", l)][::3][:400]
only = {k: [l for l in lines if "--backend %d " % k in l][:120] for k in (1, 6, 8)}
builds = [("r03", os.path.join(ROOT, "scripts", "probe", "ab_r03", "tests", "native", "build", "transpose_test_R32")),
          ("head", os.path.join(ROOT, "tests", "native", "build", "transpose_test_R32"))]


def run(label, exe, cases, env=None):
    with tempfile.NamedTemporaryFile("w", suffix="_cases.txt", delete=False) as f:
        f.write("\n".join(cases) + "\n")
        path = f.name
    t0 = time.time()
    logs = run_binary_ranks(4, [exe, "--testfile", path], timeout=900, extra_env=env)
    wall = time.time() - t0
    os.unlink(path)
    for l in logs[0].splitlines():
        if l.startswith("Phase times") or "DEBUG destroy timing" in l:
            print("    " + l, flush=True)
    m = re.search(r"Completed all tests, running time ([0-9.]+) s", logs[0])
    ok = logs[0].count(" PASSED") == len(cases)
    print("%-28s %4d cases  wall %6.1f s  in-program %6.1f s  %6.1f ms per case  %s" % (label, len(cases), wall, float(m.group(1)) if m else -1,
          1000 * float(m.group(1)) / len(cases) if m else -1, "ok" if ok else "FAILED"), flush=True)


libs = os.path.join(ROOT, "scripts", "probe", "ab_libs")
T = {"CUDECOMP_DEBUG_DESTROY_TIMING": "1", "CUDECOMP_TEST_PHASE_TIMES": "1"}
big = {"LD_LIBRARY_PATH": os.path.join(libs, "head_big")}
run("HEAD (small device code)", builds[1][1], lines, T)
run("HEAD with the big device code, epoch read through pinned memory", builds[1][1], lines, dict(T, **big))
run("HEAD with the big device code, pageable epoch read", builds[1][1], lines, dict(T, CUDECOMP_DEBUG_PAGEABLE_EPOCH_READ="1", **big))
