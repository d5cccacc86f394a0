"""Per-case speed of the native test programs with the default library (0.52 MB of device code) against the
`make TUNING_VARIANTS=1` library (cudecomp_amd/lib_tuning, 0.72 MB) on the SAME box, alternating: the A/B in which round 4
saw the "device code size cliff" (profiles/r04_tuning.md: 8 vs 42 ms per case).  4 ranks, a slice of the reference's
transpose_test_cc case list (one-sided enums 1, 2, 3, 6, 7, 8).  Prints seconds per launch and ms per case."""
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
exe = os.path.join(ROOT, "tests", "native", "build", "transpose_test_R32")


def run(label, cases, env=None):
    with tempfile.NamedTemporaryFile("w", suffix="_cases.txt", delete=False) as f:
        f.write("\n".join(cases) + "\n")
        path = f.name
    t0 = time.time()
    logs = run_binary_ranks(4, [exe, "--testfile", path], timeout=900, extra_env=env)
    wall = time.time() - t0
    os.unlink(path)
    for l in logs[0].splitlines():
        if l.startswith("Phase times"):
            print("    " + l, flush=True)
    m = re.search(r"Completed all tests, running time ([0-9.]+) s", logs[0])
    ok = logs[0].count(" PASSED") == len(cases)
    print("%-44s %4d cases  wall %6.1f s  in-program %6.1f s  %6.1f ms per case  %s" % (label, len(cases), wall, float(m.group(1)) if m else -1,
          1000 * float(m.group(1)) / len(cases) if m else -1, "ok" if ok else "FAILED"), flush=True)


T = {"CUDECOMP_TEST_PHASE_TIMES": "1"}
big = {"LD_LIBRARY_PATH": os.path.join(ROOT, "cudecomp_amd", "lib_tuning")}
for rep in range(2):
    run("default library (0.52 MB device code)", lines, T)
    run("TUNING_VARIANTS library (0.72 MB device code)", lines, dict(T, **big))
