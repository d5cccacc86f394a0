"""Round 5: the deterministic failure of the reference's 8-rank transpose_test_cc list (R64) at HEAD -- which ingredient?
Arms (each = the whole 1728-case list through tests/native/build/transpose_test_R64 on 8 ranks sharing the GPU):
    head_gate_on      library at HEAD, input-integrity gate on (the failing combination)
    head_gate_off     CUDECOMP_TEST_INPUT_GATE=0
    presplit_gate_on  the library as it was before the kernels were split over several code objects (scripts/probe/ab_state)
    head_reuse        gate on, CUDECOMP_TEST_REUSE_BUFFERS=1 (no hipMalloc / hipFree of the data buffers per case)
    head_nopool       gate on, CUDECOMP_WORKSPACE_POOL_MIB=0
One line per arm; the per-rank excerpts of failing arms are kept under gpurun_out/r05_repro/."""
import glob
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.mp import run_binary_ranks  # noqa: E402
from tests.test_gpu_runner_cases import load_cases  # noqa: E402
from tests.test_gpu_native import SHIM  # noqa: E402

lines = load_cases(8)["transpose_test_cc"]
if len(sys.argv) > 2:
    lines = [l for l in lines if sys.argv[2] in l]
exe = os.path.join(ROOT, "tests", "native", "build", "transpose_test_R64")
out = os.path.join(ROOT, "gpurun_out", "r05_repro")
os.makedirs(out, exist_ok=True)
arms = [("head_gate_on", {}), ("head_gate_off", {"CUDECOMP_TEST_INPUT_GATE": "0"}),
        ("presplit_gate_on", {"LD_LIBRARY_PATH": os.path.join(ROOT, "scripts", "probe", "ab_state")}),
        ("head_reuse", {"CUDECOMP_TEST_REUSE_BUFFERS": "1"}), ("head_nopool", {"CUDECOMP_WORKSPACE_POOL_MIB": "0"}),
        # library builds of earlier commits of the round (scripts/probe/old_libs/<sha>, built from git worktrees)
        ("lib_e73bfd2", {"LD_LIBRARY_PATH": os.path.join(ROOT, "scripts", "probe", "old_libs", "e73bfd2")}),
        ("lib_5e4db36", {"LD_LIBRARY_PATH": os.path.join(ROOT, "scripts", "probe", "old_libs", "5e4db36")}),
        ("head_census", {"CUDECOMP_QUEUE_CENSUS": "1"})]
only = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] != "all" else None
todo = arms if not only else [(n, dict(arms)[n]) for n in only]  # (a name may be given several times)
for rep, (name, env) in enumerate(todo):
    with tempfile.NamedTemporaryFile("w", suffix="_cases.txt", delete=False) as f:
        f.write("\n".join(lines) + "\n")
        path = f.name
    e = dict(env)
    if os.path.exists(SHIM):
        e["LD_PRELOAD"] = SHIM
    t0 = time.time()
    try:
        logs = run_binary_ranks(8, [exe, "--testfile", path], timeout=900, extra_env=e)
        ok = logs[0].count(" PASSED") == len(lines) and " FAILED" not in logs[0]
        print("%-18s %d cases: %s in %.1f s" % (name, len(lines), "passed" if ok else "FAILED (rank 0 verdicts)", time.time() - t0), flush=True)
    except AssertionError as ex:
        msg = str(ex)
        print("%-18s %d cases: FAILED in %.1f s: %d 'differ' lines, %d gate trips in the tails" % (name, len(lines), time.time() - t0,
              msg.count("cells differ"), msg.count("input stale")), flush=True)
        for fpath in glob.glob(os.path.join(ROOT, "gpurun_out", "ranks_failure_transpose_test_R64_%d.log" % os.getpid())):
            shutil.move(fpath, os.path.join(out, "ranks_failure_%s_%d.log" % (name, rep)))
    os.unlink(path)
