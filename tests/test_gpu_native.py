"""Native test programs (tests/native): the reference's transpose_test / halo_test command lines, test-file mode and
output protocol, implemented against the public C API.  The case files written here follow the structure of the
reference's tests/test_config.yaml sweeps (grid 128 x 124 x 132 there; a smaller uneven grid here to bound GPU time):
backends x axis-contiguous flags x gdims_dist x halos x padding x in/out of place, memory orders, autotuning, all four
data types.  A maintainer can equally point the reference's tests/test_runner.py at these binaries."""
import itertools
import os
import subprocess
import tempfile

import pytest

from tests.mp import ROOT, run_binary_ranks

pytestmark = pytest.mark.gpu
NATIVE = os.path.join(ROOT, "tests", "native")
SHIM = os.path.join(ROOT, "tests", "shim", "libfake_rccl.so")


def _binary(name):
    path = os.path.join(NATIVE, "build", name)
    if not os.path.exists(path):
        subprocess.run(["make", "-C", NATIVE, "build/" + name], check=True, capture_output=True)
    return path


def _check_logs(binary, lines, logs):
    out = logs[0]
    ok = out.count(" PASSED") == len(lines) and " FAILED" not in out and "Passed all tests." in out
    if not ok:  # keep every rank's output where gpurun merges it back
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "native_failure_%s_%d.log" % (binary, os.getpid())), "w") as f:
            for r, text in enumerate(logs):
                f.write("===== rank %d =====\n%s\n" % (r, text[-20000:]))
    assert ok, out[-3000:]


def _run(binary, nranks, lines, env=None):
    """Runs the case list through the native test program (reference protocol: every case PASSED, "Passed all tests.").
    Every failure is a failure; the only repetition is the one tests/mp.py IPC_EXPORT_REFUSED describes."""
    _run_side_by_side([(binary, nranks, lines, env)])


def _run_side_by_side(jobs, path_of=None):
    """jobs = [(binary, nranks, lines, env)]: independent case lists, launched together while they fit on the GPU side by side
    (tests/mp.py: run_binary_groups); every list is checked like _run's.  The programs stop at their first failing case
    (CUDECOMP_TEST_STOP_AT_FIRST_FAILURE, tests/native/native_test.h): what follows a failure is ranks out of step, not evidence."""
    from tests.mp import run_binary_groups
    paths, groups = [], []
    for binary, nranks, lines, env in jobs:
        with tempfile.NamedTemporaryFile("w", suffix="_cases.txt", delete=False) as f:
            f.write("\n".join(lines) + "\n")
            paths.append(f.name)
        exe = path_of(binary) if path_of else _binary(binary)
        env = dict(env or {})
        env.setdefault("CUDECOMP_TEST_STOP_AT_FIRST_FAILURE", "1")
        env.setdefault("CUDECOMP_TEST_VERDICT_TIMEOUT", "60")
        groups.append((nranks, [exe, "--testfile", paths[-1]], 900, env))
    try:
        all_logs = run_binary_groups(groups)
    finally:
        for p in paths:
            os.unlink(p)
    for (binary, nranks, lines, env), logs in zip(jobs, all_logs):
        _check_logs(binary, lines, logs)


def _transpose_lines(pdims_list, backends, full):
    lines = []
    acs = [(0, 0, 0), (1, 1, 1), (1, 0, 1)] if full else [(0, 0, 0), (1, 1, 1)]
    extras = [("0 0 0", "0 0 0", "0 0 0", "0 0 0", "0 0 0"),      # gd, hex=hez, hey, pdx=pdz, pdy
              ("1 2 3", "1 1 1", "0 0 0", "0 0 0", "1 0 2"),
              ("0 0 0", "2 1 1", "1 2 1", "1 1 0", "0 0 0")]
    for (pr, pc), b, ac, (gd, hxz, hy, pxz, py), oop in itertools.product(pdims_list, backends, acs, extras, ("", "-o")):
        lines.append("--pr %d --pc %d --gx 32 --gy 30 --gz 34 --backend %d --acx %d --acy %d --acz %d --gd %s --hex %s "
                     "--hey %s --hez %s --pdx %s --pdy %s --pdz %s %s" % (pr, pc, b, ac[0], ac[1], ac[2], gd, hxz, hy, hxz,
                                                                        pxz, py, pxz, oop))
    return lines


@pytest.mark.parametrize("dtype", ["R32", "R64", "C32", "C64"])
def test_native_transpose_single_rank(dtype):
    lines = _transpose_lines([(1, 1)], [4], full=(dtype == "R64"))
    if dtype == "R64":  # memory-order sweep: same order for X and Z pencils, as the reference's runner generates it
        perms = [" ".join(map(str, p)) for p in itertools.permutations((0, 1, 2))]
        lines += ["--pr 1 --pc 1 --gx 20 --gy 18 --gz 22 --backend 4 --mem_order %s %s %s %s" % (x, y, x, o)
                  for x, y in itertools.product(perms, perms) for o in ("", "-o")]
    _run("transpose_test_" + dtype, 1, lines)


def test_native_list_stops_at_its_first_failing_case():
    """CUDECOMP_TEST_STOP_AT_FIRST_FAILURE=1 (what _run_side_by_side sets): the second of four cases asks for a 3 x 1 grid on two
    ranks -- refused by cudecompGridDescCreate on every rank -- and the program ends there, failing, within seconds."""
    import time
    good = _transpose_lines([(2, 1)], [1], full=False)[:3]
    lines = [good[0], good[1].replace("--pr 2", "--pr 3"), good[1], good[2]]
    with tempfile.NamedTemporaryFile("w", suffix="_cases.txt", delete=False) as f:
        f.write("\n".join(lines) + "\n")
    t0 = time.time()
    try:
        with pytest.raises(AssertionError) as e:
            run_binary_ranks(2, [_binary("transpose_test_R64"), "--testfile", f.name], 120,
                             {"CUDECOMP_TEST_STOP_AT_FIRST_FAILURE": "1", "CUDECOMP_TEST_VERDICT_TIMEOUT": "60"})
    finally:
        os.unlink(f.name)
    text = str(e.value)
    assert "Stopping at the first failing case (2 of 4 run)" in text and text.count(" PASSED") == 1, text[-2000:]
    assert time.time() - t0 < 60


@pytest.mark.parametrize("dtype", ["R64", "C32"])
def test_native_transpose_four_ranks(dtype):
    backends = [1, 2, 3, 6, 7, 8] if dtype == "R64" else [1, 8]
    lines = _transpose_lines([(2, 2), (1, 4), (4, 1)], backends, full=False)
    lines += ["--pr 0 --pc 0 --gx 32 --gy 30 --gz 34 --backend 0 --acx 1 --acy 1 --acz 1",      # grid + backend autotuning
              "--pr 2 --pc 2 --gx 32 --gy 30 --gz 34 --backend 0 -o",
              "--pr 2 --pc 2 --rank-order 2 --gx 32 --gy 30 --gz 34 --backend 1 --gd 1 1 1"]
    env = {"CUDECOMP_AUTOTUNE_TRANSPOSE_BACKENDS": "^NCCL,NCCL_PL"}  # RCCL cannot place four ranks on one GPU
    _run("transpose_test_" + dtype, 4, lines, env)


def test_native_transpose_four_ranks_rccl_code_path():
    if not os.path.exists(SHIM):
        pytest.skip("tests/shim/libfake_rccl.so not built")
    lines = _transpose_lines([(2, 2), (1, 4)], [4, 5], full=False)
    _run("transpose_test_R64", 4, lines, {"LD_PRELOAD": SHIM})


def _halo_lines(pdims_list, backends):
    lines = []
    for (pr, pc), b, ax, ac, (h, per, pad) in itertools.product(
            pdims_list, backends, (0, 1, 2), (0, 1),
            [((1, 1, 1), (1, 1, 1), (0, 0, 0)), ((2, 1, 2), (1, 0, 1), (1, 0, 2)), ((1, 2, 0), (0, 0, 0), (0, 1, 0))]):
        lines.append("--pr %d --pc %d --gx 16 --gy 20 --gz 18 --backend %d --ax %d --ac %d --hex %d --hey %d --hez %d "
                     "--hpx %d --hpy %d --hpz %d --pdx %d --pdy %d --pdz %d" % ((pr, pc, b, ax, ac) + h + per + pad))
    return lines


@pytest.mark.parametrize("dtype", ["R32", "R64", "C32", "C64"])
def test_native_halo_single_rank(dtype):
    lines = _halo_lines([(1, 1)], [3])
    if dtype == "R64":
        lines += ["--pr 1 --pc 1 --gx 12 --gy 14 --gz 10 --backend 3 --ax %d --mem_order %d %d %d --hex 1 --hey 2 --hez 1"
                  % ((ax,) + p) for ax in range(3) for p in itertools.permutations((0, 1, 2))]
    _run("halo_test_" + dtype, 1, lines)


def test_native_halo_four_ranks():
    lines = _halo_lines([(2, 2), (1, 4), (4, 1)], [1, 2, 4, 5])
    lines += ["--pr 0 --pc 0 --gx 16 --gy 20 --gz 18 --backend 0 --ax 1 --hex 1 --hey 1 --hez 1"]  # autotuned
    _run("halo_test_R64", 4, lines, {"CUDECOMP_AUTOTUNE_HALO_BACKENDS": "^NCCL"})


def test_native_halo_four_ranks_rccl_code_path():
    if not os.path.exists(SHIM):
        pytest.skip("tests/shim/libfake_rccl.so not built")
    _run("halo_test_R64", 4, _halo_lines([(2, 2), (4, 1)], [3]), {"LD_PRELOAD": SHIM})


def test_native_under_mpirun_without_linking_mpi():
    """`mpirun -np 4 ./transpose_test ...` as the reference's runner launches it: the default (MPI-free) build finds its
    ranks in the PMI environment hydra exports."""
    import shutil
    mpirun = shutil.which("mpirun") or "/opt/conda/bin/mpirun"
    if not os.path.exists(mpirun):
        pytest.skip("no mpirun in this image")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"HSA_ENABLE_IPC_MODE_LEGACY": "0", "CUDECOMP_BOOTSTRAP_PORT": "29733"})
    cmd = [mpirun, "-np", "4", _binary("transpose_test_R64"), "--pr", "2", "--pc", "2", "--gx", "32", "--gy", "30", "--gz", "34",
           "--backend", "1", "--acx", "1", "--acy", "1", "--acz", "1", "--hex", "1", "1", "1", "--hez", "1", "1", "1", "-o"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "PASSED" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]
