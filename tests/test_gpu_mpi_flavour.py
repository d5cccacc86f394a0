"""The MPI flavour of the library (libcudecomp_mpi.so, csrc/bootstrap_mpi.cc) on the reference's own case matrix.

`north_star` names a "ROCm-aware-MPI fallback"; the reference's counterparts are the MPI_P2P / MPI_P2P_PL / MPI_A2A
transposes (include/internal/comm_routines.h:325-413, 585-619) and the MPI / MPI_BLOCKING halo exchanges (:708-762).
The native test programs are built a second time as REAL MPI programs (tests/native `make mpi`: MPI_Init,
MPI_COMM_WORLD handed to cudecompInit, linked against libcudecomp_mpi.so) and run under mpirun on 4 ranks over a
deterministic slice of tests/golden/reference_runner_cases_ngpu4.txt.gz restricted to the MPI backend enums: every
configuration of the matrix (memory orders, halos, padding, gdims_dist, axis-contiguous flags, rank orders), all four
data types on the base configurations.  The image's MPICH is host-only, so the exchange is staged through pinned host
buffers (the default of the flavour); CUDECOMP_MPI_GPU_AWARE=1 hands device pointers to MPI and is exercised only where
the caller says the MPI is GPU-aware (CUDECOMP_TEST_MPI_IS_GPU_AWARE=1)."""
import os
import re
import subprocess
import tempfile

import pytest

from tests.mp import ROOT
from tests.test_gpu_runner_cases import ALL_DTYPES, load_cases

pytestmark = pytest.mark.gpu
NATIVE = os.path.join(ROOT, "tests", "native")
MPIRUN = "/opt/conda/bin/mpirun"

# every how-many-th case of a configuration is run (after the restriction to the MPI enums), per data type
TRANSPOSE_STEP = {"transpose_test_cc": 6, "transpose_test_halo_cc": 24, "transpose_test_padding_cc": 24,
                  "transpose_test_gdimdist_cc": 6, "transpose_test_mix_cc": 54, "transpose_test_ac_cc": 2,
                  "transpose_test_rank_order_cc": 1}
HALO_STEP = {"halo_test_cc": 2, "halo_test_halomix_cc": 18, "halo_test_padding_cc": 6, "halo_test_gdimdist_cc": 2,
             "halo_test_mix_cc": 90, "halo_test_ac_cc": 1, "halo_test_rank_order_cc": 1}


def _need():
    if not os.path.exists(MPIRUN) or not os.path.exists("/opt/conda/include/mpi.h"):
        pytest.skip("no MPI installation")
    if not os.path.exists(os.path.join(ROOT, "cudecomp_amd", "lib", "libcudecomp_mpi.so")):
        pytest.skip("MPI flavour of the library not built")


def _binary(name):
    path = os.path.join(NATIVE, "build_mpi", name)
    if not os.path.exists(path):
        subprocess.run(["make", "-s", "-C", NATIVE, "build_mpi/" + name], check=True, capture_output=True)
    return path


def _backend(line):
    return int(re.search(r"--backend (\d+)", line).group(1))


def _run_mpi(binary, lines, nranks=4, env=None):
    """mpirun -np N <binary> --testfile F, the reference runner's launch line; returns rank 0's output."""
    with tempfile.NamedTemporaryFile("w", suffix="_cases.txt", delete=False) as f:
        f.write("\n".join(lines) + "\n")
        path = f.name
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    try:
        out = subprocess.run([MPIRUN, "-np", str(nranks), _binary(binary), "--testfile", path], env=e, stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, timeout=1200)
    finally:
        os.unlink(path)
    text = out.stdout.decode(errors="replace")
    ok = out.returncode == 0 and text.count(" PASSED") == len(lines) and " FAILED" not in text and "Passed all tests." in text
    if not ok:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "mpi_flavour_failure_%s_%d.log" % (binary, os.getpid())), "w") as f:
            f.write(text[-60000:])
    assert ok, text[-3000:]
    return text


def _dtypes(config):
    return ALL_DTYPES if config in ("transpose_test_cc", "halo_test_cc") else ["R32"]


def _run_mpi_jobs(jobs):
    """jobs = [(binary, lines)]: two mpirun launches of four ranks at a time (eight processes on the GPU, the programs are
    host-bound); returns the outputs in order."""
    from concurrent.futures import ThreadPoolExecutor
    from tests import mp
    mp.pool_stop()
    with ThreadPoolExecutor(max_workers=2) as ex:
        return list(ex.map(lambda j: _run_mpi(*j), jobs))


def _transposes(thin):
    configs = load_cases()
    jobs, labels = [], []
    for config, step in sorted(TRANSPOSE_STEP.items()):
        lines = [l for l in configs[config] if _backend(l) in (1, 2, 3)]
        for k, dtype in enumerate(_dtypes(config)):
            jobs.append(("transpose_test_" + dtype, lines[k % step::step][::thin]))
            labels.append((config, dtype))
    total = through_mpi = 0
    for (config, dtype), (binary, pick), text in zip(labels, jobs, _run_mpi_jobs(jobs)):
        n = int(re.search(r"MPI-path transposes: (\d+)", text).group(1))
        assert n > 0, "%s %s: no transpose took the MPI path" % (config, dtype)
        total += len(pick)
        through_mpi += n
    assert through_mpi >= 4 * total  # 4 ranks x (at least 2 exchanging hops of 4) x ... per case, summed over ranks
    return total


def _halos(thin):
    configs = load_cases()
    jobs = []
    for config, step in sorted(HALO_STEP.items()):
        lines = [l for l in configs[config] if _backend(l) in (1, 2)]
        for k, dtype in enumerate(_dtypes(config)):
            jobs.append(("halo_test_" + dtype, lines[k % step::step][::thin]))
    _run_mpi_jobs(jobs)
    return sum(len(j[1]) for j in jobs)


def test_mpi_flavour_transposes_over_the_reference_matrix():
    """MPI_P2P (1), MPI_P2P_PL (2), MPI_A2A (3), every case through csrc/bootstrap_mpi.cc:mpiAlltoall: every second case of the
    slice in the default run (>= 250), the whole slice (>= 500) in the extended one."""
    _need()
    assert _transposes(2) >= 250


def test_mpi_flavour_halos_over_the_reference_matrix():
    """HALO_COMM_MPI (1) and HALO_COMM_MPI_BLOCKING (2) through csrc/bootstrap_mpi.cc:mpiHaloExchange."""
    _need()
    assert _halos(2) >= 125


@pytest.mark.extended
def test_mpi_flavour_whole_slices():
    _need()
    assert _transposes(1) >= 500
    assert _halos(1) >= 250


def test_mpi_flavour_gpu_aware_path():
    """Device pointers handed straight to MPI (CUDECOMP_MPI_GPU_AWARE=1): only meaningful with a GPU-aware MPI."""
    _need()
    if os.environ.get("CUDECOMP_TEST_MPI_IS_GPU_AWARE") != "1":
        pytest.skip("the MPI of this image is host-only (set CUDECOMP_TEST_MPI_IS_GPU_AWARE=1 on a box with a ROCm-aware MPI)")
    configs = load_cases()
    lines = [l for l in configs["transpose_test_cc"] if _backend(l) in (1, 3)][::12]
    _run_mpi("transpose_test_R64", lines, env={"CUDECOMP_MPI_GPU_AWARE": "1"})
    hl = [l for l in configs["halo_test_cc"] if _backend(l) in (1, 2)][::4]
    _run_mpi("halo_test_R64", hl, env={"CUDECOMP_MPI_GPU_AWARE": "1"})
