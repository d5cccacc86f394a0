"""Link compatibility of the drop-in boundary: a plain-C program written against cudecomp.h
(examples/c/basic_usage.c, the shape of the reference's examples/cc/basic_usage) is compiled with gcc,
linked against libcudecomp.so and run on 1 and 4 ranks; it checks its own results against the closed form."""
import os
import subprocess

import pytest

from tests.mp import free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "examples", "c")


def build(target):
    subprocess.check_call(["make", "-s", "-C", EX, target])
    return os.path.join(EX, target)


def launch(exe, n, extra_env=None):
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   CUDECOMP_BOOTSTRAP_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "PASSED" in o, "rank %d:\n%s" % (r, o)


def test_c_example_single_rank():
    launch(build("basic_usage"), 1)


def test_c_example_four_ranks_peer_transport():
    # defaults (MPI_P2P / HALO_MPI enums -> xGMI peer transport), ranks sharing the GPU
    launch(build("basic_usage"), 4)
    launch(build("basic_usage"), 2, {"EXAMPLE_TRANSPOSE_BACKEND": "7", "EXAMPLE_HALO_BACKEND": "4"})


def test_c_example_under_mpirun_with_mpi_flavour():
    mpirun = "/opt/conda/bin/mpirun"
    if not os.path.exists(mpirun) or not os.path.exists(os.path.join(ROOT, "cudecomp_amd", "lib", "libcudecomp_mpi.so")):
        pytest.skip("no MPI installation / MPI flavour not built")
    exe = build("basic_usage_mpi")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for n in (1, 4):
        out = subprocess.run([mpirun, "-np", str(n), exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                             timeout=300)
        text = out.stdout.decode()
        assert out.returncode == 0 and text.count("PASSED") == n, text


def test_transposes_inside_sub_communicators_default_build():
    """Two groups of two ranks, each with its own handle on its own MPI sub-communicator, run an X -> Y -> X round trip at
    the same time through the DEFAULT build (control plane over the program's MPI found at run time, data over the
    one-sided transport): tests/native/subcomm_test.c."""
    mpirun = "/opt/conda/bin/mpirun"
    if not os.path.exists(mpirun) or not os.path.exists("/opt/conda/include/mpi.h"):
        pytest.skip("no MPI installation")
    native = os.path.join(ROOT, "tests", "native")
    subprocess.check_call(["make", "-s", "-C", native, "build/subcomm_test"])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SUBCOMM_GROUP="2", SUBCOMM_TRANSPOSE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for attempt in (1, 2):
        out = subprocess.run([mpirun, "-np", "4", os.path.join(native, "build", "subcomm_test")], env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        text = out.stdout.decode()
        # Seen ONCE in eight suite runs of round 6 (gpurun_out/r06_sixth): the runtime refused to export a fresh workspace
        # ("hipIpcGetMemHandle failed: invalid argument") in one of the two groups; three repetitions on another box and every
        # later run passed.  Not what this test is about (sub-communicators): that one platform error gets one more attempt,
        # and says so; anything else fails at once.
        if attempt == 1 and out.returncode != 0 and "hipIpcGetMemHandle failed: invalid argument" in text:
            print("subcomm_test: hipIpcGetMemHandle refused a fresh workspace (platform hiccup, seen before); one more attempt")
            continue
        break
    assert out.returncode == 0 and "PASSED (4 ranks in groups of 2)" in text, text
