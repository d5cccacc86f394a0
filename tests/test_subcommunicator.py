"""cudecompInit on a SUB-communicator with the default (MPI-free) build of the library: an ordinary MPICH program
(tests/native/subcomm_test.c, real <mpi.h>, mpirun) splits the world into groups, every group creates its own handle on
its own communicator and must see exactly its group -- the library finds the program's MPI at run time
(csrc/bootstrap_dynmpi.cc; reference behaviour: src/cudecomp.cc:903-1008, tests/ctest/mpi_test_utils.cc:56-66).
CPU only: geometry queries need no GPU (the data path of the same program runs in tests/test_gpu_c_example.py)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MPIRUN = "/opt/conda/bin/mpirun"
NATIVE = os.path.join(ROOT, "tests", "native")


def build():
    if not os.path.exists(MPIRUN) or not os.path.exists("/opt/conda/include/mpi.h"):
        pytest.skip("no MPI installation on this machine")
    subprocess.check_call(["make", "-s", "-C", NATIVE, "build/subcomm_test"])
    return os.path.join(NATIVE, "build", "subcomm_test")


def run(exe, n, env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    out = subprocess.run([MPIRUN, "-np", str(n), exe], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    return out.returncode, out.stdout.decode()


@pytest.mark.parametrize("n,group", [(4, 2), (4, 4), (2, 1), (6, 2)])
def test_handles_on_sub_communicators_see_their_group_only(n, group):
    rc, text = run(build(), n, {"SUBCOMM_GROUP": str(group)})
    assert rc == 0 and "PASSED (%d ranks in groups of %d)" % (n, group) in text, text


def test_without_mpi_discovery_the_communicator_is_a_token_for_the_world():
    # the launcher-environment bootstrap (what a torchrun / ctypes harness gets) cannot express sub-communicators: with the
    # discovery switched off the same program sees the world and the group-sized grid is refused
    from tests.mp import free_port   # (a port of its own: the library's default port may be in use by something else on the machine)
    rc, text = run(build(), 4, {"SUBCOMM_GROUP": "2", "CUDECOMP_DISABLE_MPI_DISCOVERY": "1", "CUDECOMP_BOOTSTRAP_PORT": str(free_port())})
    assert rc != 0 and "product of pdims values must equal number of ranks" in text, text
