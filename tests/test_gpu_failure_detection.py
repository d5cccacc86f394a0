"""A peer that never enters a collective call must cost seconds, not a hung stream: the host rendezvous of the MPI
enums and the device-side flag waits of the NVSHMEM enums both give up after CUDECOMP_PEER_TIMEOUT and surface as
CUDECOMP_RESULT_NVSHMEM_ERROR (8) -- the reference has no counterpart (an absent NVSHMEM / MPI peer hangs it)."""
import pytest

import cudecomp_amd as cd
from tests.mp import run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("backend", [cd.TRANSPOSE_COMM_MPI_P2P, cd.TRANSPOSE_COMM_NVSHMEM, cd.TRANSPOSE_COMM_NVSHMEM_SM],
                         ids=["mpi_p2p_host_rendezvous", "nvshmem_device_wait", "nvshmem_sm"])
def test_absent_peer_is_reported_not_waited_for_forever(backend):
    args = {"gdims": (32, 24, 40), "pdims": (2, 1), "kind": 1, "transpose_backend": backend, "absent_for": 5.0}
    res = run_ranks(2, "tests.gpu_bodies", "absent_peer", args, timeout=120, extra_env={"CUDECOMP_PEER_TIMEOUT": "1.5"})
    r0 = [r for r in res if r["rank"] == 0][0]
    assert r0["error_code"] == cd.RESULT_NVSHMEM_ERROR, r0
    assert 1.0 < r0["seconds"] < 4.5, r0  # PEER_TIMEOUT = 1.5 s; the absent rank stays away for 5 s
