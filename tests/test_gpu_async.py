"""The one-sided transports are ordered on the STREAM (device-side epoch flags, csrc/sync.hip), not by the host:
  * a transpose call returns to the host long before the GPU has finished it -- measured at BASELINE config 3's size
    on 8 ranks sharing the GPU (reference behaviour being matched: the NVSHMEM backends enqueue everything on the
    caller's stream, include/internal/comm_routines.h:122-258);
  * a whole X->Y->Z->Y->X cycle can be captured from the caller's stream into ONE hipGraph and replayed, with
    fresh data every replay (the call counter of the exchanges lives in device memory)."""
import pytest

import cudecomp_amd as cd
from tests.mp import run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("backend", [cd.TRANSPOSE_COMM_MPI_P2P, cd.TRANSPOSE_COMM_NVSHMEM, cd.TRANSPOSE_COMM_NVSHMEM_PL,
                                     cd.TRANSPOSE_COMM_NVSHMEM_SM], ids=["mpi_p2p", "nvshmem", "nvshmem_pl", "nvshmem_sm"])
def test_calls_return_before_the_gpu_is_done(backend):
    args = {"gdims": (1024, 1024, 1024), "pdims": (2, 4), "kind": 1, "ac": (1, 1, 1), "transpose_backend": backend,
            "cycles": 1, "burst_cycles": 4, "data_alloc": "malloc"}
    res = run_ranks(8, "tests.gpu_bodies", "cycle_exact", args, timeout=600)
    for r in res:
        assert r["failures"] == []
    # 16 transposes issued back to back: the host is done issuing them long before the device has run them (a
    # host-ordered exchange -- stream sync, barrier, copies, sync, barrier -- would make the two times equal).  The MPI
    # enums and NVSHMEM_SM make the host wait for its peers to ENTER each call (descriptor rendezvous), never for GPU work.
    host = sum(r["burst"]["host_ms"] for r in res)
    total = sum(r["burst"]["total_ms"] for r in res)
    assert host < 0.25 * total, (host, total, [r["burst"] for r in res])


@pytest.mark.parametrize("backend", [cd.TRANSPOSE_COMM_NVSHMEM, cd.TRANSPOSE_COMM_NVSHMEM_PL, cd.TRANSPOSE_COMM_NVSHMEM_SM],
                         ids=["nvshmem", "nvshmem_pl", "nvshmem_sm"])
@pytest.mark.parametrize("n,pdims", [(4, (2, 2)), (4, (1, 4)), (2, (2, 1))])
def test_whole_cycle_in_one_user_graph(backend, n, pdims):
    args = {"gdims": (96, 80, 112), "pdims": pdims, "kind": 1, "ac": (1, 1, 1), "transpose_backend": backend, "replays": 3}
    for r in run_ranks(n, "tests.gpu_bodies", "graph_cycle", args, timeout=300):
        assert r["failures"] == []


@pytest.mark.parametrize("gdims,rotations", [((64, 64, 64), 8), ((64, 64, 48), 0)], ids=["cubic_rotation_kernel", "staged"])
def test_in_place_cycle_of_one_rank_in_one_user_graph(gdims, rotations):
    """Single rank, IN PLACE, captured in the caller's hipGraph and replayed on fresh data: the cubic grid runs the in-place
    rotation kernel (csrc/kernels_rotate.hip: 4 hops x (warm-up + capture); replays launch no library code), the other one the
    staged form (permute into the workspace, copy back); X and Z pencils of every replay checked cell by cell."""
    args = {"gdims": gdims, "pdims": (1, 1), "kind": 1, "ac": (1, 1, 1), "replays": 3, "in_place": True}
    for r in run_ranks(1, "tests.gpu_bodies", "graph_cycle", args, timeout=300):
        assert r["failures"] == []
        assert r["counters"]["rotations"] == rotations, r["counters"]
