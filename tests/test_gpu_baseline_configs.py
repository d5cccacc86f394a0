"""The configurations BASELINE.json names, at their FULL sizes, through size-independent properties (the
oracle would take minutes at these sizes): every hop preserves the global multiset of values and the
X->Y->Z->Y->X round trip reproduces every rank's input bit for bit.  Ranks share the GPU (xGMI peer
transport); the RCCL variants of the same plans are covered at small sizes by the CPU plan tests."""
import pytest

import cudecomp_amd as cd
from tests import gpu_bodies as B
from tests.mp import run_ranks

pytestmark = pytest.mark.gpu

CONFIGS = [
    ("C1_256cube_f32_2x1", 2, {"gdims": (256, 256, 256), "pdims": (2, 1), "kind": 0}),
    ("C1_256cube_f32_1x2", 2, {"gdims": (256, 256, 256), "pdims": (1, 2), "kind": 0}),
    ("C2_512cube_f64_2x1", 2, {"gdims": (512, 512, 512), "pdims": (2, 1), "kind": 1}),
    ("C2_512cube_f64_1x2", 2, {"gdims": (512, 512, 512), "pdims": (1, 2), "kind": 1}),
    ("C3_1024cube_f64_2x4", 8, {"gdims": (1024, 1024, 1024), "pdims": (2, 4), "kind": 1}),
    ("C3_1024cube_f64_1x8_contiguous", 8, {"gdims": (1024, 1024, 1024), "pdims": (1, 8), "kind": 1, "ac": (1, 1, 1)}),
    ("C4_1024cube_c64_2x4", 8, {"gdims": (1024, 1024, 1024), "pdims": (2, 4), "kind": 2, "ac": (1, 1, 1)}),
]


@pytest.mark.parametrize("name,nranks,args", CONFIGS, ids=[c[0] for c in CONFIGS])
@pytest.mark.parametrize("backend", [cd.TRANSPOSE_COMM_MPI_P2P, cd.TRANSPOSE_COMM_NVSHMEM_PL, cd.TRANSPOSE_COMM_NVSHMEM_SM],
                         ids=["peer_copy", "peer_pipelined", "peer_put"])
def test_full_size_cycle_properties(name, nranks, args, backend):
    res = run_ranks(nranks, "tests.gpu_bodies", "cycle_properties", dict(args, transpose_backend=backend), timeout=600)
    assert all(r["round_trip_exact"] for r in res)
    totals = [sum(r["sums"][hop] for r in res) for hop in range(5)]
    assert all(t == totals[0] for t in totals), totals


def test_config5_halo_full_size():
    # C5: 2048 x 2048 x 1024 fp64, 2x4, halo width 2, periodic: UpdateHalos{X} dims 0,1,2; verified on a strided
    # sample of halo cells against the closed form (interior initialised with the global linear index)
    args = {"gdims": (2048, 2048, 1024), "pdims": (2, 4), "kind": 1, "halo": (2, 2, 2), "periods": (1, 1, 1),
            "axes": [0], "halo_backend": cd.HALO_COMM_MPI, "sample": 200003}
    for failures in run_ranks(8, "tests.gpu_bodies", "halo_sampled", args, timeout=900):
        assert failures == []
