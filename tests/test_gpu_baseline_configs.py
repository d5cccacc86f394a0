"""The configurations BASELINE.json names, at their FULL sizes.  The host oracle would take minutes at these sizes,
so the pencils are filled with the global linear index on the device and EVERY cell of every output pencil is compared
on the device with the closed form of that pencil (bit-exact, nothing sampled); the older property checks (global
multiset preserved, round trip exact on random payloads) stay for a second kind of payload.  Ranks share the GPU
(one-sided transport; the RCCL code path at these sizes runs through the stand-in of tests/shim, real RCCL with one
rank in tests/test_gpu_self_exchange.py)."""
import os

from tests.mp import ROOT

SHIM = os.path.join(ROOT, "tests", "shim", "libfake_rccl.so")
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
@pytest.mark.parametrize("backend", [pytest.param(cd.TRANSPOSE_COMM_MPI_P2P, marks=pytest.mark.extended),
                                     pytest.param(cd.TRANSPOSE_COMM_NVSHMEM_PL, marks=pytest.mark.extended), cd.TRANSPOSE_COMM_NVSHMEM_SM],
                         ids=["peer_copy", "peer_pipelined", "peer_put"])
def test_full_size_cycle_properties(name, nranks, args, backend):
    # (a second kind of payload next to test_full_size_every_cell, which runs every transport: one transport in the default run,
    # the other two with CUDECOMP_TEST_EXTENDED=1 -- profiles/r06_gpu_suite_extended.log)
    res = run_ranks(nranks, "tests.gpu_bodies", "cycle_properties", dict(args, transpose_backend=backend), timeout=600)
    assert all(r["round_trip_exact"] for r in res)
    totals = [sum(r["sums"][hop] for r in res) for hop in range(5)]
    assert all(t == totals[0] for t in totals), totals


BACKENDS = [(cd.TRANSPOSE_COMM_MPI_P2P, "torch"), (cd.TRANSPOSE_COMM_NVSHMEM, "torch"),
            (cd.TRANSPOSE_COMM_NVSHMEM_PL, "torch"), (cd.TRANSPOSE_COMM_NVSHMEM_SM, "torch"),
            (cd.TRANSPOSE_COMM_NVSHMEM_SM, "malloc")]


@pytest.mark.parametrize("name,nranks,args", CONFIGS, ids=[c[0] for c in CONFIGS])
@pytest.mark.parametrize("backend,data", BACKENDS, ids=["mpi_p2p", "nvshmem", "nvshmem_pl", "nvshmem_sm", "nvshmem_sm_direct_put"])
def test_full_size_every_cell(name, nranks, args, backend, data):
    a = dict(args, transpose_backend=backend, data_alloc=data)
    for r in run_ranks(nranks, "tests.gpu_bodies", "cycle_exact", a, timeout=600):
        assert r["failures"] == []
        if data == "malloc":
            assert r["counters"]["direct_puts"] > 0 and r["counters"]["direct_puts"] == r["counters"]["peer_fused"]


SCALING_POINTS = [
    # bench.py --gpus 2 / 4: the 1024^3 problem on fewer ranks -- 4- and 2-GiB pencils, 8- and 4-GiB workspaces, all from
    # cudecompMalloc as in the bench (allocation sizes around the 2-GiB IPC limit of the platform)
    ("S_1024cube_f64_2x1", 2, {"gdims": (1024, 1024, 1024), "pdims": (2, 1), "kind": 1, "ac": (1, 1, 1)}),
    ("S_1024cube_f64_1x4", 4, {"gdims": (1024, 1024, 1024), "pdims": (1, 4), "kind": 1, "ac": (1, 1, 1)}),
    ("S_1024cube_f64_2x2", 4, {"gdims": (1024, 1024, 1024), "pdims": (2, 2), "kind": 1}),
]


@pytest.mark.parametrize("name,nranks,args", SCALING_POINTS, ids=[c[0] for c in SCALING_POINTS])
@pytest.mark.parametrize("backend", [cd.TRANSPOSE_COMM_NVSHMEM, cd.TRANSPOSE_COMM_NVSHMEM_SM], ids=["nvshmem", "nvshmem_sm_direct_put"])
def test_bench_scaling_points_every_cell(name, nranks, args, backend):
    a = dict(args, transpose_backend=backend, data_alloc="malloc")
    for r in run_ranks(nranks, "tests.gpu_bodies", "cycle_exact", a, timeout=600):
        assert r["failures"] == []
        if backend == cd.TRANSPOSE_COMM_NVSHMEM_SM:
            assert r["counters"]["direct_puts"] > 0


RCCL_PATH_CONFIGS = [c for c in CONFIGS if c[0].startswith("C2")] + [
    # 8 ranks on a 2x4 grid through the stand-in at 512^3: its messages travel as files under /dev/shm, and the 8 GiB a
    # 1024^3 exchange parks there at once does not fit every test box (the one-sided transports cover 1024^3 above)
    ("C3grid_512cube_f64_2x4", 8, {"gdims": (512, 512, 512), "pdims": (2, 4), "kind": 1})]


@pytest.mark.suite_order(2)  # (the stand-in is preloaded: another rank pool -- after the module's other tests, next to test_gpu_rccl_path)
@pytest.mark.parametrize("name,nranks,args", RCCL_PATH_CONFIGS, ids=[c[0] for c in RCCL_PATH_CONFIGS])
@pytest.mark.parametrize("backend", [cd.TRANSPOSE_COMM_NCCL, cd.TRANSPOSE_COMM_NCCL_PL], ids=["nccl", "nccl_pl"])
def test_full_size_every_cell_rccl_code_path(name, nranks, args, backend):
    """Config 2 names "RCCL a2a": the library's RCCL path (ncclAllToAll / grouped send-recv, pipelined variant) at full
    size, in place as well; the bytes under ncclSend/ncclRecv go through the test stand-in because the ranks share
    one GPU (real RCCL: tests/test_gpu_self_exchange.py)."""
    if not os.path.exists(SHIM):
        pytest.skip("tests/shim/libfake_rccl.so not built")
    for inplace in (False, True):
        a = dict(args, transpose_backend=backend, inplace=inplace)
        for r in run_ranks(nranks, "tests.gpu_bodies", "cycle_exact", a, timeout=900, extra_env={"CUDECOMP_TEST_RCCL_SHIM": SHIM}):
            assert r["failures"] == []
            assert r["counters"]["rccl"] > 0


def test_config3_autotuned_process_grid_at_full_size():
    """C3 says "autotuned pgrid": the autotuner runs at 1024^3 on 8 ranks (all one-sided transports x all process
    grids), every rank ends up with the same selection, and a cycle with it is exact in every cell."""
    res = run_ranks(8, "tests.gpu_bodies", "autotune_full_size", {"gdims": (1024, 1024, 1024), "kind": 1, "ac": (1, 1, 1)},
                    timeout=900)
    picks = [r["picked"] for r in res]
    assert all(p == picks[0] for p in picks), picks
    assert picks[0]["pdims"][0] * picks[0]["pdims"][1] == 8
    for r in res:
        assert r["failures"] == []


@pytest.mark.parametrize("axes,backend,env", [([0, 1, 2], cd.HALO_COMM_MPI, {}),
                                              ([0, 1, 2], cd.HALO_COMM_NVSHMEM, {}),
                                              pytest.param([0, 1, 2], cd.HALO_COMM_NCCL, {"CUDECOMP_FORCE_HALO_OVERLAP": "1"},
                                                           marks=pytest.mark.suite_order(2))],
                         ids=["mpi_xyz", "nvshmem_xyz", "rccl_overlapped_xyz"])
def test_config5_halo_full_size(axes, backend, env):
    # C5: 2048 x 2048 x 1024 fp64, 2x4, halo width 2, periodic: UpdateHalos{X,Y,Z} dims 0,1,2 with pack / exchange /
    # unpack overlapped; EVERY cell of every halo-carrying pencil (4.4 GB per rank) is compared on the device with the
    # closed form (reference: tests/ctest/halo_tests.cc:229-272 compares whole pencils; nothing is sampled here either).
    # The RCCL variant (two send/recv groups on a side stream) runs through the stand-in.
    env = dict(env)
    if backend == cd.HALO_COMM_NCCL:
        if not os.path.exists(SHIM):
            pytest.skip("tests/shim/libfake_rccl.so not built")
        env["CUDECOMP_TEST_RCCL_SHIM"] = SHIM
    args = {"gdims": (2048, 2048, 1024), "pdims": (2, 4), "kind": 1, "halo": (2, 2, 2), "periods": (1, 1, 1),
            "axes": axes, "halo_backend": backend}
    for failures in run_ranks(8, "tests.gpu_bodies", "halo_exact", args, timeout=900, extra_env=env):
        assert failures == []
