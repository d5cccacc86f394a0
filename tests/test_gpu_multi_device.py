"""One rank per GPU, nothing shimmed: real `librccl` between devices, cross-device IPC mappings, remote stores and SDMA
copies over xGMI.  Every test here SKIPS unless the box shows at least as many GPUs as the case has ranks (the pool this
was developed on has one GPU per box), so the first multi-GPU box produces evidence in one pytest run:

    python -m pytest tests/test_gpu_multi_device.py -m gpu -q

What they cover that the shared-GPU tests cannot: ncclCommInitRank with several ranks, peer translation through
ci.global_ranks (reference include/internal/comm_routines.h:296-322), the pipelined per-peer flow under real asynchrony
(:533-584), the halo pair exchange and its two-rank periodic ordering rule (:686-707), communicator creation inside the
autotuner (reference src/cudecomp.cc:59-72, src/autotune.cc:275-291), hipIpcOpenMemHandle / hipDeviceEnablePeerAccess
across devices, write-through stores into a remote HBM, the start-up link probe.  Every cell of every pencil is
compared with its closed form, as in tests/test_gpu_baseline_configs.py."""
import os

import pytest

import cudecomp_amd as cd
from tests import cases as K
from tests.mp import gpus_on_this_host, run_ranks
from tests.test_gpu_baseline_configs import CONFIGS
from tests.test_gpu_native import _halo_lines, _run as _run_native, _transpose_lines

pytestmark = pytest.mark.gpu


def _need(nranks):
    have = gpus_on_this_host()  # (from the kernel driver's topology: the pytest process itself never opens the GPU)
    if have < nranks:
        pytest.skip("needs %d GPUs (one rank per device, real RCCL / cross-device IPC); this box has %d" % (nranks, have))


# one hardware queue per copy stream is only needed with the pipelined one-sided transport's parked waits; harmless else
ENV = {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}

REAL = [c for c in CONFIGS if c[0].startswith(("C1", "C2", "C3"))]


@pytest.mark.parametrize("name,nranks,args", REAL, ids=[c[0] for c in REAL])
@pytest.mark.parametrize("backend", [cd.TRANSPOSE_COMM_NCCL, cd.TRANSPOSE_COMM_NCCL_PL], ids=["nccl", "nccl_pl"])
def test_real_rccl_every_cell(name, nranks, args, backend):
    """Configs 1-3 over REAL RCCL, one rank per GPU, out of place and in place (config 2 names "RCCL a2a")."""
    _need(nranks)
    for inplace in (False, True):
        a = dict(args, transpose_backend=backend, inplace=inplace)
        for r in run_ranks(nranks, "tests.gpu_bodies", "cycle_exact", a, timeout=900, extra_env=ENV):
            assert r["failures"] == []
            assert r["counters"]["rccl"] > 0


ONE_SIDED = [(cd.TRANSPOSE_COMM_MPI_P2P, "torch"), (cd.TRANSPOSE_COMM_MPI_P2P_PL, "torch"), (cd.TRANSPOSE_COMM_NVSHMEM, "torch"),
             (cd.TRANSPOSE_COMM_NVSHMEM_PL, "torch"), (cd.TRANSPOSE_COMM_NVSHMEM_SM, "torch"),
             (cd.TRANSPOSE_COMM_NVSHMEM_SM, "malloc")]


@pytest.mark.parametrize("name,nranks,args", REAL, ids=[c[0] for c in REAL])
@pytest.mark.parametrize("backend,data", ONE_SIDED,
                         ids=["mpi_p2p", "mpi_p2p_pl", "nvshmem", "nvshmem_pl", "nvshmem_sm", "nvshmem_sm_direct_put"])
@pytest.mark.parametrize("engine", ["sdma", "cu"])
def test_one_sided_transport_across_devices_every_cell(name, nranks, args, backend, data, engine):
    """The xGMI peer transport between DIFFERENT devices: IPC mappings of another GPU's memory, copy engines vs the
    library's copy kernel, kernel-driven remote stores (NVSHMEM_SM) and the direct put."""
    _need(nranks)
    a = dict(args, transpose_backend=backend, data_alloc=data)
    env = dict(ENV, CUDECOMP_PEER_COPY_ENGINE=engine)
    for r in run_ranks(nranks, "tests.gpu_bodies", "cycle_exact", a, timeout=900, extra_env=env):
        assert r["failures"] == []
        if data == "malloc":
            assert r["counters"]["direct_puts"] > 0


@pytest.mark.parametrize("n", [2, 4, 8])
def test_real_rccl_small_grids_all_layouts(n):
    """The multi-rank sweep of tests/test_gpu_rccl_path.py without the stand-in: grouped a2a(v) with uneven chunks,
    ncclAllToAll on slab grids, the pipelined variant, all four element sizes."""
    _need(n)
    jobs = []
    for pdims in [(2, 1), (1, 2), (2, 2), (1, 4), (4, 1), (2, 4), (4, 2), (1, 8), (8, 1)]:
        if pdims[0] * pdims[1] != n:
            continue
        for ac, backend, kind, gdims in ((K.DEFAULT_AC, cd.TRANSPOSE_COMM_NCCL, 1, (32, 24, 40)),
                                         (K.ALL_AC, cd.TRANSPOSE_COMM_NCCL, 0, (32, 24, 40)),
                                         (K.ALL_AC, cd.TRANSPOSE_COMM_NCCL_PL, 3, (32, 24, 40)),
                                         (K.DEFAULT_AC, cd.TRANSPOSE_COMM_NCCL_PL, 1, (31, 25, 38)),
                                         (K.ALL_AC, cd.TRANSPOSE_COMM_NCCL, 2, (29, 23, 37))):
            jobs.append({"fn": "transpose_chain", "id": "P%dx%d_%s_b%d_k%d" % (pdims[0], pdims[1], ac, backend, kind),
                         "args": {"gdims": gdims, "pdims": pdims, "ac": ac, "kind": kind, "transpose_backend": backend}})
    for failures in run_ranks(n, "tests.gpu_bodies", "many", {"jobs": jobs}, timeout=600, extra_env=ENV):
        assert failures == []


@pytest.mark.parametrize("backend", [cd.HALO_COMM_NCCL, cd.HALO_COMM_MPI, cd.HALO_COMM_NVSHMEM], ids=["rccl", "mpi", "nvshmem"])
@pytest.mark.parametrize("overlap", [False, True], ids=["one_group", "overlapped"])
def test_config5_halo_full_size_across_devices(backend, overlap):
    """Config 5 (2048 x 2048 x 1024 fp64, 2x4, halo 2, periodic) with real neighbours: RCCL send/recv pairs (one group, and
    the two overlapped direction groups on the side stream) and the one-sided halo exchange, all three pencils."""
    _need(8)
    env = dict(ENV)
    env["CUDECOMP_FORCE_HALO_OVERLAP" if overlap else "CUDECOMP_DISABLE_HALO_OVERLAP"] = "1"
    args = {"gdims": (2048, 2048, 1024), "pdims": (2, 4), "kind": 1, "halo": (2, 2, 2), "periods": (1, 1, 1),
            "axes": [0, 1, 2], "halo_backend": backend}
    for failures in run_ranks(8, "tests.gpu_bodies", "halo_exact", args, timeout=900, extra_env=env):
        assert failures == []


def test_two_rank_periodic_halo_ordering_real_rccl():
    """Two ranks along a periodic dim: both neighbours are the SAME peer, RCCL pairs sends and receives in issue order
    (transport.cc haloExchange: the high face goes first).  Small grid, whole-pencil compare against the oracle."""
    _need(2)
    lines = _halo_lines([(2, 1), (1, 2)], [3])
    _run_native("halo_test_R64", 2, lines, ENV)


@pytest.mark.parametrize("n", [2, 4, 8])
def test_autotune_all_backends_across_devices(n):
    """Grid + backend autotuning with EVERY backend competing (RCCL communicator created inside the sweep, one-sided
    transports over real links), then a checked cycle with the winner; every rank must report the same selection."""
    _need(n)
    res = run_ranks(n, "tests.gpu_bodies", "autotune_full_size", {"gdims": (512, 512, 512), "kind": 1, "ac": (1, 1, 1)},
                    timeout=900, extra_env=ENV)
    picks = [r["picked"] for r in res]
    assert all(p == picks[0] for p in picks), picks
    assert picks[0]["pdims"][0] * picks[0]["pdims"][1] == n
    for r in res:
        assert r["failures"] == []


@pytest.mark.parametrize("n", [2, 4, 8])
def test_native_programs_across_devices(n):
    """The reference-protocol test programs (tests/native) on one rank per GPU: all eight transpose backends and all
    five halo backends, halos / padding / gdims_dist, in and out of place."""
    _need(n)
    grids = {2: [(2, 1), (1, 2)], 4: [(2, 2), (1, 4), (4, 1)], 8: [(2, 4), (4, 2), (1, 8), (8, 1)]}[n]
    _run_native("transpose_test_R64", n, _transpose_lines(grids, [1, 2, 3, 4, 5, 6, 7, 8], full=False), ENV)
    _run_native("halo_test_R64", n, _halo_lines(grids, [1, 2, 3, 4, 5]), ENV)


def test_link_probe_reports_cross_device_rates():
    """The start-up link probe (transport.cc peerMeasureLink) must see different devices and measure both engines."""
    _need(2)
    res = run_ranks(2, "tests.gpu_bodies", "link_info", {}, timeout=300, extra_env=ENV)
    for r in res:
        assert r["measured"] and r["crosses_devices"]
        assert r["gbps_sdma"] > 1.0 and r["gbps_cu"] > 1.0
