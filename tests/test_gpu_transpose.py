"""HIP transposes through the C ABI vs the oracle: the reference's ctest matrix (single rank in-process,
2x2 / 3x1 / 1x4 as processes sharing the GPU over the xGMI peer transport), the legacy 36-memory-order
sweep, and size-independent properties at the benchmark size."""
import itertools

import numpy as np
import pytest
import torch

import cudecomp_amd as cd
from tests import cases as K
from tests import gpu_bodies as B
from tests.mp import run_ranks

pytestmark = pytest.mark.gpu

SINGLE = [c for c in K.ctest_transpose_cases(pdims_list=((1, 1),)) if tuple(c["pdims"]) == (1, 1)]


@pytest.mark.parametrize("c", SINGLE, ids=K.case_id)
def test_ctest_cases_single_rank(c):
    assert B.single_transpose(0, 1, c) == []


@pytest.mark.parametrize("mo", K.mem_order_combos(), ids=lambda m: "".join("".join(map(str, r)) for r in m))
def test_all_mem_orders_single_rank(mo):
    # legacy sweep shape scaled down; fp64 keeps 16-byte vector paths reachable (even extents) and the
    # 9x10x11 run hits the scalar paths
    for gdims, kind in (((16, 12, 20), 1), ((9, 10, 11), 0)):
        args = {"gdims": gdims, "pdims": (1, 1), "mem_order": mo, "kind": kind}
        assert B.transpose_chain(0, 1, args) == []


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_legacy_grid_single_rank(kind):
    # tests/test_config.yaml:23-25: 128 x 124 x 132, default and axis-contiguous layouts, halos + padding
    for ac in (K.DEFAULT_AC, K.ALL_AC):
        args = {"gdims": (128, 124, 132), "pdims": (1, 1), "ac": ac, "kind": kind}
        assert B.transpose_chain(0, 1, args) == []
    args = {"gdims": (128, 124, 132), "pdims": (1, 1), "ac": K.ALL_AC, "kind": kind,
            "halos": [(1, 1, 1), (0, 0, 0), (1, 1, 1)], "pads": [(0, 0, 0), (1, 1, 1), (0, 0, 0)]}
    assert B.transpose_chain(0, 1, args) == []


MULTI = [c for c in K.ctest_transpose_cases(pdims_list=((2, 2),)) if tuple(c["pdims"]) != (1, 1)]
# a spread of the multi-rank matrix (every scenario, one dtype/op each) keeps GPU minutes bounded
MULTI_PICK = {}
for c in MULTI:
    MULTI_PICK.setdefault((c["name"], c["op"], c["out_of_place"]), c)
MULTI_PICK = [c for i, c in enumerate(MULTI_PICK.values()) if c["name"] != "BaselineDefaultLayout" or i % 2 == 0]


def _by_nranks(cases):
    groups = {}
    for c in cases:
        groups.setdefault(c["pdims"][0] * c["pdims"][1], []).append(c)
    return groups


@pytest.mark.parametrize("backend", [cd.TRANSPOSE_COMM_MPI_P2P, cd.TRANSPOSE_COMM_NVSHMEM_PL],
                         ids=["peer", "peer_pipelined"])
def test_ctest_cases_multi_rank_peer_transport(backend):
    # one process group per rank count runs the whole matrix (a group launch costs ~10 s of imports)
    for n, cases in sorted(_by_nranks(MULTI_PICK).items()):
        jobs = [{"fn": "single_transpose", "id": K.case_id(c), "args": dict(c, transpose_backend=backend)}
                for c in cases]
        for failures in run_ranks(n, "tests.gpu_bodies", "many", {"jobs": jobs}, timeout=600):
            assert failures == []


@pytest.mark.parametrize("n", [2, 3, 4])
def test_cycle_multi_rank_peer_transport(n):
    jobs = []
    for pdims in [(2, 2), (1, 4), (4, 1), (2, 1), (1, 2), (3, 1), (1, 3)]:
        if pdims[0] * pdims[1] != n:
            continue
        for ac, work, backend in ((K.DEFAULT_AC, "malloc", cd.TRANSPOSE_COMM_MPI_P2P),
                                  (K.ALL_AC, "torch", cd.TRANSPOSE_COMM_MPI_A2A),
                                  (K.ALL_AC, "malloc", cd.TRANSPOSE_COMM_NVSHMEM),
                                  (K.ALL_AC, "malloc", cd.TRANSPOSE_COMM_NVSHMEM_SM),
                                  (K.DEFAULT_AC, "torch", cd.TRANSPOSE_COMM_NVSHMEM_SM)):
            jobs.append({"fn": "transpose_chain", "id": "P%dx%d_%s_%s_b%d" % (pdims[0], pdims[1], ac, work, backend),
                         "args": {"gdims": (32, 24, 40), "pdims": pdims, "ac": ac, "kind": 1, "work_alloc": work,
                                  "transpose_backend": backend}})
        # pencils from cudecompMalloc: NVSHMEM_SM packs straight into the peers' OUTPUT pencils (no unpack pass)
        for ac, gdims in ((K.ALL_AC, (32, 24, 40)), (K.DEFAULT_AC, (31, 25, 38))):
            jobs.append({"fn": "transpose_chain", "id": "P%dx%d_%s_direct_put" % (pdims[0], pdims[1], ac),
                         "args": {"gdims": gdims, "pdims": pdims, "ac": ac, "kind": 1, "data_alloc": "malloc",
                                  "transpose_backend": cd.TRANSPOSE_COMM_NVSHMEM_SM, "expect_path": ["direct_puts"]}})
    for failures in run_ranks(n, "tests.gpu_bodies", "many", {"jobs": jobs}, timeout=600):
        assert failures == []


def test_rccl_backend_single_rank_world():
    # RCCL path with a one-rank world: communicator creation is skipped, transposes are local
    args = {"gdims": (16, 12, 20), "pdims": (1, 1), "ac": K.ALL_AC, "kind": 1, "transpose_backend": cd.TRANSPOSE_COMM_NCCL}
    assert B.transpose_chain(0, 1, args) == []


def test_round_trip_and_checksum_at_benchmark_size():
    """1024^3 fp64 (BASELINE.json): X->Y->Z->Y->X reproduces the input bit for bit, and every intermediate
    pencil holds a permutation of the input (wrapping 64-bit sum and xor of the raw words are invariant)."""
    from tests import gpu_util as G
    n = 1024
    h = B._handle(0)
    gd = cd.cudecompGridDescCreate(h, cd.make_config((n, n, n), (1, 1), axis_contiguous=(1, 1, 1)))
    nel = n ** 3
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    a = torch.randint(-2**62, 2**62, (nel,), dtype=torch.int64, device="cuda", generator=g)
    ref_sum, ref_xor = int(a.sum()), int(torch.bitwise_xor(a[::2], a[1::2]).sum())
    keep = a.clone()
    b = torch.zeros_like(a)
    work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * 8)
    cur, nxt = a, b
    for op in cd.OPS:
        cd.cudecompTranspose(op, h, gd, cur.data_ptr(), nxt.data_ptr(), work, cd.DOUBLE, stream=G.stream_ptr())
        torch.cuda.synchronize()
        assert int(nxt.sum()) == ref_sum
        cur, nxt = nxt, cur
    assert torch.equal(cur, keep)
    # spot-check the X->Y permutation against the index map: out[y + Y*(z + Z*x)] == in[x + X*(y + Y*z)]
    cd.cudecompTranspose("XToY", h, gd, keep.data_ptr(), b.data_ptr(), work, cd.DOUBLE, stream=G.stream_ptr())
    torch.cuda.synchronize()
    idx = torch.randint(0, n, (3, 4096), device="cuda", generator=g)
    x, y, z = idx[0], idx[1], idx[2]
    assert torch.equal(b[y + n * (z + n * x)], keep[x + n * (y + n * z)])
    cd.cudecompFree(h, gd, work)
    cd.cudecompGridDescDestroy(h, gd)


@pytest.mark.parametrize("inplace", [False, True], ids=["out_of_place", "in_place"])
@pytest.mark.parametrize("ac", [(1, 1, 1), (0, 0, 0)], ids=["axis_contiguous", "default_layout"])
def test_bench_workload_every_cell(ac, inplace):
    """bench.py's N=1 workload itself -- 1x1 grid, 1024^3 fp64, both layouts, in and out of place -- with EVERY cell of
    every hop compared on the device against the closed form of that pencil (reference:
    tests/ctest/transpose_tests.cc:356-378 compares every interior cell; nothing is sampled here either)."""
    args = {"gdims": (1024, 1024, 1024), "pdims": (1, 1), "kind": 1, "ac": ac, "inplace": inplace}
    for r in run_ranks(1, "tests.gpu_bodies", "cycle_exact", args, timeout=600):
        assert r["failures"] == []


@pytest.mark.parametrize("kind", [1, 2, 3])
def test_in_place_rotation_of_cubic_grids(kind):
    """Single rank, in place, cubic, axis-contiguous: every hop is ONE in-place rotation kernel (csrc/kernels_rotate.hip; the
    reference stages such transposes through the workspace, include/internal/transpose.h:326-362).  Against the oracle after
    every hop; cubes the tiles divide run it (counted), others and CUDECOMP_DISABLE_INPLACE_ROTATION keep the staged form."""
    tile = 8 if kind == 3 else 16
    for n in (tile, 2 * tile, 3 * tile, 5 * tile, 9 * tile):   # (walk cubes of 1, 2, 2, 4 and 8 blocks per edge, whole and ragged)
        args = {"gdims": (n, n, n), "pdims": (1, 1), "ac": K.ALL_AC, "kind": kind, "out_of_place": [False], "expect_path": ["rotations"]}
        assert B.transpose_chain(0, 1, args) == []
    # a cube the tiles do not divide, a non-cubic grid, fp32: staged, same results
    for gdims, k in (((tile + 4,) * 3, kind), ((32, 32, 48), kind), ((32, 32, 32), 0)):
        args = {"gdims": gdims, "pdims": (1, 1), "ac": K.ALL_AC, "kind": k, "out_of_place": [False]}
        assert B.transpose_chain(0, 1, args) == []
    h = B._handle(0)
    gd = cd.cudecompGridDescCreate(h, cd.make_config((tile + 4,) * 3, (1, 1), axis_contiguous=K.ALL_AC))
    assert cd.cudecompExtGetCounters(h, gd)["rotations"] == 0
    cd.cudecompGridDescDestroy(h, gd)


def test_in_place_rotation_under_every_triple_of_memory_orders():
    """Cubic single-rank grids IN PLACE under all 216 triples of memory orders (X, Y and Z pencils independently): a hop runs the
    in-place rotation kernel exactly where the planner says the two orders are a rotation of each other (the CPU planner's word,
    cudecompExtPlanTranspose(...).rotate, against the executor's counter), the staged form everywhere else, and every hop of every
    chain matches the oracle.  fp64 and complex128, two tiles per edge (432 chains in well under a second)."""
    import itertools
    perms = list(itertools.permutations((0, 1, 2)))
    triples = list(itertools.product(perms, perms, perms))
    rotated = 0
    for kind, n, step in ((1, 32, 1), (3, 16, 1)):
        for mo in triples[::step]:
            spec = cd.make_grid_spec((n, n, n), (1, 1), [list(o) for o in mo])
            want = sum(1 for op in cd.OPS if cd.cudecompExtPlanTranspose(spec, 0, op, inplace=True).rotate != 0)
            rotated += want
            args = {"gdims": (n, n, n), "pdims": (1, 1), "mem_order": mo, "kind": kind, "out_of_place": [False],
                    "expect_counts": {"rotations": want}}
            assert B.transpose_chain(0, 1, args) == [], mo
    assert rotated > 100   # (a third of the order pairs are rotations of each other)


def test_more_than_2_31_elements_per_pencil_every_cell():
    """Maximum-size edge, every cell: 2048 x 1024 x 1056 fp32 = 2.2e9 elements (> 2^31) in one pencil, each hop compared
    on the device with the closed form (low 31 bits of the global linear index, which itself exceeds 2^31)."""
    for ac in ((1, 1, 1), (0, 0, 0)):
        args = {"gdims": (2048, 1024, 1056), "pdims": (1, 1), "kind": 0, "ac": ac}
        for r in run_ranks(1, "tests.gpu_bodies", "cycle_exact", args, timeout=900):
            assert r["failures"] == []


def test_more_than_2_31_elements_per_pencil():
    """Maximum-size edge: 2048 x 1024 x 1056 fp32 = 2.2e9 elements (> 2^31) in one pencil; 64-bit indexing in
    every kernel flavour.  Properties only: the cycle returns the input bit for bit and each hop preserves the
    multiset (sum of the raw 32-bit words)."""
    from tests import gpu_util as G
    gdims = (2048, 1024, 1056)
    nel = gdims[0] * gdims[1] * gdims[2]
    assert nel > 2**31
    h = B._handle(0)
    for ac in ((1, 1, 1), (0, 0, 0)):
        gd = cd.cudecompGridDescCreate(h, cd.make_config(gdims, (1, 1), axis_contiguous=ac))
        g = torch.Generator(device="cuda")
        g.manual_seed(7)
        a = torch.randint(-2**31, 2**31 - 1, (nel,), dtype=torch.int32, device="cuda", generator=g)
        ref_sum = int(a.sum(dtype=torch.int64))
        keep_head, keep_tail = a[:4096].clone(), a[-4096:].clone()
        probe = torch.randint(0, nel, (8192,), device="cuda", generator=g)
        keep_probe = a[probe].clone()
        b = torch.empty_like(a)
        work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * 4)
        cur, nxt = a, b
        for op in cd.OPS:
            cd.cudecompTranspose(op, h, gd, cur.data_ptr(), nxt.data_ptr(), work, cd.FLOAT, stream=G.stream_ptr())
            torch.cuda.synchronize()
            assert int(nxt.sum(dtype=torch.int64)) == ref_sum, (ac, op)
            cur, nxt = nxt, cur
        assert torch.equal(cur[:4096], keep_head) and torch.equal(cur[-4096:], keep_tail)
        assert torch.equal(cur[probe], keep_probe)
        cd.cudecompFree(h, gd, work)
        cd.cudecompGridDescDestroy(h, gd)
        del a, b, cur, nxt
        torch.cuda.empty_cache()


@pytest.mark.parametrize("backend", [cd.TRANSPOSE_COMM_MPI_P2P, cd.TRANSPOSE_COMM_NVSHMEM, cd.TRANSPOSE_COMM_NVSHMEM_PL,
                                     cd.TRANSPOSE_COMM_NVSHMEM_SM],
                         ids=["peer_copy_rendezvous", "peer_copy", "peer_pipelined", "peer_put"])
def test_one_sided_exchanges_tolerate_rank_skew(backend):
    """Ranks reach every transpose tens of milliseconds apart and in changing order: the flag-ordered one-sided
    exchanges must neither overwrite a receive area that is still being unpacked nor unpack a chunk that has not
    landed (ready / landed epochs of csrc/sync.hip), with and without the per-call host rendezvous."""
    for pdims in ((2, 2), (1, 4)):
        args = {"gdims": (64, 48, 80), "pdims": pdims, "ac": K.ALL_AC, "kind": 1, "transpose_backend": backend,
                "iterations": 3, "skew_ms": 8}
        for r in run_ranks(4, "tests.gpu_bodies", "repeated_cycle", args, timeout=300):
            assert r["failures"] == []


@pytest.mark.parametrize("backends", [(cd.TRANSPOSE_COMM_NVSHMEM, cd.TRANSPOSE_COMM_MPI_P2P),
                                      (cd.TRANSPOSE_COMM_NVSHMEM_PL, cd.TRANSPOSE_COMM_NVSHMEM_SM)],
                         ids=["nvshmem+mpi_p2p", "nvshmem_pl+nvshmem_sm"])
def test_two_live_handles_transpose_alternately(backends):
    """api_tests.cc:575-656 with work: two handles, independent descriptors / workspaces / transports, exchanges of both
    in flight together, first handle finalised while the second continues (4 ranks, every cell)."""
    args = {"gdims": (64, 48, 80), "pdims": (2, 2), "kind": 1, "ac": K.ALL_AC, "backends": list(backends)}
    for r in run_ranks(4, "tests.gpu_bodies", "two_handles_alternating", args, timeout=300):
        assert r["failures"] == []
