"""The product's transpose PLANS (libcudecomp.so, cudecompExtGetTransposePlan) executed with numpy block
moves and a real multi-process exchange over torch.distributed/gloo, world size 2..4, checked against the
analytic oracle after every hop of X->Y->Z->Y->X.  This covers the N>1 host logic (peer schedule, counts,
offsets, one-sided receive offsets, staging choices) without a GPU; the HIP kernels that execute the same
plans are covered by the -m gpu tests."""
import pytest

import cudecomp_amd as cd
from tests import cases as K
from tests.mp import run_ranks

BACKENDS = [cd.TRANSPOSE_COMM_NCCL, cd.TRANSPOSE_COMM_NCCL_PL, cd.TRANSPOSE_COMM_MPI_P2P]


@pytest.mark.parametrize("pdims", [(2, 1), (1, 2)], ids=lambda p: "P%dx%d" % p)
@pytest.mark.parametrize("layout", ["default", "contiguous", "mixed"])
def test_world2_plans(pdims, layout):
    mo = {"default": None, "contiguous": None, "mixed": ((0, 1, 2), (0, 2, 1), (1, 2, 0))}[layout]
    ac = (1, 1, 1) if layout == "contiguous" else (0, 0, 0)
    args = {"gdims": (16, 12, 20), "pdims": pdims, "ac": ac, "mem_order": mo, "kind": 1, "backends": BACKENDS}
    for failures in run_ranks(2, "tests.bodies", "plan_transpose_gloo", args):
        assert failures == []


def test_world2_halos_and_padding():
    args = {"gdims": (9, 10, 11), "pdims": (2, 1), "kind": 0, "backends": BACKENDS,
            "halos": [K.IN_HALO, K.OUT_HALO, K.IN_HALO], "pads": [K.IN_PAD, K.OUT_PAD, K.IN_PAD]}
    for failures in run_ranks(2, "tests.bodies", "plan_transpose_gloo", args):
        assert failures == []


@pytest.mark.parametrize("nranks,pdims,ro", [(4, (2, 2), 0), (4, (2, 2), 2), (3, (3, 1), 0), (4, (1, 4), 0)])
def test_world34_plans(nranks, pdims, ro):
    for ac, gdd in (((0, 0, 0), None), ((1, 1, 1), (7, 8, 9))):
        args = {"gdims": (9, 10, 11), "pdims": pdims, "ac": ac, "gdims_dist": gdd, "rank_order": ro, "kind": 2,
                "backends": BACKENDS}
        for failures in run_ranks(nranks, "tests.bodies", "plan_transpose_gloo", args):
            assert failures == []


@pytest.mark.parametrize("nranks,pdims,hosts", [(3, (3, 1), "abc"), (6, (6, 1), "aabbcc"), (6, (1, 6), "aaabbb")])
def test_synthetic_host_groups(nranks, pdims, hosts):
    # reference tests/ctest/transpose_tests.cc:239-273 (SyntheticInterGroup*): ranks spread over several "hosts"
    # exercise the two-level (intra-group then inter-group) ring schedule of non-power-of-two communicators
    args = {"gdims": (13, 12, 14), "pdims": pdims, "kind": 0,
            "backends": [cd.TRANSPOSE_COMM_NCCL, cd.TRANSPOSE_COMM_NCCL_PL],
            "halos": [K.IN_HALO, K.OUT_HALO, K.IN_HALO], "pads": [K.IN_PAD, K.OUT_PAD, K.IN_PAD]}
    env = [{"CUDECOMP_HOSTNAME_OVERRIDE": "node-" + h} for h in hosts]
    for failures in run_ranks(nranks, "tests.bodies", "plan_transpose_gloo", args, per_rank_env=env):
        assert failures == []


@pytest.mark.parametrize("nranks,pdims,stages", [(2, (2, 1), 2), (2, (1, 2), 3), (4, (2, 2), 2), (4, (1, 4), 4), (3, (1, 3), 3)])
def test_staged_pipeline_over_gloo(nranks, pdims, stages):
    """The staged pipeline of the one-sided pipelined transports (NVSHMEM_PL / MPI_P2P_PL enums) on 2-4 processes: the
    product's plan is cut into stages exactly as csrc/transport.cc does, the contiguous sub-chunks of each stage travel
    over gloo, and each stage's range is unpacked from a poisoned receive area; every hop of the cycle is compared with
    the analytic oracle, in and out of place, halos and padding included."""
    for ac, halos, pads in (((1, 1, 1), None, None), ((0, 0, 0), [K.IN_HALO, K.OUT_HALO, K.IN_HALO], [K.IN_PAD, K.OUT_PAD, K.IN_PAD])):
        args = {"gdims": (12, 10, 14), "pdims": pdims, "ac": ac, "kind": 1, "stages": stages,
                "backends": [cd.TRANSPOSE_COMM_NVSHMEM_PL, cd.TRANSPOSE_COMM_MPI_P2P_PL]}
        if halos:
            args.update({"halos": halos, "pads": pads})
        for failures in run_ranks(nranks, "tests.bodies", "plan_transpose_gloo", args):
            assert failures == []


@pytest.mark.parametrize("nranks,pdims,per_cycle", [(4, (2, 2), 4), (6, (2, 3), 2), (6, (3, 2), 2)])
def test_two_hop_relay_over_gloo(nranks, pdims, per_cycle):
    """The two-hop relay of two-member exchanges (CUDECOMP_TWO_HOP_RELAY; csrc/plan.h RelayPlan) on 4 and 6 processes: the
    scatter / forward moves of the stateless planner travel over gloo, relay region and receive area start poisoned, every
    hop of the cycle is compared with the analytic oracle, in and out of place, with halos and padding."""
    for ac, halos, pads in (((1, 1, 1), None, None), ((0, 0, 0), [K.IN_HALO, K.OUT_HALO, K.IN_HALO], [K.IN_PAD, K.OUT_PAD, K.IN_PAD])):
        args = {"gdims": (13, 12, 14), "pdims": pdims, "ac": ac, "kind": 1, "relay": True, "expect_relayed": 2 * per_cycle,
                "backends": [cd.TRANSPOSE_COMM_NVSHMEM]}
        if halos:
            args.update({"halos": halos, "pads": pads})
        for failures in run_ranks(nranks, "tests.bodies", "plan_transpose_gloo", args):
            assert failures == []
