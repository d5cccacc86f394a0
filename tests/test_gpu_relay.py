"""Two-hop relay of low-fan-out exchanges (CUDECOMP_TWO_HOP_RELAY=1; csrc/plan.h RelayPlan, csrc/transport.cc
peerRelayAlltoall): on a pencil grid whose X<->Y or Y<->Z exchange has two members, every chunk is cut into one slice per
rank of the node and travels source -> relay -> destination, so that all links carry data instead of one.  Ranks share the
test box's GPU here, so this checks the protocol (flags of the communicator of all ranks, relay slots, forwarding) and the
results -- every cell of every pencil after every hop -- not the speed-up; scripts/first_multi_gpu.sh measures it.
The plan itself is simulated on the host for random decompositions in tests/test_plan_sim.py."""
import itertools

import pytest

import cudecomp_amd as cd
from tests import cases as K
from tests.mp import run_ranks
from tests.test_gpu_native import _run
from tests.test_gpu_native_sweep import _mem_orders, _tcase

pytestmark = pytest.mark.gpu
ENV = {"CUDECOMP_TWO_HOP_RELAY": "1"}


@pytest.mark.parametrize("nranks,pdims,relayed_per_cycle", [(4, (2, 2), 4), (8, (2, 4), 2), (8, (4, 2), 2), (6, (2, 3), 2)],
                         ids=["2x2", "2x4", "4x2", "2x3"])
@pytest.mark.parametrize("inplace", [False, True], ids=["out_of_place", "in_place"])
def test_relayed_cycle_every_cell(nranks, pdims, relayed_per_cycle, inplace):
    # uneven extents on purpose (slices and slots of different sizes); two cycles: relay slots and flags are reused
    for ac, gdims in ((K.ALL_AC, (132, 100, 124)), (K.DEFAULT_AC, (96, 97, 90))):
        args = {"gdims": gdims, "pdims": pdims, "kind": 1, "ac": ac, "transpose_backend": cd.TRANSPOSE_COMM_NVSHMEM,
                "inplace": inplace, "cycles": 2}
        for r in run_ranks(nranks, "tests.gpu_bodies", "cycle_exact", args, timeout=600, extra_env=ENV):
            assert r["failures"] == []
            assert r["counters"]["relayed"] == 2 * relayed_per_cycle, r["counters"]


def test_relay_off_by_default_and_on_slab_grids():
    args = {"gdims": (64, 48, 80), "pdims": (2, 2), "kind": 1, "ac": K.ALL_AC, "transpose_backend": cd.TRANSPOSE_COMM_NVSHMEM}
    for r in run_ranks(4, "tests.gpu_bodies", "cycle_exact", args, timeout=300):
        assert r["failures"] == [] and r["counters"]["relayed"] == 0
    args["pdims"] = (1, 4)
    for r in run_ranks(4, "tests.gpu_bodies", "cycle_exact", args, timeout=300, extra_env=ENV):
        assert r["failures"] == [] and r["counters"]["relayed"] == 0  # every link is busy already


def test_relayed_config3_grid_at_full_size():
    """BASELINE config 3 on the grid it is quoted on -- 1024^3 fp64, 2 x 4 -- with the X<->Y exchanges relayed: 512 MiB per
    rank cut into eight 64-MiB slices, every cell of every pencil checked."""
    args = {"gdims": (1024, 1024, 1024), "pdims": (2, 4), "kind": 1, "transpose_backend": cd.TRANSPOSE_COMM_NVSHMEM}
    for r in run_ranks(8, "tests.gpu_bodies", "cycle_exact", args, timeout=900, extra_env=ENV):
        assert r["failures"] == []
        assert r["counters"]["relayed"] == 2


def test_relayed_native_sweep_slice():
    """The reference-style case lines (memory orders, halos, padding, in / out of place) on 2 x 2 and 2 x 4 / 4 x 2 grids
    with the NVSHMEM enum relayed."""
    lines = [_tcase(2, 2, 6, extra=mo, oop=oop) for mo, oop in itertools.product(_mem_orders()[::7], (True, False))]
    lines += [_tcase(2, 2, 6, hx="1 1 1", hy="1 1 1", hz="1 1 1", px="1 1 1", pz="1 1 1", gd="16 16 16",
                     extra="--acx 1 --acy 1 --acz 1", oop=oop) for oop in (True, False)]
    _run("transpose_test_R64", 4, lines, dict(ENV))
    _run("transpose_test_C64", 4, lines[::3], dict(ENV))
    lines8 = [_tcase(pr, pc, 6, extra=mo, oop=oop) for (pr, pc), mo, oop in
              itertools.product([(2, 4), (4, 2)], _mem_orders()[::12], (True, False))]
    _run("transpose_test_R32", 8, lines8, dict(ENV))
