"""HIP halo updates through the C ABI vs the oracle's restatement of the reference's analytic halo oracle
(tests/ctest/halo_tests.cc:229-253) over the reference's case matrix (:103-146)."""
import itertools

import pytest

import cudecomp_amd as cd
from tests import cases as K
from tests import gpu_bodies as B
from tests.mp import run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mo", list(itertools.permutations((0, 1, 2))), ids=lambda m: "".join(map(str, m)))
def test_periodic_self_copy_single_rank(mo):
    for kind, gdims in ((0, (9, 10, 11)), (1, (16, 12, 20)), (3, (8, 6, 10))):
        for periods in ((1, 1, 1), (0, 0, 0), (1, 0, 1)):
            for padding in (K.ZERO, (1, 2, 1)):
                args = {"gdims": gdims, "pdims": (1, 1), "mem_order": (mo, mo, mo), "kind": kind, "halo": (2, 1, 3),
                        "periods": periods, "padding": padding}
                assert B.halo_sweep(0, 1, args) == []


CASES = K.ctest_halo_cases()
PICK = [c for i, c in enumerate(CASES) if not c["name"].startswith("Baseline") or c["kind"] == 0]


def test_ctest_halo_cases_multi_rank_peer_transport():
    groups = {}
    for c in PICK:
        groups.setdefault(c["pdims"][0] * c["pdims"][1], []).append(c)
    for n, cases in sorted(groups.items()):
        jobs = [{"fn": "halo_sweep", "id": K.hcase_id(c),
                 "args": dict(c, axes=[c["axis"]], halo_backend=cd.HALO_COMM_MPI)} for c in cases]
        for failures in run_ranks(n, "tests.gpu_bodies", "many", {"jobs": jobs}, timeout=600):
            assert failures == []


@pytest.mark.parametrize("pdims", [(2, 2), (1, 4), (4, 1)], ids=lambda p: "%dx%d" % p)
def test_every_cell_closed_form_matches_the_oracle(pdims):
    """The device-side closed form that checks config 5 in every cell (gpu_bodies.halo_exact) against the oracle's
    restatement of the reference's analytic halo oracle on a small ragged grid, mixed periodicity, both layouts."""
    jobs = [{"fn": "halo_exact", "id": "ac%d_%s" % (ac, "".join(map(str, per))),
             "args": {"gdims": (21, 18, 26), "pdims": pdims, "kind": 1, "halo": (2, 1, 3), "periods": per, "axes": [0, 1, 2],
                      "ac": (ac, ac, ac), "halo_backend": cd.HALO_COMM_MPI, "check_closed_form_against_oracle": True}}
            for ac in (0, 1) for per in ((1, 1, 1), (1, 0, 1), (0, 0, 0))]
    for failures in run_ranks(4, "tests.gpu_bodies", "many", {"jobs": jobs}, timeout=600):
        assert failures == []
