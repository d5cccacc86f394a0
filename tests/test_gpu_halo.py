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
