"""Pin the oracle's halo restatement against the reference's analytic halo oracle
(tests/ctest/halo_tests.cc:229-253) over the reference's case matrix (:103-146)."""
import itertools

import pytest

from tests import cases as K
from tests import oracle_runner as R


@pytest.mark.parametrize("staged", [False, True], ids=["direct", "staged"])
@pytest.mark.parametrize("c", K.ctest_halo_cases(), ids=K.hcase_id)
def test_ctest_halo_cases(c, staged):
    ok, msg = R.run_halo_case(c, staged=staged)
    assert ok, msg


@pytest.mark.parametrize("mo", list(itertools.permutations((0, 1, 2))), ids=lambda m: "".join(map(str, m)))
@pytest.mark.parametrize("axis", [0, 1, 2])
def test_halo_all_mem_orders(axis, mo):
    # legacy halo sweep: every memory order of the pencil (tests/test_runner.py:80-90, is_halo_test)
    for pdims in ((2, 2), (1, 4), (4, 1), (1, 1), (3, 2)):
        for periods in ((1, 1, 1), (0, 0, 0), (1, 0, 1)):
            for padding in (K.ZERO, (1, 2, 1)):
                c = K.hcase("Sweep", axis, gdims=(16, 12, 20), pdims=pdims, mem_order=(mo, mo, mo),
                            halo=(2, 1, 3), periods=periods, padding=padding)
                ok, msg = R.run_halo_case(c)
                assert ok, (pdims, periods, padding, msg)


def test_halo_wider_than_neighbor_is_invalid():
    # include/internal/halo.h:120-144
    c = K.hcase("TooWide", 0, gdims=(9, 4, 11), pdims=(2, 2), halo=(1, 3, 1))
    ok, msg = R.run_halo_case(c)
    assert not ok and "rc=1" in msg
