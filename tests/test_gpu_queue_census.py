"""Ranks that share a GPU compete for its hardware queue slots: the library counts the compute queues
all processes hold on its device from the kernel driver's tables and knows the device's slots (`num_cp_queues`), so that it
can say when the device is oversubscribed.  Here: the count sees at least the ranks of the job, the slots are reported, and
a job of four ranks stays below them."""
import pytest

import cudecomp_amd as cd
from tests import cases as K
from tests.mp import run_ranks

pytestmark = pytest.mark.gpu


def test_queue_census_sees_the_ranks_of_the_job():
    args = {"gdims": (64, 48, 80), "pdims": (2, 2), "kind": 1, "ac": K.ALL_AC, "transpose_backend": cd.TRANSPOSE_COMM_NVSHMEM}
    res = run_ranks(4, "tests.gpu_bodies", "queue_census", args, timeout=300)
    for r in res:
        assert r["failures"] == []
    if any(r["compute"] < 0 for r in res):
        pytest.skip("the kernel driver's queue tables are not readable here")
    for r in res:
        assert r["slots"] >= 8, r
        assert 4 <= r["compute"] <= r["slots"], r  # at least one queue per rank; four ranks do not oversubscribe the device
