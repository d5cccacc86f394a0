"""The library's RCCL transport on MORE than one rank, on a one-GPU box.

RCCL itself will not put several ranks on one device, so these tests load `tests/shim/libfake_rccl.so` (a
test-only stand-in for ncclSend/ncclRecv/ncclGroup*/ncclAllToAll that moves the bytes through /dev/shm) ahead of
libcudecomp.so.  Everything above those calls is the product's code: grouped a2a(v) with global peer ranks,
ncclAllToAll for slab grids, the pipelined per-peer variant with its side stream and events, the halo pair
exchange and the RCCL candidates of the autotuner.  Results are compared with the oracle bit for bit, exactly
like the peer-transport tests."""
import os
import subprocess

import pytest

import cudecomp_amd as cd
from tests import cases as K
from tests.mp import ROOT, run_ranks

pytestmark = pytest.mark.gpu
SHIM_DIR = os.path.join(ROOT, "tests", "shim")
SHIM = os.path.join(SHIM_DIR, "libfake_rccl.so")


@pytest.fixture(scope="module")
def shim_env():
    if not os.path.exists(SHIM):
        subprocess.run(["make", "-C", SHIM_DIR], check=True, capture_output=True)
    return {"CUDECOMP_TEST_RCCL_SHIM": SHIM}


def _run(n, jobs, env):
    for failures in run_ranks(n, "tests.gpu_bodies", "many", {"jobs": jobs}, timeout=600, extra_env=env):
        assert failures == []


@pytest.mark.parametrize("n", [2, 3, 4])
def test_transpose_cycle_rccl(n, shim_env):
    jobs = []
    for pdims in [(2, 2), (1, 4), (4, 1), (2, 1), (1, 2), (3, 1), (1, 3)]:
        if pdims[0] * pdims[1] != n:
            continue
        for ac, backend, kind, gdims in ((K.DEFAULT_AC, cd.TRANSPOSE_COMM_NCCL, 1, (32, 24, 40)),
                                         (K.ALL_AC, cd.TRANSPOSE_COMM_NCCL, 0, (32, 24, 40)),
                                         (K.ALL_AC, cd.TRANSPOSE_COMM_NCCL_PL, 3, (32, 24, 40)),
                                         (K.DEFAULT_AC, cd.TRANSPOSE_COMM_NCCL_PL, 1, (31, 25, 38)),
                                         (K.ALL_AC, cd.TRANSPOSE_COMM_NCCL, 2, (29, 23, 37))):
            jobs.append({"fn": "transpose_chain", "id": "P%dx%d_%s_b%d_k%d" % (pdims[0], pdims[1], ac, backend, kind),
                         "args": {"gdims": gdims, "pdims": pdims, "ac": ac, "kind": kind,
                                  "transpose_backend": backend}})
    _run(n, jobs, shim_env)


@pytest.mark.parametrize("backend", [cd.TRANSPOSE_COMM_NCCL, cd.TRANSPOSE_COMM_NCCL_PL], ids=["nccl", "nccl_pipelined"])
def test_ctest_transpose_cases_rccl(backend, shim_env):
    multi = [c for c in K.ctest_transpose_cases(pdims_list=((2, 2),)) if tuple(c["pdims"]) != (1, 1)]
    pick = {}
    for c in multi:
        pick.setdefault((c["name"], c["op"], c["out_of_place"]), c)
    groups = {}
    for c in pick.values():
        groups.setdefault(c["pdims"][0] * c["pdims"][1], []).append(c)
    for n, cs in sorted(groups.items()):
        _run(n, [{"fn": "single_transpose", "id": K.case_id(c), "args": dict(c, transpose_backend=backend)}
                 for c in cs], shim_env)


def test_ctest_halo_cases_rccl(shim_env):
    cases = [c for c in K.ctest_halo_cases() if not c["name"].startswith("Baseline") or c["kind"] == 0]
    groups = {}
    for c in cases:
        groups.setdefault(c["pdims"][0] * c["pdims"][1], []).append(c)
    for n, cs in sorted(groups.items()):
        if n == 1:
            continue
        _run(n, [{"fn": "halo_sweep", "id": K.hcase_id(c),
                  "args": dict(c, axes=[c["axis"]], halo_backend=cd.HALO_COMM_NCCL)} for c in cs], shim_env)


@pytest.mark.parametrize("mode", ["transpose", "halo"])
def test_autotune_with_rccl_candidates(mode, shim_env):
    # all backends compete (as on a real multi-GPU node); whatever wins, every rank agrees and the data is right
    args = {"gdims": (32, 24, 40), "disable_nccl": False, "skip_threshold": 0.0, "grid_mode_halo": mode == "halo"}
    res = run_ranks(4, "tests.gpu_bodies", "autotune_then_cycle", args, timeout=600, extra_env=shim_env)
    picks = [r["picked"] for r in res]
    assert all(p == picks[0] for p in picks), picks
    assert picks[0]["pdims"][0] * picks[0]["pdims"][1] == 4
    for r in res:
        assert r["failures"] == []
