"""cudecompMalloc / cudecompFree on a multi-rank job: released workspaces are parked WITH the peers' IPC mappings and
handed out again (csrc/transport.cc: takeFromPool / park), new mappings are verified by page tags.  Reference contract:
cudecompMalloc / cudecompFree are collective (include/cudecomp.h:433-455); what they do underneath is the library's
business."""
import pytest

import cudecomp_amd as cd
from tests import cases as K
from tests.mp import run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pool", ["default", "off"])
def test_released_workspaces_are_recycled(pool):
    args = {"gdims": (64, 48, 80), "pdims": (2, 2), "kind": 1, "ac": K.ALL_AC, "transpose_backend": cd.TRANSPOSE_COMM_NVSHMEM}
    env = {"CUDECOMP_WORKSPACE_POOL_MIB": "0"} if pool == "off" else {}
    for r in run_ranks(4, "tests.gpu_bodies", "pool_behaviour", args, timeout=300, extra_env=env):
        assert r["failures"] == []
        assert r["distinct_large"]
        if pool == "default":
            assert r["same_pointer"] and r["pool_hits"] == 2
        else:
            assert r["pool_hits"] == 0


def test_pool_gives_memory_back_under_pressure():
    """Round-3 advice on the pool: parked memory is released by cudecompExtTrimWorkspacePool, a cudecompMalloc that runs
    out of memory drains the pool before it gives up, and an allocation failure is agreed on by all ranks (same result
    code everywhere, no rank left waiting in a collective)."""
    args = {"gdims": (64, 48, 80), "pdims": (2, 2), "kind": 1, "ac": K.ALL_AC, "transpose_backend": cd.TRANSPOSE_COMM_NVSHMEM}
    res = run_ranks(4, "tests.gpu_bodies", "pool_pressure", args, timeout=600)
    for r in res:
        assert r["failures"] == [], r
        assert r["parked_bytes"] >= 512 << 20 and r["parked_after_trim"] == 0
        assert r["parked_before_big"] >= 8 << 30
        assert r["impossible_code"] == res[0]["impossible_code"] != 0
        assert r["parked_after_impossible"] == 0
