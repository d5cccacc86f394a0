"""Autotune entry points (cudecompGridDescCreate with options; reference src/autotune.cc): the sweep runs
through the public transposes / halo updates on the GPU, every rank ends up with the same selection, the
selection is reported back through the in/out config, and the selected configuration computes correctly."""
import pytest

import cudecomp_amd as cd
from tests import gpu_bodies as B
from tests.mp import run_ranks

pytestmark = pytest.mark.gpu


def test_autotune_single_rank(capfd):
    r = B.autotune_then_cycle(0, 1, {"gdims": (32, 24, 40), "ac": (1, 1, 1)})
    assert r["failures"] == [] and r["picked"]["pdims"] == [1, 1]
    assert 1 <= r["picked"]["tb"] <= 8 and 1 <= r["picked"]["hb"] <= 5
    out = capfd.readouterr().out
    # the log format is an interface (parsed by the reference's benchmark_runner.py)
    assert "CUDECOMP: Running transpose autotuning..." in out
    assert "CUDECOMP:\tgrid: 1 x 1, backend: NCCL" in out
    assert "CUDECOMP:\tTotal time min/max/avg/std [ms]:" in out and "CUDECOMP:\tTransposeXY time min/max/avg/std" in out
    assert "CUDECOMP: SELECTED: grid: 1 x 1, backend:" in out
    assert "CUDECOMP: Running halo autotuning..." in out and "CUDECOMP: Autotune halo axis: x" in out


@pytest.mark.parametrize("mode", ["transpose", "halo"])
def test_autotune_four_ranks_sharing_the_gpu(mode):
    # RCCL cannot put several ranks on one device, so the NCCL candidates are disabled here
    args = {"gdims": (32, 24, 40), "disable_nccl": True, "skip_threshold": 0.8 if mode == "transpose" else 0.0,
            "grid_mode_halo": mode == "halo"}
    res = run_ranks(4, "tests.gpu_bodies", "autotune_then_cycle", args, timeout=600)
    picks = [r["picked"] for r in res]
    assert all(p == picks[0] for p in picks), picks
    assert picks[0]["pdims"][0] * picks[0]["pdims"][1] == 4
    assert not cd.TRANSPOSE_COMM_NCCL <= picks[0]["tb"] <= cd.TRANSPOSE_COMM_NCCL_PL
    for r in res:
        assert r["failures"] == []
