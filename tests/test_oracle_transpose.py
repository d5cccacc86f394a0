"""Pin the oracle's transpose restatement against the reference's analytic test oracle over the
reference's own case matrices (tests/ctest/transpose_tests.cc, tests/test_config.yaml)."""
import itertools

import pytest

from oracle import oracle as orc
from tests import cases as K
from tests import oracle_runner as R


@pytest.mark.parametrize("pipelined", [False, True], ids=["a2a", "pipelined"])
@pytest.mark.parametrize("c", K.ctest_transpose_cases(), ids=K.case_id)
def test_ctest_cases(c, pipelined):
    ok, msg = R.run_transpose_case(c, pipelined=pipelined)
    assert ok, msg


SMALL = (16, 12, 20)  # uneven over 3 and over 8


@pytest.mark.parametrize("pdims", [(1, 1), (1, 4), (2, 2), (4, 1), (3, 2), (1, 3)], ids=lambda p: "P%dx%d" % p)
@pytest.mark.parametrize("mo", K.mem_order_combos(), ids=lambda m: "".join("".join(map(str, r)) for r in m))
def test_all_mem_orders_cycle(mo, pdims):
    g = orc.Grid(SMALL, pdims, mem_order=mo)
    z = (K.ZERO,) * 3
    for oop in (True, False):
        for pl in (False, True):
            ok, msg = R.run_transpose_cycle(g, 0, z, z, oop, pipelined=pl)
            assert ok, (oop, pl, msg)


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("pdims", [(1, 4), (2, 2), (4, 1)], ids=lambda p: "P%dx%d" % p)
def test_legacy_grid_default_and_contiguous(pdims, kind):
    # tests/test_config.yaml:23-25 (128 x 124 x 132) - all four dtypes
    for ac in (K.DEFAULT_AC, K.ALL_AC):
        g = orc.Grid((128, 124, 132), pdims, axis_contiguous=ac)
        z = (K.ZERO,) * 3
        ok, msg = R.run_transpose_cycle(g, kind, z, z, True)
        assert ok, msg


HP = [(0, 0, 0), (1, 1, 1)]


@pytest.mark.parametrize("pdims", [(2, 2), (1, 4), (3, 1)], ids=lambda p: "P%dx%d" % p)
@pytest.mark.parametrize("mo", K.mem_order_combos()[::5], ids=lambda m: "".join("".join(map(str, r)) for r in m))
def test_halo_padding_mix(mo, pdims):
    # transpose_test_halo / _padding / _mix sweeps of tests/test_config.yaml (hex == hez, pdx == pdz)
    g = orc.Grid(SMALL, pdims, mem_order=mo)
    for hx, hy, px, py in itertools.product(HP, HP, HP, HP):
        if not (any(hx) or any(hy) or any(px) or any(py)):
            continue
        for oop, pl in ((True, False), (False, True)):
            ok, msg = R.run_transpose_cycle(g, 0, (hx, hy, hx), (px, py, px), oop, pipelined=pl)
            assert ok, (hx, hy, px, py, oop, pl, msg)


@pytest.mark.parametrize("rank_order", [0, 1, 2])
@pytest.mark.parametrize("pdims", [(2, 2), (2, 3)], ids=lambda p: "P%dx%d" % p)
def test_gdims_dist_and_rank_order(pdims, rank_order):
    # transpose_test_gdimdist (gd = 16 16 16 -> gdims_dist = gdims - 16) and transpose_test_rank_order
    gd = tuple(x - 16 for x in (48, 44, 52))
    for ac in (K.DEFAULT_AC, K.ALL_AC, (1, 0, 1)):
        g = orc.Grid((48, 44, 52), pdims, gdims_dist=gd, rank_order=rank_order, axis_contiguous=ac)
        z = (K.ZERO,) * 3
        for oop in (True, False):
            ok, msg = R.run_transpose_cycle(g, 1, z, z, oop)
            assert ok, msg
        ok, msg = R.run_transpose_cycle(g, 0, ((1, 1, 1),) * 3, ((1, 0, 2),) * 3, True, pipelined=True)
        assert ok, msg


def test_empty_pencils_not_supported():
    # include/internal/transpose.h:257-259
    g = orc.Grid((1, 4, 4), (2, 2))
    c = K.tcase("Empty", "XToY", gdims=(1, 4, 4))
    ok, msg = R.run_transpose_case(c, grid=g)
    assert not ok and "rc=2" in msg


def test_axis_contiguous_combinations():
    # transpose_test_ac sweep (acx, acy, acz in {0,1})
    for ac in itertools.product((0, 1), repeat=3):
        g = orc.Grid(SMALL, (2, 2), axis_contiguous=ac)
        z = (K.ZERO,) * 3
        for oop in (True, False):
            ok, msg = R.run_transpose_cycle(g, 2, z, z, oop)
            assert ok, (ac, msg)


@pytest.mark.parametrize("ranks,pr,pc,contiguous", [(1, 1, 1, 1), (4, 2, 2, 1), (4, 1, 4, 0), (6, 3, 2, 1), (8, 2, 4, 0)])
def test_host_mpi_cpu_path_round_trip(ranks, pr, pc, contiguous):
    """oracle/cpu_mpi_cycle (the host-MPI CPU baseline of bench.py): the oracle's pack / unpack around a real
    MPI_Alltoallv between processes must return every rank's X pencil after the four hops."""
    import json
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mpirun = shutil.which("mpirun") or "/opt/conda/bin/mpirun"
    if not os.path.exists(mpirun) or not os.path.exists("/opt/conda/include/mpi.h"):
        pytest.skip("no MPI installation in this image")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "cpu_mpi_cycle"])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([mpirun, "-np", str(ranks), os.path.join(root, "oracle", "cpu_mpi_cycle"), "30", str(pr), str(pc),
                          str(contiguous), "0", "1"], env=env, capture_output=True, text=True, timeout=120)
    rec = json.loads([line for line in out.stdout.splitlines() if line.startswith("{")][-1])
    assert out.returncode == 0 and rec["round_trip_ok"] and rec["ranks"] == ranks, out.stdout + out.stderr
