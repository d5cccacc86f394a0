"""Fortran module (`fortran/cudecomp_m.f90`, module `cudecomp`) through its test twins in tests/fortran.

CPU part: the geometry API on 1 and 4 ranks, one-based conventions checked against the reference's golden
vectors (tests/golden, transcribed from tests/ctest/api_tests.cc) and the oracle.  GPU part: transposes and halo
updates on device buffers, each program comparing every stage with the closed-form pencil contents
(same analytic oracle as the C/Python parity tests).  Everything is skipped when no Fortran compiler produced
the binaries (fortran/build, made by `__graft_entry__.build()` with amdflang)."""
import json
import os
import shutil
import subprocess

import pytest

from oracle import oracle as orc
from tests.mp import ROOT, run_binary_ranks

BUILD = os.path.join(ROOT, "fortran", "build")


def _binary(name):
    path = os.path.join(BUILD, name)
    if not os.path.exists(path):
        if shutil.which("amdflang") is None:
            pytest.skip("no Fortran compiler (amdflang) and no prebuilt fortran/build/%s" % name)
        subprocess.run(["make", "-C", os.path.join(ROOT, "cudecomp_amd")], check=True, capture_output=True)
        subprocess.run(["make", "-C", os.path.join(ROOT, "fortran"), "tests"], check=True, capture_output=True)
    return path


def _records(logs):
    out = {}
    for text in logs:
        for line in text.splitlines():
            f = line.split()
            if f and f[0] in ("PINFO", "HALOWS", "TRANSWS", "SHIFT", "DONE", "PASS"):
                out.setdefault(f[0], []).append([int(x) for x in f[1:]])
    return out


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


@pytest.mark.parametrize("variant", ["row_major", "col_major", "gdims_dist"])
def test_fortran_pencil_info_golden_4_ranks(golden_dir, variant):
    gold = _load(golden_dir, "pencil_info.json")
    rank_order = 2 if variant == "col_major" else 0
    rec = _records(run_binary_ranks(4, [_binary("api_test"), rank_order, 1 if variant == "gdims_dist" else 0]))
    assert sorted(r[0] for r in rec["DONE"]) == [0, 1, 2, 3]
    got = {(r[0], r[1]): r[2:] for r in rec["PINFO"]}
    assert len(got) == 12
    for g in gold[variant]:
        v = got[(g["rank"], g["axis"] + 1)]  # the Fortran API counts axes from 1
        assert v[0:3] == g["shape"]
        assert v[3:6] == [x + 1 for x in g["lo"]]  # one-based global coordinates
        assert v[6:9] == [x + 1 for x in g["hi"]]
        assert v[9:12] == [x + 1 for x in g["order"]]
        assert v[12:15] == g["halo_extents"] and v[15:18] == g["padding"] and v[18] == g["size"]


@pytest.mark.parametrize("variant", ["row_major", "col_major"])
def test_fortran_shifted_rank_golden_4_ranks(golden_dir, variant):
    gold = _load(golden_dir, "shifted_rank.json")
    rec = _records(run_binary_ranks(4, [_binary("api_test"), 2 if variant == "col_major" else 0, 0]))
    got = {(r[0], r[1], r[2], r[3]): r[4] for r in rec["SHIFT"]}  # (rank, dim1, disp, periodic) -> rank
    checked = 0
    for q in gold[variant]:
        if q["axis"] != 0:
            continue  # the twin queries x-pencils (axis 1 in Fortran terms)
        for rank, want in enumerate(q["expected_by_rank"]):
            assert got[(rank, q["dim"] + 1, q["displacement"], int(q["periodic"]))] == want, q
            checked += 1
    assert checked >= 16


@pytest.mark.parametrize("nranks", [1, 2, 4])
def test_fortran_workspace_sizes_match_oracle(nranks):
    rec = _records(run_binary_ranks(nranks, [_binary("api_test"), 0, 0]))
    pdims = (2, 2) if nranks == 4 else (1, nranks)
    g = orc.Grid((9, 10, 11), pdims)
    for rank, ws in rec["TRANSWS"]:
        assert ws == g.transpose_workspace_size()
    for rank, axis, ws in rec["HALOWS"]:
        assert ws == g.halo_workspace_size(rank, axis - 1, (1, 2, 1))
    for r in rec["PINFO"]:
        info = g.pencil_info(r[0], r[1] - 1, (1, 2, 1), (1, 0, 2)).as_dict()
        assert r[2:5] == list(info["shape"]) and r[-1] == info["size"]


# ------------------------------------------------------------------------------------------------------------
# GPU: data path through the Fortran entry points
# backend numbers: transposes 1 MPI_P2P (xGMI peer transport in the non-MPI flavour), 4 NCCL, 6 NVSHMEM;
# halos 1 MPI, 3 NCCL, 4 NVSHMEM
TRANSPOSE_CASES = [
    # nranks, gdims, pdims, backend, axis_contiguous, halo, inplace, dtype
    (1, (16, 12, 10), (1, 1), 4, (0, 0, 0), (0, 0, 0), 0, 2),
    (1, (16, 12, 10), (1, 1), 4, (1, 1, 1), (0, 0, 0), 0, 2),
    (1, (33, 17, 21), (1, 1), 4, (1, 1, 1), (1, 2, 1), 0, 1),
    (1, (33, 17, 21), (1, 1), 1, (0, 1, 0), (0, 0, 0), 1, 3),
    (1, (20, 24, 28), (1, 1), 4, (1, 1, 1), (0, 0, 0), 1, 4),
    (2, (16, 12, 10), (2, 1), 1, (0, 0, 0), (0, 0, 0), 0, 2),
    (2, (31, 18, 23), (1, 2), 2, (1, 1, 1), (1, 1, 1), 0, 1),
    (4, (32, 24, 20), (2, 2), 1, (1, 1, 1), (0, 0, 0), 0, 2),
    (4, (19, 23, 17), (2, 2), 6, (0, 0, 0), (2, 1, 1), 1, 4),
    (4, (19, 23, 17), (4, 1), 8, (1, 0, 1), (0, 0, 0), 0, 3),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", TRANSPOSE_CASES, ids=lambda c: "r%d-%s-p%dx%d-b%d-dt%d" % (
    c[0], "x".join(map(str, c[1])), c[2][0], c[2][1], c[3], c[7]))
def test_fortran_transpose_cycle(case):
    nranks, gd, pd, backend, ac, halo, inplace, dtype = case
    argv = [_binary("transpose_test"), *gd, *pd, backend, *ac, *halo, inplace, dtype]
    rec = _records(run_binary_ranks(nranks, argv))
    assert sorted(r[0] for r in rec["PASS"]) == list(range(nranks))


HALO_CASES = [
    # nranks, gdims, pdims, backend, axis (one-based), halo, periods, padding, axis_contiguous
    (1, (12, 10, 14), (1, 1), 3, 1, (1, 1, 1), (1, 1, 1), (0, 0, 0), 0),
    (1, (12, 10, 14), (1, 1), 3, 2, (2, 1, 2), (1, 0, 1), (1, 0, 2), 0),
    (1, (12, 10, 14), (1, 1), 1, 3, (1, 2, 1), (0, 0, 0), (0, 0, 0), 1),
    (2, (16, 12, 10), (2, 1), 1, 1, (1, 1, 1), (1, 1, 1), (0, 0, 0), 0),
    (4, (16, 12, 14), (2, 2), 1, 1, (2, 1, 1), (1, 0, 1), (0, 1, 0), 0),
    (4, (16, 12, 14), (2, 2), 4, 2, (1, 1, 2), (1, 1, 0), (0, 0, 0), 1),
    (4, (16, 12, 14), (2, 2), 2, 3, (1, 1, 1), (0, 1, 1), (2, 0, 1), 0),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", HALO_CASES, ids=lambda c: "r%d-p%dx%d-b%d-axis%d" % (
    c[0], c[2][0], c[2][1], c[3], c[4]))
def test_fortran_halo_update(case):
    nranks, gd, pd, backend, axis, halo, periods, pad, ac = case
    argv = [_binary("halo_test"), *gd, *pd, backend, axis, *halo, *periods, *pad, ac]
    rec = _records(run_binary_ranks(nranks, argv))
    assert sorted(r[0] for r in rec["PASS"]) == list(range(nranks))


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [1, 2])
def test_fortran_basic_usage_example(nranks):
    exe = os.path.join(BUILD, "basic_usage_f")
    if not os.path.exists(exe):
        if shutil.which("amdflang") is None:
            pytest.skip("no Fortran compiler (amdflang) and no prebuilt example")
        subprocess.run(["make", "-C", os.path.join(ROOT, "fortran"), "example"], check=True, capture_output=True)
    logs = run_binary_ranks(nranks, [exe])
    assert sum("round trip OK" in text for text in logs) == nranks, logs
