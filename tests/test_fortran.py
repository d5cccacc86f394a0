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
# GPU: data path through the Fortran entry points.  The twins take the reference's Fortran test command lines
# (tests/fortran/*.f90 there): one-based --ax and --mem_order, data type from the executable name.
# backend numbers: transposes 1 MPI_P2P (xGMI peer transport in the non-MPI flavour), 4 NCCL, 6 NVSHMEM;
# halos 1 MPI, 3 NCCL, 4 NVSHMEM
def _fbin(name, dtype):
    _binary(name)
    path = os.path.join(BUILD, "fortran", "%s_%s" % (name, dtype))
    if not os.path.exists(path):
        subprocess.run(["make", "-C", os.path.join(ROOT, "fortran"), "tests"], check=True, capture_output=True)
    return path


def _run_fortran(name, dtype, nranks, lines, env=None):
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix="_cases.txt", delete=False) as f:
        f.write("\n".join(lines) + "\n")
        path = f.name
    try:
        logs = run_binary_ranks(nranks, [_fbin(name, dtype), "--testfile", path], timeout=900, extra_env=env)
    finally:
        os.unlink(path)
    out = logs[0]
    ok = out.count(" PASSED") == len(lines) and " FAILED" not in out and "Passed all tests." in out
    if not ok:  # keep every rank's output where gpurun merges it back
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "fortran_failure_%s_%d.log" % (name, os.getpid())), "w") as f:
            for r, text in enumerate(logs):
                f.write("===== rank %d =====\n%s\n" % (r, text[-20000:]))
    assert ok, out[-3000:]


TRANSPOSE_LINES = {
    1: ["--pr 1 --pc 1 --gx 16 --gy 12 --gz 10 --backend 4",
        "--pr 1 --pc 1 --gx 16 --gy 12 --gz 10 --backend 4 --acx 1 --acy 1 --acz 1 -o",
        "--pr 1 --pc 1 --gx 33 --gy 17 --gz 21 --backend 4 --acx 1 --acy 1 --acz 1 --hex 1 2 1 --hey 1 2 1 --hez 1 2 1",
        "--pr 1 --pc 1 --gx 33 --gy 17 --gz 21 --backend 1 --acy 1 --pdx 1 0 1 --pdz 1 0 1",
        "--pr 1 --pc 1 --gx 20 --gy 24 --gz 28 --backend 4 --mem_order 2 3 1 1 3 2 3 1 2 -o"],
    2: ["--pr 2 --pc 1 --gx 16 --gy 12 --gz 10 --backend 1 -o",
        "--pr 1 --pc 2 --gx 31 --gy 18 --gz 23 --backend 2 --acx 1 --acy 1 --acz 1 --hex 1 1 1 --hey 1 1 1 --hez 1 1 1 -o"],
    4: ["--pr 2 --pc 2 --gx 32 --gy 24 --gz 20 --backend 1 --acx 1 --acy 1 --acz 1 -o",
        "--pr 2 --pc 2 --gx 19 --gy 23 --gz 17 --backend 6 --hex 2 1 1 --hey 2 1 1 --hez 2 1 1 --gd 1 2 1",
        "--pr 4 --pc 1 --gx 19 --gy 23 --gz 17 --backend 8 --acx 1 --acz 1 --rank-order 2 -o",
        "--pr 0 --pc 0 --gx 32 --gy 24 --gz 20 --backend 0 --acx 1 --acy 1 --acz 1"],
}


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["R32", "R64", "C32", "C64"])
@pytest.mark.parametrize("nranks", [1, 2, 4])
def test_fortran_transpose_cycle(nranks, dtype):
    env = {"CUDECOMP_AUTOTUNE_TRANSPOSE_BACKENDS": "^NCCL,NCCL_PL"} if nranks > 1 else None
    _run_fortran("transpose_test", dtype, nranks, TRANSPOSE_LINES[nranks], env)


HALO_LINES = {
    1: ["--pr 1 --pc 1 --gx 12 --gy 10 --gz 14 --backend 3 --ax 1",
        "--pr 1 --pc 1 --gx 12 --gy 10 --gz 14 --backend 3 --ax 2 --hex 2 --hey 1 --hez 2 --hpy 0 --pdx 1 --pdz 2",
        "--pr 1 --pc 1 --gx 12 --gy 10 --gz 14 --backend 1 --ax 3 --hey 2 --hpx 0 --hpy 0 --hpz 0 --ac 1",
        "--pr 1 --pc 1 --gx 12 --gy 10 --gz 14 --backend 3 --ax 2 --mem_order 3 1 2"],
    2: ["--pr 2 --pc 1 --gx 16 --gy 12 --gz 10 --backend 1 --ax 1"],
    4: ["--pr 2 --pc 2 --gx 16 --gy 12 --gz 14 --backend 1 --ax 1 --hex 2 --hpy 0 --pdy 1",
        "--pr 2 --pc 2 --gx 16 --gy 12 --gz 14 --backend 4 --ax 2 --hez 2 --hpz 0 --ac 1",
        "--pr 2 --pc 2 --gx 16 --gy 12 --gz 14 --backend 2 --ax 3 --hpx 0 --pdx 2 --pdz 1 --mem_order 2 1 3",
        "--pr 0 --pc 0 --gx 16 --gy 12 --gz 14 --backend 0 --ax 2"],
}


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["R32", "R64", "C32", "C64"])
@pytest.mark.parametrize("nranks", [1, 2, 4])
def test_fortran_halo_update(nranks, dtype):
    env = {"CUDECOMP_AUTOTUNE_HALO_BACKENDS": "^NCCL"} if nranks > 1 else None
    _run_fortran("halo_test", dtype, nranks, HALO_LINES[nranks], env)


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
