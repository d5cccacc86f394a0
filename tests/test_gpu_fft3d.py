"""BASELINE config 4 harness: distributed 3-D C2C FFT (hipFFT/rocFFT local lines + the library's
transposes), checked by the analytic plane-wave spectrum on the distributed Z pencils and by the
forward+inverse residual with the reference's tolerances (benchmark/benchmark.cu:23-27)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "benchmark"))
import run_fft3d  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("extra", [[], ["--double", "-o"], ["--default-layout"], ["--default-layout", "-o", "--double"]],
                         ids=["c64_contig_inplace", "c128_contig_oop", "c64_default_inplace", "c128_default_oop"])
def test_fft3d_single_rank(extra):
    rec, _ = run_fft3d.run(1, ["--gx", "64", "--gy", "48", "--gz", "80", "--pr", "1", "--pc", "1", "--backend", "4",
                               "--warmup", "1", "--trials", "2"] + extra)
    assert rec["ok"], rec


@pytest.mark.parametrize("pr,pc,extra", [(2, 2, []), (1, 4, ["-o"]), (4, 1, ["--default-layout", "--double"]),
                                         (2, 2, ["--default-layout", "-o"])])
def test_fft3d_four_ranks_peer_transport(pr, pc, extra):
    # uneven grid on purpose (reference tests/test_config.yaml uses 128 x 124 x 132)
    rec, _ = run_fft3d.run(4, ["--gx", "64", "--gy", "60", "--gz", "68", "--pr", str(pr), "--pc", str(pc),
                               "--backend", "8", "--warmup", "1", "--trials", "2"] + extra)
    assert rec["ok"], rec
    assert rec["pdims"] == [pr, pc]


def _same_spectrum(a, b, rel):
    da = complex(*a["spectrum_checksum"]) - complex(*b["spectrum_checksum"])
    return abs(da) <= rel * max(a["spectrum_abs_sum"], b["spectrum_abs_sum"])


@pytest.mark.parametrize("mode", [[], ["--r2c"]], ids=["c2c", "r2c"])
@pytest.mark.parametrize("prec", [[], ["--double"]], ids=["single", "double"])
@pytest.mark.parametrize("nranks,pr,pc,expect", [(1, 1, 1, "xyz"), (4, 1, 4, "xy"), (4, 4, 1, "yz")])
def test_fft3d_slab_shortcuts_agree_with_the_plain_passes(nranks, pr, pc, expect, prec, mode):
    """benchmark.cu:340-373: a process grid with a 1 lets whole planes (or the whole array) be transformed at once --
    one 3-D FFT on 1x1, 2-D x-y planes on 1xQ, 2-D y-z planes on Px1 (skipping the Y<->Z transposes).  Shortcuts on and off
    must produce the same spectrum (weighted checksum over global indices, 5e-4 single / 1e-10 double, the round-trip
    tolerances of benchmark.cu:23-27) and both must pass the plane-wave and round-trip checks; also for the
    real-to-complex flavour (:238-330), whose decomposed grid is (gx/2+1) x gy x gz."""
    base = ["--gx", "64", "--gy", "60", "--gz", "68", "--pr", str(pr), "--pc", str(pc), "--backend", "8" if nranks > 1 else "4",
            "--warmup", "1", "--trials", "2"] + prec + mode
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=2) as ex:  # the two runs side by side: at most 2 x 4 ranks on the GPU
        f_on, f_off = ex.submit(run_fft3d.run, nranks, base), ex.submit(run_fft3d.run, nranks, base + ["--no-slab-opt"])
        (on, _), (off, _) = f_on.result(), f_off.result()
    assert on["ok"] and off["ok"], (on, off)
    assert on["slab"] == expect and off["slab"] == "none", (on["slab"], off["slab"])
    assert on["mode"] == ("r2c" if mode else "c2c") and on["spectrum_checked"] and off["spectrum_checked"]
    assert _same_spectrum(on, off, 1e-10 if prec else 5e-4), (on["spectrum_checksum"], off["spectrum_checksum"])


def test_fft3d_r2c_round_trip_on_a_pencil_grid():
    """R2C / C2R on a 2 x 2 pencil grid (no shortcut applies), out of place, odd-ish extents; and the same spectrum as the
    one-rank single-3-D-FFT run of the same field would need the same random field on every decomposition -- the field is
    seeded per rank, so only the checks of each run are asserted here."""
    for extra in ([], ["--double", "-o"], ["--default-layout"]):
        rec, _ = run_fft3d.run(4, ["--gx", "62", "--gy", "60", "--gz", "68", "--pr", "2", "--pc", "2", "--backend", "8",
                                   "--warmup", "1", "--trials", "2", "--r2c"] + extra)
        assert rec["ok"] and rec["mode"] == "r2c" and rec["slab"] == "none", rec


@pytest.mark.parametrize("nranks,args", [(1, ["--n", "48"]), (1, ["--n", "32", "--default-layout"]),
                                         (4, ["--n", "48", "--pr", "2", "--pc", "2", "--backend", "1"]),
                                         (4, ["--n", "40", "--pr", "1", "--pc", "4", "--backend", "8", "--default-layout"])])
def test_poisson_example(nranks, args):
    """examples/cc/poisson: spectral Poisson solve (FFT -> divide by -|k|^2 on the Z pencils -> inverse FFT) against the
    analytic solution; exercises in-place complex128 transposes from a solver's point of view."""
    import json
    import subprocess

    from tests.mp import run_binary_ranks
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples", "cc"), "poisson"])
    logs = run_binary_ranks(nranks, [os.path.join(ROOT, "examples", "cc", "poisson")] + args)
    recs = [json.loads(line) for text in logs for line in text.splitlines() if line.startswith("{")]
    assert len(recs) == nranks and all(r["ok"] and r["max_abs_err"] < 1e-11 for r in recs), recs


@pytest.mark.parametrize("backend,name", [(6, "nvshmem"), (8, "nvshmem_sm")])
def test_config4_fft_full_size_on_2x4(backend, name):
    """BASELINE config 4 on ITS grid: 1024^3 complex<fp32>, 2 x 4 pencils (8 ranks sharing the GPU), hipFFT lines +
    the library's transposes, forward + inverse residual within the reference's tolerance (benchmark/benchmark.cu:23-27,
    489-611: max-abs round-trip error <= 5e-4 for complex<float>), plus the plane-wave spectrum check on the distributed
    Z pencils.  Every rank holds 1 GiB of data + 2 GiB of workspace."""
    rec, ranks = run_fft3d.run(8, ["--gx", "1024", "--gy", "1024", "--gz", "1024", "--pr", "2", "--pc", "4", "--backend",
                                   str(backend), "--warmup", "1", "--trials", "2"], timeout=900)
    assert rec["ok"], rec
    assert rec["pdims"] == [2, 4]
    assert rec["roundtrip_max_abs_err"] <= 5e-4, rec
