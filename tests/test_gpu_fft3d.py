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
