"""Performance report (CUDECOMP_ENABLE_PERFORMANCE_REPORT and friends): same environment switches, CSV file names
and column headers as the reference (docs/env_vars.rst, src/performance.cc:44-153, 700-770)."""
import tempfile

import pytest

import cudecomp_amd as cd
from tests import cases as K
from tests.mp import run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nranks,pdims", [(1, (1, 1)), (4, (2, 2))])
def test_performance_report_csv(nranks, pdims):
    outdir = tempfile.mkdtemp(prefix="cudecomp_perf_")
    env = {"CUDECOMP_ENABLE_PERFORMANCE_REPORT": "1", "CUDECOMP_PERFORMANCE_REPORT_DETAIL": "2",
           "CUDECOMP_PERFORMANCE_REPORT_SAMPLES": "4", "CUDECOMP_PERFORMANCE_REPORT_WARMUP_SAMPLES": "1",
           "CUDECOMP_PERFORMANCE_REPORT_WRITE_DIR": outdir}
    args = {"gdims": (32, 24, 40), "pdims": pdims, "ac": K.ALL_AC, "kind": 1,
            "transpose_backend": cd.TRANSPOSE_COMM_MPI_P2P, "halo_backend": cd.HALO_COMM_MPI, "repeat": 4}
    res = run_ranks(nranks, "tests.gpu_bodies", "perf_report", args, timeout=600, extra_env=env)
    files = [r["files"] for r in res if r["files"]][0]
    stem = "tcomm_1-hcomm_1-pdims_%dx%d-gdims_32x24x40-memorder_012120201.csv" % pdims
    for table in ("transpose-aggregated", "transpose-samples", "halo-aggregated", "halo-samples"):
        assert "cudecomp-perf-report-%s-%s" % (table, stem) in files, sorted(files)
    agg = files["cudecomp-perf-report-transpose-aggregated-" + stem].splitlines()
    assert agg[0] == "# Transpose backend: MPI_P2P" and agg[2] == "# Process grid: [%d, %d]" % pdims
    head = [i for i, line in enumerate(agg) if not line.startswith("#")][0]
    assert agg[head] == ("operation,dtype,input_halo_extents,output_halo_extents,input_padding,output_padding,inplace,"
                         "managed,samples,total_ms,A2A_ms,local_ms,A2A_BW_GBps")
    rows = [line.split(",") for line in agg[head + 1:]]
    assert [r[0] for r in rows] == ["TransposeXY", "TransposeYZ", "TransposeZY", "TransposeYX"]
    assert all(r[1] == "D" and r[2] == '"[0' for r in rows)
    for line in agg[head + 1:]:
        tail = line.rsplit(",", 5)
        assert tail[1] == "3" and float(tail[2]) > 0  # 4 calls, the first one is warm-up
    samples = files["cudecomp-perf-report-transpose-samples-" + stem].splitlines()
    shead = [i for i, line in enumerate(samples) if not line.startswith("#")][0]
    assert samples[shead].endswith("rank,sample,total_ms,A2A_ms,local_ms,A2A_BW_GBps")
    ranks_seen = {line.rsplit(",", 6)[1] for line in samples[shead + 1:]}
    assert ranks_seen == {str(r) for r in range(nranks)}
    hal = files["cudecomp-perf-report-halo-aggregated-" + stem].splitlines()
    hhead = [i for i, line in enumerate(hal) if not line.startswith("#")][0]
    assert hal[hhead] == "operation,dtype,dim,halo_extent,periods,padding,managed,samples,total_ms,SR_ms,local_ms,SR_BW_GBps"
    assert [line.split(",")[0:3] for line in hal[hhead + 1:]] == [["HaloX", "D", "1"], ["HaloY", "D", "2"]]


def test_performance_report_of_in_place_rotations():
    """Single rank, cubic, in place: the hops are in-place rotation kernels (csrc/kernels_rotate.hip), and the report has them as
    in-place rows (inplace = T) with their samples and a local time."""
    outdir = tempfile.mkdtemp(prefix="cudecomp_perf_")
    env = {"CUDECOMP_ENABLE_PERFORMANCE_REPORT": "1", "CUDECOMP_PERFORMANCE_REPORT_DETAIL": "2",
           "CUDECOMP_PERFORMANCE_REPORT_SAMPLES": "4", "CUDECOMP_PERFORMANCE_REPORT_WARMUP_SAMPLES": "1",
           "CUDECOMP_PERFORMANCE_REPORT_WRITE_DIR": outdir}
    args = {"gdims": (64, 64, 64), "pdims": (1, 1), "ac": K.ALL_AC, "kind": 1, "repeat": 4, "in_place": True}
    res = run_ranks(1, "tests.gpu_bodies", "perf_report", args, timeout=600, extra_env=env)
    assert res[0]["counters"]["rotations"] == 16, res[0]["counters"]
    files = res[0]["files"]
    name = [f for f in files if f.startswith("cudecomp-perf-report-transpose-aggregated-")]
    assert len(name) == 1, sorted(files)
    agg = files[name[0]].splitlines()
    head = [i for i, line in enumerate(agg) if not line.startswith("#")][0]
    rows = [line.rsplit(",", 7) for line in agg[head + 1:]]
    assert [r[0].split(",")[0] for r in rows] == ["TransposeXY", "TransposeYZ", "TransposeZY", "TransposeYX"]
    for r in rows:   # ..., inplace, managed, samples, total_ms, A2A_ms, local_ms, A2A_BW_GBps
        assert r[1] == "T" and r[2] == "F" and r[3] == "3" and float(r[4]) > 0 and float(r[6]) > 0, r
