"""Row copies of whole rows onto halo-carrying pencils: rows_dense_kernel (cudecomp_amd/csrc/kernels_rows.hip) -- whole cache
lines across the row ends, the few halo / padding cells between consecutive rows read and written back unchanged.

Kernel level: cudecompExtMove3D with the "whole rows" word (flag 256) against the numpy restatement of a block move on the
same seeded inputs, EVERY byte of the destination compared (so a gap cell that comes back different, or a byte outside the
span that is touched, fails).  API level: transposes onto halo-carrying / padded pencils (reference semantics:
include/internal/transpose.h:830-895, the unpack copies; halo / padding cells are not the transpose's to change) on one
rank and on 2 x 2 ranks sharing the GPU, every byte of the output buffers compared.  Bit-exact (tolerance 0)."""
import os

import numpy as np
import pytest
import torch

import cudecomp_amd as cd
from oracle import oracle as orc
from tests import gpu_bodies as B
from tests import gpu_util as G
from tests.mp import run_ranks

pytestmark = pytest.mark.gpu

WHOLE, ALWAYS, STREAMING = 256, 4, 2


def dense_move(es, extent, ss, ds, src_len, dst_len, src_off, dst_off, seed, flags, expect_dense=True):
    src = G.random_payload(src_len, es, seed)
    dst0 = G.random_payload(dst_len, es, seed + 1)
    exp = dst0.copy()
    orc.move3d_reference(src, exp, extent, ss, ds, src_off, dst_off)
    d_src, d_dst = G.to_device(src.view(np.uint8)), G.to_device(dst0.view(np.uint8))
    cls = cd.cudecompExtMove3D(d_src.data_ptr() + src_off * es, d_dst.data_ptr() + dst_off * es, es, extent, ss, ds, flags,
                               G.stream_ptr())
    torch.cuda.synchronize()
    name = cd.cudecompExtLastKernelName()
    got = G.to_host(d_dst)
    assert cls == 0, (cls, name)
    assert name.startswith("rows_dense_kernel") == expect_dense, (name, es, extent, ss, ds, flags)
    assert np.array_equal(got, exp.view(np.uint8)), (name, es, extent, ss, ds, dst_off, flags,
                                                    np.nonzero(got != exp.view(np.uint8))[0][:8])


@pytest.mark.parametrize("es", [4, 8, 16])
def test_dense_rows_move_by_move(es):
    # (row elements, rows, planes, x gap in cells, extra rows between planes, source row padding, dst offset in elements)
    shapes = [(128, 9, 3, 2, 2, 0, 1), (128, 9, 3, 2, 0, 0, 1), (96, 40, 2, 4, 2, 3, 2), (1030, 33, 3, 2, 1, 0, 1),
              (2048, 70, 2, 3, 4, 0, 3), (257, 21, 4, 1, 0, 5, 0), (64 * 16 // es, 300, 1, 8, 0, 0, 5), (520, 5, 7, 6, 2, 1, 9)]
    for w, hgt, d, gap, prow, spad, doff in shapes:
        if w * es < 256 or gap * es * 8 > w * es:
            continue
        for swap in (False, True):
            sp = w + spad
            ss = [1, sp, sp * hgt]
            if swap:  # the two slower dims in the other order on the source side (a chunk in another wire order)
                ss = [1, sp * d, sp]
            ds = [1, w + gap, (w + gap) * (hgt + prow)]
            src_len, dst_len = sp * hgt * d + 64, doff + ds[2] * d + 64
            for flags in (WHOLE | ALWAYS, WHOLE | ALWAYS | STREAMING):
                # rows that happen to sit on the 64-byte grid need no special kernel at all
                aligned = (doff * es) % 64 == 0 and (ds[1] * es) % 64 == 0 and ((ds[2] * es) % 64 == 0 or d == 1)
                dense_move(es, (w, hgt, d), ss, ds, src_len, dst_len, 0, doff, seed=w + hgt, flags=flags, expect_dense=not aligned)
            # without the planner's word the cells between the rows are not the move's: the shifted kernel
            dense_move(es, (w, hgt, d), ss, ds, src_len, dst_len, 0, doff, seed=w, flags=ALWAYS, expect_dense=False)


def test_dense_rows_large_moves_take_it_by_themselves():
    # 1024-wide fp64 pencil with a halo of one cell, 48 MiB: streaming access and the dense walk without being asked
    w, hgt, d, es = 1024, 768, 8, 8
    ds = [1, w + 2, (w + 2) * (hgt + 2)]
    dense_move(es, (w, hgt, d), [1, w, w * hgt], ds, w * hgt * d + 16, 1 + ds[1] + ds[2] * d + 16, 0, 1 + ds[1], seed=3, flags=WHOLE)
    assert cd.cudecompExtLastKernelName() == "rows_dense_kernel<1>"
    # a plane of more than 4 GiB of span (row / offset arithmetic beyond 32 bits): one plane of 4.25 million rows of 1 KiB
    w, hgt, es = 256, 4250000, 4
    ds = [1, w + 2, 0]
    dense_move(es, (w, hgt, 1), [1, w, 0], ds, w * hgt + 16, 3 + ds[1] * hgt + 16, 0, 3, seed=4, flags=WHOLE)


def test_dense_rows_random_sweep():
    """Randomly drawn whole-row copies (any element size, row length, gap, alignment, plane padding, dim order on the source
    side), forced through the dense kernel at small sizes."""
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st

    @settings(max_examples=200, deadline=None, suppress_health_check=list(HealthCheck))
    @given(es=st.sampled_from([4, 8, 16]), wq=st.integers(16, 400), hgt=st.integers(2, 40), d=st.integers(1, 6),
           gap=st.integers(1, 8), prow=st.integers(0, 3), spad=st.integers(0, 4), doff=st.integers(0, 33), swap=st.booleans(),
           stream=st.booleans(), seed=st.integers(0, 1 << 20))
    def check(es, wq, hgt, d, gap, prow, spad, doff, swap, stream, seed):
        w = max(wq, 256 // es)
        if gap * 8 > w:
            gap = max(1, w // 8)
        sp = w + spad
        ss = [1, sp * d, sp] if swap else [1, sp, sp * hgt]
        ds = [1, w + gap, (w + gap) * (hgt + prow)]
        aligned = (doff * es) % 64 == 0 and (ds[1] * es) % 64 == 0 and ((ds[2] * es) % 64 == 0 or d == 1)
        dense_move(es, (w, hgt, d), ss, ds, sp * hgt * d + 64, doff + ds[2] * d + 64, 0, doff, seed=seed,
                   flags=WHOLE | ALWAYS | (STREAMING if stream else 0), expect_dense=not aligned)

    check()


DENSE = "rows_dense_kernel"
HALOS = [((0, 0, 0), (1, 1, 1), (2, 0, 1)), ((1, 0, 0), (1, 0, 0), (1, 0, 0))]
PADS = [((0, 0, 0), (0, 0, 0), (0, 0, 0)), ((1, 0, 0), (0, 2, 0), (3, 0, 1))]


@pytest.mark.parametrize("kind", [0, 1, 3])
def test_transposes_onto_halo_pencils_every_byte_single_rank(kind):
    # default layout on one rank: every hop is ONE row copy input -> output; with halos / padding on x it is the dense one
    for halos, pads in zip(HALOS, PADS):
        args = {"gdims": (192, 64, 48), "pdims": (1, 1), "kind": kind, "halos": list(halos), "pads": list(pads),
                "expect_kernel": {"XToY": DENSE, "YToZ": DENSE, "ZToY": DENSE if (halos[1][0] or pads[1][0]) else "rows_",
                                  "YToX": DENSE if (halos[0][0] or pads[0][0]) else "rows_"}}
        assert B.transpose_every_byte(0, 1, args) == []
    # axis-contiguous layouts (permutations: the window kernel's business) through the same every-byte check
    args = {"gdims": (96, 64, 48), "pdims": (1, 1), "kind": kind, "ac": (1, 1, 1), "halos": list(HALOS[0]), "pads": list(PADS[1])}
    assert B.transpose_every_byte(0, 1, args) == []
    # ... and at rows long enough for the whole-line permutation kernels: forward hops transpose_lines_kernel, inverse hops
    # transpose_rowlines_kernel (halos of one cell everywhere; fp32 rows hold fewer than two of its windows + a unit: window kernel)
    if kind != 0:
        args = {"gdims": (176, 200, 168), "pdims": (1, 1), "kind": kind, "ac": (1, 1, 1), "halos": [(1, 1, 1)] * 3, "pads": [(0, 0, 0)] * 3,
                "out_of_place": [True],
                "expect_kernel": {"XToY": "transpose_lines_kernel", "YToZ": "transpose_lines_kernel",
                                  "ZToY": "transpose_rowlines_kernel", "YToX": "transpose_rowlines_kernel"}}
        assert B.transpose_every_byte(0, 1, args) == []
        assert B.transpose_every_byte(0, 1, dict(args, out_of_place=[False], expect_kernel=None)) == []  # in place: staged through the workspace


@pytest.mark.parametrize("backend", [cd.TRANSPOSE_COMM_MPI_P2P, cd.TRANSPOSE_COMM_NVSHMEM_PL, cd.TRANSPOSE_COMM_NVSHMEM],
                         ids=["peer", "peer_staged", "peer_one_launch"])
def test_transposes_onto_halo_pencils_every_byte_four_ranks(backend):
    # 2 x 2 ranks sharing the GPU: the unpacks of X->Y, Y->Z and Z->Y write whole rows (chunks are slabs along y / z) -> dense;
    # Y->X cuts the rows themselves (chunks along x) -> the shifted kernel.  Ragged extents in a second job.
    jobs = []
    for gdims, (halos, pads) in (((256, 128, 96), (HALOS[0], PADS[0])), ((250, 130, 94), (HALOS[1], PADS[1]))):
        expect = {"XToY": DENSE, "YToZ": DENSE, "ZToY": DENSE} if backend != cd.TRANSPOSE_COMM_NVSHMEM_PL else None
        jobs.append({"fn": "transpose_every_byte", "id": "%dx%dx%d" % gdims,
                     "args": {"gdims": gdims, "pdims": (2, 2), "kind": 1, "halos": list(halos), "pads": list(pads),
                              "transpose_backend": backend, "expect_kernel": expect}})
    # a 1 x 4 slab grid in the axis-contiguous layout with halos everywhere: X <-> Y is LOCAL on every rank (whole-line permutation
    # kernels: lines forward, row lines back), Y <-> Z exchanges among the four
    expect = {"XToY": "transpose_lines_kernel", "YToX": "transpose_rowlines_kernel"} if backend != cd.TRANSPOSE_COMM_NVSHMEM_PL else None
    slab = {"gdims": (200, 176, 96), "pdims": (1, 4), "kind": 1, "ac": (1, 1, 1), "halos": [(1, 1, 1)] * 3, "pads": [(0, 0, 0)] * 3,
            "transpose_backend": backend}
    jobs.append({"fn": "transpose_every_byte", "id": "slab_1x4_contiguous", "args": dict(slab, out_of_place=[True], expect_kernel=expect)})
    jobs.append({"fn": "transpose_every_byte", "id": "slab_1x4_contiguous_in_place", "args": dict(slab, out_of_place=[False])})
    for failures in run_ranks(4, "tests.gpu_bodies", "many", {"jobs": jobs}, timeout=600):
        assert failures == []


# ---- permutations onto halo-carrying pencils whose consecutive batch planes are adjacent rows: transpose_lines_kernel ----------
LINES = "transpose_lines_kernel"
# The random sweeps of the two whole-line permutation kernels draw the SAME examples in every run by default (the suite is run
# with -x by the driver: a run must not depend on a seed); CUDECOMP_TEST_SWEEP_RANDOM=1 draws fresh ones, CUDECOMP_TEST_SWEEP_EXAMPLES=N
# more of them (profiles/r06_whole_line_kernels_random_sweeps.log: 2 x 3000 fresh examples, no failure).
SWEEP = dict(max_examples=int(os.environ.get("CUDECOMP_TEST_SWEEP_EXAMPLES", "150")), deadline=None,
             derandomize=not os.environ.get("CUDECOMP_TEST_SWEEP_RANDOM"))


def lines_move(es, ei, ej, ek, gap, extra_rows, slab_pad, spad, doff, seed, flags, expect_lines=True):
    """dst[doff + i*di + k*dk + j] = src[i + j*sj + k*sk]: source rows along i (pitch ei + spad), destination rows along j of
    pitch dk = ej + gap, consecutive k adjacent, slabs di = dk * (ek + extra_rows) + slab_pad apart; every byte compared."""
    sj = ei + spad
    sk = sj * ej
    dk = ej + gap
    di = dk * (ek + extra_rows) + slab_pad
    extent, ss, ds = (ei, ej, ek), (1, sj, sk), (di, 1, dk)
    src = G.random_payload(sk * ek + 64, es, seed)
    dst0 = G.random_payload(doff + di * ei + 64, es, seed + 1)
    exp = dst0.copy()
    orc.move3d_reference(src, exp, extent, ss, ds, 0, doff)
    d_src, d_dst = G.to_device(src.view(np.uint8)), G.to_device(dst0.view(np.uint8))
    cls = cd.cudecompExtMove3D(d_src.data_ptr(), d_dst.data_ptr() + doff * es, es, extent, ss, ds, flags, G.stream_ptr())
    torch.cuda.synchronize()
    name = cd.cudecompExtLastKernelName()
    got = G.to_host(d_dst)
    assert cls == 1, (cls, name)
    assert name.startswith(LINES) == expect_lines, (name, es, extent, ss, ds, flags)
    assert np.array_equal(got, exp.view(np.uint8)), (name, es, extent, ss, ds, doff, flags,
                                                    np.nonzero(got != exp.view(np.uint8))[0][:8] // es)


@pytest.mark.parametrize("es", [4, 8, 16])
def test_lines_kernel_move_by_move(es):
    # (ei, ej, ek, gap, extra rows per slab, slab padding, source row padding, dst offset): edge tiles along i and along the
    # linear positions, odd plane counts, one and several row ends per window, slabs whose phase differs from slab to slab
    shapes = [(64, 256, 3, 2, 2, 0, 0, 1), (70, 300, 5, 2, 0, 1, 0, 1), (128, 200, 7, 1, 1, 3, 2, 0), (33, 1026, 2, 4, 2, 0, 0, 5),
              (200, 160, 9, 6, 0, 0, 1, 3), (96, 513, 4, 3, 1, 7, 0, 2), (64, 2050, 2, 2, 2, 0, 0, 2), (130, 384, 11, 8, 0, 5, 3, 7)]
    for ei, ej, ek, gap, xr, sp, spad, doff in shapes:
        if gap * 8 > ej:
            continue
        for flags in (WHOLE | ALWAYS, WHOLE | ALWAYS | STREAMING):
            lines_move(es, ei, ej, ek, gap, xr, sp, spad, doff, seed=ei + ej + ek, flags=flags)
        # without the planner's word the gap cells are not the move's: the window kernel, same result
        lines_move(es, ei, ej, ek, gap, xr, sp, spad, doff, seed=ej, flags=ALWAYS, expect_lines=False)


def test_lines_kernel_large_moves_take_it_by_themselves():
    # 1024-wide fp64 rows with a halo of one cell, 64 slabs x 48 planes = 24 MiB + a run walk with a ragged last run
    lines_move(8, 64, 1024, 48, 2, 2, 0, 0, 1 + 1026, seed=5, flags=WHOLE)
    assert cd.cudecompExtLastKernelName() == "transpose_lines_kernel<8,2,64,64,0,128>"
    lines_move(8, 256, 2048, 36, 4, 4, 0, 0, 2 + 2 * 2052, seed=6, flags=WHOLE | STREAMING)
    assert cd.cudecompExtLastKernelName() == "transpose_lines_kernel<8,2,64,64,4,128>"


def test_lines_kernel_random_sweep():
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st

    @settings(suppress_health_check=list(HealthCheck), **SWEEP)
    @given(es=st.sampled_from([4, 8, 16]), ei=st.integers(4, 200), ejq=st.integers(40, 700), ek=st.integers(2, 9),
           gap=st.integers(1, 8), xr=st.integers(0, 3), sp=st.integers(0, 9), spad=st.integers(0, 3), doff=st.integers(0, 40),
           stream=st.booleans(), seed=st.integers(0, 1 << 20))
    def check(es, ei, ejq, ek, gap, xr, sp, spad, doff, stream, seed):
        tj, u = {4: (128, 32), 8: (64, 16), 16: (32, 8)}[es]
        ej = max(ejq, tj + u, 8 * gap)   # (rows at least one window + one unit long: at most one row end per tile)
        dk, di = ej + gap, (ej + gap) * (ek + xr) + sp
        aligned = (doff * es) % 64 == 0 and (dk * es) % 64 == 0 and (di * es) % 64 == 0
        if aligned:
            doff += 1
        lines_move(es, ei, ej, ek, gap, xr, sp, spad, doff, seed, WHOLE | ALWAYS | (STREAMING if stream else 0))

    check()


def test_preserve_output_halos_switch_keeps_concurrent_halo_writes(tmp_path):
    """CUDECOMP_PRESERVE_OUTPUT_HALOS=1: a transpose must then never touch the halo / padding cells of its output (the
    reference's behaviour, include/internal/transpose.h:830-895), so a kernel of the caller that writes those cells on ANOTHER
    stream while the transpose runs keeps its values.  Without the switch the whole-line kernels run (and are named), with it
    the shifted / window kernels; both give the right interior."""
    import json
    import subprocess
    import sys
    from tests.mp import ROOT
    code = r'''
import json, os, sys
import numpy as np, torch
import cudecomp_amd as cd
torch.cuda.set_device(0)
h = cd.cudecompInit()
out = {}
for ac in ((0, 0, 0), (1, 1, 1)):
    gdims, halo = (512, 256, 128), (1, 1, 1)
    gd = cd.cudecompGridDescCreate(h, cd.make_config(gdims, (1, 1), axis_contiguous=ac))
    px, py = cd.cudecompGetPencilInfo(h, gd, 0, halo), cd.cudecompGetPencilInfo(h, gd, 1, halo)
    a = torch.arange(px.size, dtype=torch.float64, device="cuda")
    b = torch.full((py.size,), -5.0, dtype=torch.float64, device="cuda")
    work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * 8)
    # interior mask of the Y pencil (memory order of py): halo cells are everything else
    shape = [py.shape[2], py.shape[1], py.shape[0]]
    m = torch.zeros(shape, dtype=torch.bool, device="cuda")
    m[1:-1, 1:-1, 1:-1] = True
    m = m.reshape(-1)
    side = torch.cuda.Stream()
    st = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    for rep in range(6):
        cd.cudecompTranspose("XToY", h, gd, a.data_ptr(), b.data_ptr(), work, cd.DOUBLE, halo, halo, None, None, st)
        with torch.cuda.stream(side):   # the caller's own halo fill, concurrently
            b.masked_fill_(~m, 100.0 + rep)
    torch.cuda.synchronize()
    out[str(ac)] = {"kernel": cd.cudecompExtLastKernelName(), "halo_ok": bool((b[~m] == 105.0).all()),
                    "interior_ok": bool((b[m] >= 0).all())}
    cd.cudecompFree(h, gd, work); cd.cudecompGridDescDestroy(h, gd)
cd.cudecompFinalize(h)
print("RESULT " + json.dumps(out))
'''
    res = {}
    for switch in ("0", "1"):
        env = dict(os.environ, PYTHONPATH=ROOT, CUDECOMP_PRESERVE_OUTPUT_HALOS=switch)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        res[switch] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["0"]["(0, 0, 0)"]["kernel"].startswith("rows_dense_kernel"), res
    assert res["0"]["(1, 1, 1)"]["kernel"].startswith(LINES), res
    assert res["1"]["(0, 0, 0)"]["kernel"].startswith("rows_shifted_kernel"), res
    assert res["1"]["(1, 1, 1)"]["kernel"].startswith("transpose_window_kernel"), res
    for ac in ("(0, 0, 0)", "(1, 1, 1)"):
        assert res["1"][ac]["halo_ok"] and res["1"][ac]["interior_ok"], res
        assert res["0"][ac]["interior_ok"], res


# ---- permutations onto halo-carrying pencils whose adjacent rows are the tile's OWN rows: transpose_rowlines_kernel -------------
ROWLINES = "transpose_rowlines_kernel"


def rowlines_move(es, ei, ej, ek, gap, extra_rows, plane_pad, spad, doff, seed, flags, expect=True, src_order=0):
    """dst[doff + i*di + k*dk + j] = src[i + j*sj + k*sk]: source rows along i, destination rows along j of pitch di = ej + gap with
    consecutive i ADJACENT, planes dk = di * (ei + extra_rows) + plane_pad apart; every byte of the destination compared."""
    si = ei + spad
    if src_order == 0:   # source (i, k, j): planes near, rows far (the inverse hops of the cycle)
        sk, sj = si, si * ek
    else:                # source (i, j, k)
        sj, sk = si, si * ej
    di = ej + gap
    dk = di * (ei + extra_rows) + plane_pad
    extent, ss, ds = (ei, ej, ek), (1, sj, sk), (di, 1, dk)
    src = G.random_payload(si * ej * ek + 64, es, seed)
    dst0 = G.random_payload(doff + dk * ek + 64, es, seed + 1)
    exp = dst0.copy()
    orc.move3d_reference(src, exp, extent, ss, ds, 0, doff)
    d_src, d_dst = G.to_device(src.view(np.uint8)), G.to_device(dst0.view(np.uint8))
    cls = cd.cudecompExtMove3D(d_src.data_ptr(), d_dst.data_ptr() + doff * es, es, extent, ss, ds, flags, G.stream_ptr())
    torch.cuda.synchronize()
    name = cd.cudecompExtLastKernelName()
    got = G.to_host(d_dst)
    assert cls == 1, (cls, name)
    assert name.startswith(ROWLINES) == expect, (name, es, extent, ss, ds, flags)
    assert np.array_equal(got, exp.view(np.uint8)), (name, es, extent, ss, ds, doff, flags,
                                                    np.nonzero(got != exp.view(np.uint8))[0][:8] // es)


@pytest.mark.parametrize("es", [4, 8, 16])
def test_rowlines_kernel_move_by_move(es):
    tj, u = {4: (128, 32), 8: (64, 16), 16: (32, 8)}[es]
    long_row = 2 * (tj + u) + 1
    # (ei, ej, ek, gap, extra rows per plane, plane padding, source row padding, dst offset): whole tiles and ragged ones along i,
    # one to several windows beyond the minimum row length, several planes, rows whose phases differ from row to row
    shapes = [(64, long_row, 3, 2, 2, 0, 0, 1), (128, long_row + 37, 2, 2, 0, 1, 0, 1), (70, 1026, 2, 4, 2, 0, 1, 5), (4, long_row + 5, 5, 1, 0, 3, 0, 0),
              (65, 513 + long_row, 1, 6, 0, 0, 2, 3), (200, long_row + 64, 4, 3, 1, 7, 0, 2), (33, 2050, 2, 2, 2, 0, 0, 2), (129, long_row + 130, 3, 8, 0, 5, 3, 7)]
    for ei, ej, ek, gap, xr, pp, spad, doff in shapes:
        if gap * 8 > ej:
            continue
        for order in (0, 1):
            for flags in (WHOLE | ALWAYS, WHOLE | ALWAYS | STREAMING):
                rowlines_move(es, ei, ej, ek, gap, xr, pp, spad, doff, seed=ei + ej + ek, flags=flags, src_order=order)
        # without the planner's word the gap cells are not the move's: the window kernel, same result
        rowlines_move(es, ei, ej, ek, gap, xr, pp, spad, doff, seed=ej, flags=ALWAYS, expect=False)


def test_rowlines_kernel_large_moves_take_it_by_themselves():
    # a 1024-wide fp64 pencil with a halo of one cell: 1024 rows x 24 planes = 192 MiB
    rowlines_move(8, 1024, 1024, 24, 2, 2, 0, 0, 1 + 1026, seed=5, flags=WHOLE)
    assert cd.cudecompExtLastKernelName() == "transpose_rowlines_kernel<8,2,64,64,4,128>"


def test_rowlines_kernel_random_sweep():
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st

    @settings(suppress_health_check=list(HealthCheck), **SWEEP)
    @given(es=st.sampled_from([4, 8, 16]), ei=st.integers(4, 200), ejq=st.integers(0, 500), ek=st.integers(1, 5),
           gap=st.integers(1, 8), xr=st.integers(0, 3), pp=st.integers(0, 9), spad=st.integers(0, 3), doff=st.integers(0, 40),
           stream=st.booleans(), order=st.integers(0, 1), seed=st.integers(0, 1 << 20))
    def check(es, ei, ejq, ek, gap, xr, pp, spad, doff, stream, order, seed):
        tj, u = {4: (128, 32), 8: (64, 16), 16: (32, 8)}[es]
        ej = max(2 * (tj + u) + 1 + ejq, 8 * gap)
        di = ej + gap
        dk = di * (ei + xr) + pp
        aligned = (doff * es) % 64 == 0 and (di * es) % 64 == 0 and ((dk * es) % 64 == 0 or ek == 1)
        if aligned:  # (rows on the 64-byte grid need none of the special kernels)
            doff += 1
        rowlines_move(es, ei, ej, ek, gap, xr, pp, spad, doff, seed, WHOLE | ALWAYS | (STREAMING if stream else 0), src_order=order)

    check()
