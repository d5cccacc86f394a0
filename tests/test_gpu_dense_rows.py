"""Row copies of whole rows onto halo-carrying pencils: rows_dense_kernel (cudecomp_amd/csrc/kernels_rows.hip) -- whole cache
lines across the row ends, the few halo / padding cells between consecutive rows read and written back unchanged.

Kernel level: cudecompExtMove3D with the "whole rows" word (flag 256) against the numpy restatement of a block move on the
same seeded inputs, EVERY byte of the destination compared (so a gap cell that comes back different, or a byte outside the
span that is touched, fails).  API level: transposes onto halo-carrying / padded pencils (reference semantics:
include/internal/transpose.h:830-895, the unpack copies; halo / padding cells are not the transpose's to change) on one
rank and on 2 x 2 ranks sharing the GPU, every byte of the output buffers compared.  Bit-exact (tolerance 0)."""
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
    assert cd.cudecompExtLastKernelName() == "rows_dense_kernel<16,1>"
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
    for failures in run_ranks(4, "tests.gpu_bodies", "many", {"jobs": jobs}, timeout=600):
        assert failures == []
