"""HIP kernel parity, move by move: cudecompExtMove3D (rows / LDS-transpose / generic kernels of
cudecomp_amd/csrc/kernels_*.hip, dispatched by kernels.cc) against the numpy restatement in oracle/ on the same seeded inputs.
Bit-exact (pure data movement, tolerance 0)."""
import itertools

import numpy as np
import pytest
import torch

import cudecomp_amd as cd
from oracle import oracle as orc
from tests import gpu_util as G

pytestmark = pytest.mark.gpu


def run_move(es, extent, ss, ds, src_len, dst_len, src_off=0, dst_off=0, seed=0, expect_cls=None):
    src = G.random_payload(src_len, es, seed)
    dst0 = G.random_payload(dst_len, es, seed + 1)
    exp = dst0.copy()
    orc.move3d_reference(src, exp, extent, ss, ds, src_off, dst_off)
    # fast path, generic fallback, fast path with streaming access, window variant of the transposes (with and without
    # streaming) for every destination off the 64-byte grid; 4-byte elements: the tile-shape switches (they select other
    # tiles only in `make TUNING_VARIANTS=1` builds, the default build always uses 64 x 128); the diagnostic store policy /
    # tile walk of the shared-GPU hunt
    # 256: "the cells between consecutive destination rows are the move's" -> row copies onto rows off the 64-byte grid take the
    # dense walk (rows_dense_kernel) when the gap is a few cells; every byte of the destination is compared either way
    for force_generic in (0, 1, 2, 4, 6) + ((16, 32, 18, 34) if es == 4 else ()) + (64, 128 + 2, 256 + 4, 256 + 4 + 2):
        d_src, d_dst = G.to_device(src.view(np.uint8)), G.to_device(dst0.view(np.uint8))
        cls = cd.cudecompExtMove3D(d_src.data_ptr() + src_off * es, d_dst.data_ptr() + dst_off * es, es, extent, ss, ds,
                                   force_generic, G.stream_ptr())
        torch.cuda.synchronize()
        got = G.to_host(d_dst).view(exp.dtype)
        assert np.array_equal(got.view(np.uint8), exp.view(np.uint8)), (es, extent, ss, ds, force_generic, cls)
        if 0 in extent:
            assert cls == -1  # nothing launched
        elif force_generic == 1:
            assert cls == 2
        elif expect_cls is not None:
            assert cls == expect_cls, (cls, expect_cls, es, extent, ss, ds)


@pytest.mark.parametrize("es", [4, 8, 16])
def test_rows_contiguous_and_strided(es):
    # (w, h, d) blocks cut out of / written into larger pencils, aligned and unaligned
    for w, h, d, sp, dp, so, do in [(64, 7, 3, 80, 64, 0, 0), (128, 33, 5, 128, 128, 0, 0), (6, 10, 11, 12, 9, 1, 2),
                                    (2, 37, 9, 40, 2, 3, 0), (1, 5, 4, 9, 1, 0, 1), (1000, 3, 1, 1024, 1000, 8, 16),
                                    (513, 4, 4, 515, 600, 1, 1)]:
        ss, ds = (1, sp, sp * (h + 2)), (1, dp, dp * (h + 1))
        run_move(es, (w, h, d), ss, ds, so + ss[2] * d + 64, do + ds[2] * d + 64, so, do, seed=w,
                 expect_cls=0 if w > 1 else None)  # 1-element rows are a gather: generic kernel


@pytest.mark.parametrize("es", [4, 8, 16])
def test_transposes_all_permutations(es):
    # dense A x B x C block written in every output order, plus halo-padded variants
    for (a, b, c), pad in itertools.product([(64, 64, 3), (70, 66, 5), (9, 10, 11), (128, 12, 20), (16, 200, 2),
                                             (4, 4, 4), (130, 3, 67)], [0, 3]):
        ext_in = (a, b, c)
        sin = (1, a + pad, (a + pad) * (b + pad))
        for perm in itertools.permutations(range(3)):
            if perm == (0, 1, 2):
                continue
            # output memory position i holds input dim perm[i]
            eo = [ext_in[p] + (pad if i < 2 else 0) for i, p in enumerate(perm)]
            so = [1, eo[0], eo[0] * eo[1]]
            ds = [0, 0, 0]
            for i, p in enumerate(perm):
                ds[p] = so[i]
            run_move(es, ext_in, sin, ds, sin[2] * c + 8, so[2] * ext_in[perm[2]] + 8, seed=a * 7 + b)


@pytest.mark.parametrize("es", [4, 8, 16])
def test_transpose_vector_and_scalar_paths(es):
    vw = 16 // es
    # aligned everything -> vector lanes; odd offset / odd extent -> scalar lanes; both must agree with numpy
    for ei, ej, ek, off in [(128, 128, 2, 0), (128, 128, 2, 1), (127, 128, 2, 0), (128, 126 + (vw > 1), 3, 0),
                            (256, 64, 1, 0), (64, 256, 4, vw)]:
        sin = (1, ei + 2 * vw, (ei + 2 * vw) * ej)
        ds = (ej + 4 * vw, 1, (ej + 4 * vw) * ei)
        run_move(es, (ei, ej, ek), sin, ds, off + sin[2] * ek + 64, off + ds[2] * ek + 64, off, off, seed=ei + off,
                 expect_cls=1)


@pytest.mark.parametrize("es", [4, 8, 16])
def test_transpose_far_strided_destination_walk(es):
    """Forward hops of an axis-contiguous cycle: destination rows far apart (stride ei-independent, >= 8 source rows), their
    batch planes adjacent -- the tile walk then visits j, a run of batch planes, i (kernels.cc classify(), "far-strided
    destination").  Batch extents that hold runs of 32 / 16 / 4 planes and one that holds none, edge tiles included."""
    for ei, ej, ek in [(128, 64, 64), (192, 128, 48), (64, 64, 36), (64, 64, 30), (130, 64, 32), (128, 70, 32)]:
        sin = (1, ei, ei * ej)           # source (x, y, z)
        ds = (ej * ek, 1, ej)            # destination (y, z, x): x slowest
        run_move(es, (ei, ej, ek), sin, ds, ei * ej * ek + 8, ei * ej * ek + 8, seed=ei + ek, expect_cls=1)


def test_degenerate_and_gather_moves():
    for es in (4, 8, 16):
        # 1-element rows gathered with a large stride (halo faces along the fastest axis)
        run_move(es, (1, 50, 20), (1, 64, 64 * 52), (1, 1, 50), 64 * 52 * 20 + 8, 50 * 20 + 8, 2, 0)
        run_move(es, (50, 20, 1), (64, 64 * 52, 0), (1, 50, 0), 64 * 52 * 20 + 8, 50 * 20 + 8, 3, 1)
        run_move(es, (1, 1, 1), (0, 0, 0), (0, 0, 0), 4, 4)
        run_move(es, (5, 0, 3), (1, 5, 25), (1, 5, 25), 100, 100)  # empty: destination untouched


def test_large_move_exceeding_one_tile_row():
    # enough blocks to exercise the batched block-index decode on all three axes
    run_move(8, (520, 260, 6), (1, 520, 520 * 260), (260 * 6, 1, 260), 520 * 260 * 6, 520 * 260 * 6, seed=5,
             expect_cls=1)
    run_move(4, (1024, 300, 4), (1, 1100, 1100 * 300), (1, 1024, 1024 * 300), 1100 * 300 * 4, 1024 * 300 * 4, seed=6,
             expect_cls=0)


def test_random_moves_property_sweep():
    """Randomly drawn 3-D block moves (extents 1..150, any source / destination permutation, halo-style row and plane
    padding, arbitrary element offsets and so every alignment) through all kernel flavours against numpy: a sweep over
    the space the shape-class tests above only sample."""
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st

    @settings(max_examples=500, deadline=None, suppress_health_check=list(HealthCheck))
    @given(es=st.sampled_from([4, 8, 16]), ext=st.tuples(st.integers(1, 150), st.integers(1, 70), st.integers(1, 12)),
           sperm=st.permutations((0, 1, 2)), dperm=st.permutations((0, 1, 2)),
           spad=st.tuples(st.integers(0, 5), st.integers(0, 3)), dpad=st.tuples(st.integers(0, 5), st.integers(0, 3)),
           soff=st.integers(0, 9), doff=st.integers(0, 9), seed=st.integers(0, 1 << 20))
    def check(es, ext, sperm, dperm, spad, dpad, soff, doff, seed):
        def strides(perm, pad):
            # memory position i holds logical dim perm[i]; rows / planes padded like halo-carrying pencils
            shape = [ext[p] for p in perm]
            s_mem = [1, shape[0] + pad[0], (shape[0] + pad[0]) * (shape[1] + pad[1])]
            out = [0, 0, 0]
            for i, p in enumerate(perm):
                out[p] = s_mem[i]
            return out, s_mem[2] * shape[2]
        ss, slen = strides(sperm, spad)
        ds, dlen = strides(dperm, dpad)
        run_move(es, ext, ss, ds, soff + slen + 16, doff + dlen + 16, soff, doff, seed=seed)

    check()


@pytest.mark.parametrize("pdims,rank", [((2, 2), 0), ((2, 2), 3), ((1, 4), 2), ((3, 1), 1)])
@pytest.mark.parametrize("layout", ["default", "contiguous"])
@pytest.mark.parametrize("pipelined", [False, True])
def test_local_phases_of_a_multi_rank_plan_match_numpy(pdims, rank, layout, pipelined):
    """cudecompExtRunLocalPhases (the probe behind profiles/*_local_phases.json): the pack and unpack launches of one rank
    of a multi-rank grid, run on this GPU without the exchange, move exactly what the plan's moves say (numpy execution
    of the same moves, as the CPU plan tests do)."""
    import torch
    from tests.bodies import run_moves
    orders = {"contiguous": [(0, 1, 2), (1, 2, 0), (2, 0, 1)], "default": [(0, 1, 2)] * 3}[layout]
    gdims, es = (24, 20, 28), 8
    grid = cd.make_grid_spec(gdims, pdims, orders)
    nel = max(cd.cudecompExtPencilInfo(grid, rank, a).size for a in range(3))
    ws = max(cd.cudecompExtWorkspaceSizes(grid, rank, a, (0, 0, 0))[0] for a in range(3))
    rng = np.random.default_rng(7)
    for op in cd.OPS:
        plan = cd.cudecompExtPlanTranspose(grid, rank, op, pipelined=pipelined, symmetric_recv=True)
        host = [rng.integers(0, 2**62, size=n, dtype=np.int64) for n in (nel, nel, ws)]
        dev = [torch.from_numpy(b.copy()).cuda() for b in host]
        stream = torch.cuda.current_stream().cuda_stream
        for phase, n, moves in ((1, plan.n_pack, plan.pack), (2, plan.n_unpack, plan.unpack)):
            cd.cudecompExtRunLocalPhases(grid, rank, op, phase, dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), es,
                                         stream, pipelined=pipelined, symmetric_recv=True)
            torch.cuda.synchronize()
            run_moves(moves, n, host)
            for b in range(3):
                assert np.array_equal(dev[b].cpu().numpy(), host[b]), (op, phase, b)
