"""Planning logic of the kernel layer on the CPU (no GPU, no launch): cudecompExtDescribeMove says how csrc/kernels.cc would
run a move -- class, tile, tile walk, access mode.  The tile walk is re-stated here from the decode of
csrc/kernels_tile.h (transpose_kernel) and must visit every tile exactly once for whatever run length the planner picks; the
round-5 walks (runs of j tiles / of batch planes for far-strided destinations, taller tiles for far-strided sources) must be
chosen for the bench shapes and only for large line-aligned moves."""
import random

import numpy as np

import cudecomp_amd as cd

SRC, DST = 1 << 32, 1 << 36  # line-aligned "addresses"


def walk(d, blocks=None):
    """(bi, bj, k) of every workgroup, as the kernel decodes them."""
    ti_n, tj_n, batch, run, bits = d["tiles_i"], d["tiles_j"], d["batch"], d["run"], d["walk"]
    nb = ti_n * tj_n * batch if blocks is None else blocks
    out = np.empty((nb, 3), dtype=np.int64)
    per = nb >> 3
    for lb in range(nb):
        lt = lb
        if bits & 1 and lb < (per << 3):
            lt = (lb & 7) * per + (lb >> 3)
        if bits & 2:
            if run > 1 and not bits & 4:
                jlo, rest = lt % run, lt // run
                bi, rest = rest % ti_n, rest // ti_n
                runs = tj_n // run
                bj, rest = (rest % runs) * run + jlo, rest // runs
            elif run > 1:
                bj, rest = lt % tj_n, lt // tj_n
                klo, rest = rest % run, rest // run
                bi, rest = rest % ti_n, (rest // ti_n) * run + klo
            else:
                bj, rest = lt % tj_n, lt // tj_n
                bi, rest = rest % ti_n, rest // ti_n
        else:
            bi, rest = lt % ti_n, lt // ti_n
            bj, rest = rest % tj_n, rest // tj_n
        out[lb] = (bi, bj, rest)
    return out


def check_bijection(d):
    w = walk(d)
    assert (w[:, 0] < d["tiles_i"]).all() and (w[:, 1] < d["tiles_j"]).all() and (w[:, 2] < d["batch"]).all(), d
    key = (w[:, 2] * d["tiles_j"] + w[:, 1]) * d["tiles_i"] + w[:, 0]
    assert len(np.unique(key)) == len(key) == d["tiles_i"] * d["tiles_j"] * d["batch"], d


def test_bench_shapes_take_the_round5_walks_and_tiles():
    N = 1024
    fwd = cd.cudecompExtDescribeMove(SRC, DST, 8, (N, N, N), (1, N, N * N), (N * N, 1, N))     # X->Y of the axis-contiguous cycle
    bwd = cd.cudecompExtDescribeMove(SRC, DST, 8, (N, N, N), (1, N * N, N), (N, 1, N * N))     # Y->X
    assert (fwd["cls"], fwd["tile_i"], fwd["tile_j"], fwd["access"]) == (1, 64, 64, 2)
    assert fwd["walk"] & 2 and not fwd["walk"] & 4 and fwd["run"] * 64 * 8 == 256 << 10       # runs of 256 KiB per destination row
    assert fwd["tiles_j"] % fwd["run"] == 0
    assert (bwd["tile_i"], bwd["tile_j"], bwd["run"], bwd["access"]) == (64, 128, 0, 2)         # taller tiles, no runs
    c128 = cd.cudecompExtDescribeMove(SRC, DST, 16, (N, N, 512), (1, N, N * N), (N * 512, 1, N))
    assert (c128["tile_i"], c128["tile_j"]) == (32, 32) and c128["run"] * 32 * 16 == 256 << 10
    c128b = cd.cudecompExtDescribeMove(SRC, DST, 16, (N, N, 512), (1, N * 512, N), (N, 1, N * N))
    assert (c128b["tile_i"], c128b["tile_j"], c128b["run"]) == (32, 64, 0)
    f32 = cd.cudecompExtDescribeMove(SRC, DST, 4, (2048, N, N), (1, 2048, 2048 * N), (N * N, 1, N))
    assert (f32["tile_i"], f32["tile_j"]) == (64, 128) and f32["run"] * 128 * 4 == 256 << 10
    # forcing a walk (tuning switch) switches the runs off; small moves (cached access) never take them
    assert cd.cudecompExtDescribeMove(SRC, DST, 8, (N, N, N), (1, N, N * N), (N * N, 1, N), flags=128)["run"] == 0
    small = cd.cudecompExtDescribeMove(SRC, DST, 8, (128, 128, 64), (1, 128, 128 * 128), (128 * 64, 1, 128))
    assert small["access"] == 0 and small["run"] == 0 and small["tile_j"] == 64
    # misaligned destinations go to the window kernel / cached stores, never to the run walk
    mis = cd.cudecompExtDescribeMove(SRC, DST + 8, 8, (N, N, N), (1, N, N * N), (N * N + 2, 1, N + 2), flags=0)
    assert mis["run"] == 0


def test_every_walk_visits_every_tile_once():
    rng = random.Random(5)
    seen_runs = {"j": 0, "k": 0, "none": 0}
    for _ in range(300):
        es = rng.choice([4, 8, 16])
        fused = rng.random() < 0.5
        ei = rng.choice([64, 128, 192, 320, 1000])
        ej = rng.choice([64, 128, 256, 384, 1000])
        ek = rng.choice([4, 8, 12, 30, 32, 36, 48, 64])
        if fused:   # destination (j, k, i) dense: the planner fuses j and k into one long dim
            ss, ds = (1, ei, ei * ej), (ej * ek, 1, ej)
        else:       # padded planes in the destination keep k a dim of its own
            ss, ds = (1, ei, ei * ej), ((ej + 16) * ek, 1, ej + 16)
        d = cd.cudecompExtDescribeMove(SRC, DST, es, (ei, ej, ek), ss, ds, flags=2)  # streaming access regardless of the size
        if d["cls"] != 1:
            continue
        if d["run"] > 1:
            if d["walk"] & 4:
                assert d["batch"] % d["run"] == 0, d
                seen_runs["k"] += 1
            else:
                assert d["tiles_j"] % d["run"] == 0, d
                seen_runs["j"] += 1
        else:
            seen_runs["none"] += 1
        if d["tiles_i"] * d["tiles_j"] * d["batch"] <= 40000:
            check_bijection(d)
    assert seen_runs["k"] > 10 and seen_runs["none"] > 10, seen_runs
    # long fused destination rows: runs of j tiles (a divisor of the tile count near 256 KiB / tile width)
    for es, ei, ej, ek in [(8, 128, 1024, 64), (8, 64, 2048, 48), (16, 64, 1024, 40), (4, 128, 4096, 40), (8, 192, 1000, 72), (8, 64, 1536, 50)]:
        d = cd.cudecompExtDescribeMove(SRC, DST, es, (ei, ej, ek), (1, ei, ei * ej), (ej * ek, 1, ej), flags=2)
        assert d["cls"] == 1 and d["batch"] == 1, d          # (j, k) fused
        assert d["run"] > 1 and not d["walk"] & 4 and d["tiles_j"] % d["run"] == 0, d
        assert 64 << 10 <= d["run"] * d["tile_j"] * es <= 256 << 10, d
        check_bijection(d)


def test_run_walks_by_hand():
    for ti_n, tj_n, batch, run, bits in [(3, 8, 2, 4, 3), (5, 12, 1, 6, 3), (2, 3, 8, 4, 7), (4, 5, 6, 2, 7), (3, 7, 5, 0, 3), (3, 7, 5, 0, 1),
                                         (16, 64, 1, 8, 2)]:
        check_bijection({"tiles_i": ti_n, "tiles_j": tj_n, "batch": batch, "run": run, "walk": bits})


# ---- row copies of whole rows onto halo-carrying pencils: the dense walk (rows_dense_kernel, csrc/kernels_rows.hip) -----------
WHOLE = 256  # cudecompExtDescribeMove / cudecompExtMove3D flag: the cells between consecutive destination rows are the move's


def test_dense_row_copy_is_chosen_only_where_it_is_safe_and_pays():
    nx, ny, nz = 1024, 128, 4  # fp64: 4 MiB
    src_st = (1, nx, nx * ny)
    halo_st = (1, nx + 2, (nx + 2) * (ny + 2))  # a halo of one cell on x and y: rows 8 B off the 64-byte grid, 16-byte gaps
    d = cd.cudecompExtDescribeMove(SRC, DST + 8, 8, (nx, ny, nz), src_st, halo_st, flags=WHOLE)
    assert d["cls"] == 0 and d["tile_i"] == 2 and d["variant"] == 16, d
    span = (ny - 1) * (nx + 2) * 8 + nx * 8
    assert d["batch"] == nz and d["tiles_i"] == -(-(span + 63) // 16384) and d["tiles_j"] == 1, d
    # without the planner's word: the shifted kernel (partial lines at the row ends)
    assert cd.cudecompExtDescribeMove(SRC, DST + 8, 8, (nx, ny, nz), src_st, halo_st)["tile_i"] == 1
    # small moves keep the plain kernel unless asked (flag 4), as for the shifted kernel
    assert cd.cudecompExtDescribeMove(SRC, DST + 8, 8, (nx, 8, 2), src_st, halo_st, flags=WHOLE)["tile_i"] == 0
    assert cd.cudecompExtDescribeMove(SRC, DST + 8, 8, (nx, 8, 2), src_st, halo_st, flags=WHOLE | 4)["tile_i"] == 2
    # gaps wider than a few cells (a slab out of a wider pencil) are not rewritten
    wide = (1, 2 * nx, 2 * nx * ny)
    assert cd.cudecompExtDescribeMove(SRC, DST + 8, 8, (nx, ny, nz), src_st, wide, flags=WHOLE)["tile_i"] == 1
    # one row per plane: the next row of the MOVE is a plane away, other rows lie in between
    one = cd.cudecompExtDescribeMove(SRC, DST + 8, 8, (nx, 1, 512), (1, nx, nx), halo_st, flags=WHOLE)
    assert one["tile_i"] == 1, one
    # line-aligned rows (the gap is whole lines) need neither
    aligned = (1, nx + 16, (nx + 16) * ny)
    assert cd.cudecompExtDescribeMove(SRC, DST, 8, (nx, ny, nz), src_st, aligned, flags=WHOLE)["tile_i"] == 0
    # rows that fuse with the next dim (no gap on x, halo rows on y only) are long rows, not "whole rows"
    yhalo = (1, nx, nx * (ny + 2))
    f = cd.cudecompExtDescribeMove(SRC, DST + 8, 8, (nx, ny, nz), src_st, yhalo, flags=WHOLE)
    assert f["cls"] == 0 and f["tile_i"] == 1, f
    # the two slower dims in the other order on the source side (a chunk in another wire order): dim 1 is still the one that
    # steps by the row pitch
    swapped = cd.cudecompExtDescribeMove(SRC, DST + 8, 8, (nx, ny, nz), (1, nx * nz, nx), halo_st, flags=WHOLE)
    assert swapped["tile_i"] == 2 and swapped["batch"] == nz, swapped
    # 4- and 16-byte elements alike
    for es in (4, 16):
        assert cd.cudecompExtDescribeMove(SRC, DST + es, es, (nx, ny, nz), src_st, halo_st, flags=WHOLE)["tile_i"] == 2


def dense_walk_reference(src, dst, row_bytes, rows, pitch, spitch, dst_base_address, per_block=16384):
    """numpy restatement of rows_dense_kernel for ONE plane, byte arrays: src row r at r*spitch, dst row r at
    dst_base_address-relative offset `d0` + r*pitch.  Returns the set of destination bytes written and performs the copy the
    way the lanes do (16-byte vectors on the 64-byte grid; gap bytes re-written with their own content; masked ends)."""
    d0 = dst_base_address
    shift = d0 & 63
    span = (rows - 1) * pitch + row_bytes
    blocks = -(-(span + 63) // per_block)
    written = np.zeros(dst.size, dtype=bool)
    out = dst.copy()
    for vec in range(blocks * per_block // 16):
        p = vec * 16 - shift
        if p >= span or p + 16 <= 0:
            continue
        for k in range(4):
            pp = p + 4 * k
            if pp < 0 or pp >= span:
                continue
            r, o = divmod(pp, pitch)
            val = src[r * spitch + o:r * spitch + o + 4] if o < row_bytes else dst[d0 + pp:d0 + pp + 4]
            out[d0 + pp:d0 + pp + 4] = val
            written[d0 + pp:d0 + pp + 4] = True
    return out, written


def test_dense_walk_restatement_copies_rows_and_leaves_everything_else():
    rng = np.random.default_rng(7)
    for _ in range(40):
        row_bytes = int(rng.integers(64, 130)) * 4
        gap = int(rng.integers(1, 9)) * 4
        rows = int(rng.integers(2, 9))
        pitch = row_bytes + gap
        spitch = row_bytes + int(rng.integers(0, 3)) * 4
        d0 = int(rng.integers(0, 40)) * 4
        src = rng.integers(0, 256, rows * spitch + 64, dtype=np.uint8)
        dst = rng.integers(0, 256, d0 + rows * pitch + 128, dtype=np.uint8)
        out, written = dense_walk_reference(src, dst, row_bytes, rows, pitch, spitch, d0, per_block=1024)
        exp = dst.copy()
        for r in range(rows):
            exp[d0 + r * pitch:d0 + r * pitch + row_bytes] = src[r * spitch:r * spitch + row_bytes]
        assert np.array_equal(out, exp)
        span = (rows - 1) * pitch + row_bytes
        assert written[d0:d0 + span].all() and not written[:d0].any() and not written[d0 + span:].any()


# ---- permutations onto halo-carrying pencils whose consecutive batch planes are adjacent rows: transpose_lines_kernel ----------
LINES = 8  # walk bit of cudecompExtDescribeMove


def lines_shape(nx, ny, nz, h, es=8, whole=True, dst_off=None):
    """X->Y hop of an axis-contiguous 1 x 1 cycle onto a Y pencil with a halo of h cells on every axis: source x fastest
    (dense), destination y fastest, rows of consecutive z planes adjacent, x slabs far apart."""
    py, pz = ny + 2 * h, nz + 2 * h
    off = h + py * (h + pz * h) if dst_off is None else dst_off
    return cd.cudecompExtDescribeMove(SRC, DST + off * es, es, (nx, ny, nz), (1, nx, nx * ny), (py * pz, 1, py),
                                      flags=WHOLE if whole else 0)


def test_lines_kernel_is_chosen_for_forward_hops_onto_halo_pencils_only():
    d = lines_shape(1024, 1024, 1022, 1)
    assert d["cls"] == 1 and d["walk"] & LINES and (d["tile_i"], d["tile_j"], d["access"]) == (64, 64, 4), d
    span = 1021 * 1026 + 1024
    assert d["tiles_i"] == 16 and d["tiles_j"] == -(-(span + 15) // 64) and d["run"] == 4 and d["walk"] >> 8 == 0, d  # 16 tile rows: one group, runs of 2 KiB
    c5 = lines_shape(2048, 2048, 256, 2)  # config 5's pencil shape on a 1 x 1 grid
    assert c5["walk"] & LINES and c5["variant"] == 2 and c5["tiles_i"] == 32 and c5["walk"] >> 8 == 16 and c5["run"] == 64, c5  # two groups: runs of 32 KiB
    for es in (4, 16):
        assert lines_shape(1024, 1024, 64, 1, es=es)["walk"] & LINES
    # without the planner's word the gap cells are not the move's: the window kernel
    w = lines_shape(1024, 1024, 1022, 1, whole=False)
    assert w["cls"] == 1 and not w["walk"] & LINES and w["access"] == 4, w
    # line-aligned rows need neither kernel
    a = cd.cudecompExtDescribeMove(SRC, DST, 8, (1024, 1024, 64), (1, 1024, 1 << 20), ((1024 + 16) * 64, 1, 1024 + 16), flags=WHOLE)
    assert not a["walk"] & LINES and a["access"] == 2, a
    # inverse hops (the tile's own rows are the adjacent ones, batch planes far apart) keep the window kernel
    inv = cd.cudecompExtDescribeMove(SRC, DST + 8, 8, (1024, 1024, 64), (1, 1024 * 64, 1024), (1026, 1, 1026 * 1026), flags=WHOLE)
    assert inv["cls"] == 1 and not inv["walk"] & LINES, inv
    # wide gaps (a slab out of a wider pencil) are not rewritten
    wide = cd.cudecompExtDescribeMove(SRC, DST + 8, 8, (1024, 1024, 64), (1, 1024, 1 << 20), (2048 * 70, 1, 2048), flags=WHOLE)
    assert not wide["walk"] & LINES, wide
    # small moves stay with the plain tile kernel unless asked (flag 4), like the window kernel
    assert not lines_shape(128, 128, 4, 1)["walk"] & LINES
    small = cd.cudecompExtDescribeMove(SRC, DST + 8, 8, (128, 128, 8), (1, 128, 128 * 128), (130 * 10, 1, 130), flags=WHOLE | 4)
    assert small["walk"] & LINES and small["run"] == 4, small
    # an odd row count along i: element-wise lanes (vectors hold whole elements along i only)
    assert lines_shape(1023, 1024, 64, 1)["variant"] == 1
    assert lines_shape(1024, 1023, 64, 1)["variant"] == 2   # the row length does not matter


def lines_walk_reference(ei, ej, ek, di, dk, dst_phase, es, ti, tj, run, group=0, ub=128):
    """numpy restatement of transpose_lines_kernel's decode: which (slab i, linear position l) every workgroup stores.
    Returns the coverage count per (i, l) over [0, L)."""
    U = ub // es
    L = (ek - 1) * dk + ej
    ti_n, tl_n = -(-ei // ti), -(-(L + U - 1) // tj)
    cover = np.zeros((ei, L), dtype=np.int32)
    nb = ti_n * tl_n
    per = nb >> 3
    seen = set()
    G = group if group else ti_n
    for lb in range(nb):
        lt = (lb & 7) * per + (lb >> 3) if lb < (per << 3) else lb
        g, x = divmod(lt, G * tl_n)
        gsize = G if (g + 1) * G <= ti_n else ti_n - g * G
        r = run if run > 0 else tl_n
        full_runs = tl_n // r
        full = full_runs * r * gsize
        if x < full:
            lo, rest = x % r, x // r
            bi, bl = rest % gsize, (rest // gsize) * r + lo
        else:
            tail, y = tl_n - full_runs * r, x - full
            bl, bi = full_runs * r + y % tail, y // tail
        bi += g * G
        assert (bi, bl) not in seen and bi < ti_n and bl < tl_n, (bi, bl, ti_n, tl_n, lt)
        seen.add((bi, bl))
        lb0 = bl * tj - (U - 1)
        for i in range(bi * ti, min(ei, bi * ti + ti)):
            ph = (dst_phase + i * di) % U
            lo_l = lb0 + (U - 1) - ph
            assert (dst_phase + i * di + lo_l) % U == 0  # every window starts on a unit boundary
            a, b = max(lo_l, 0), min(lo_l + tj, L)
            if b > a:
                cover[i, a:b] += 1
    assert len(seen) == nb
    return cover


def test_lines_walk_covers_every_cell_of_every_slab_exactly_once():
    rng = random.Random(11)
    for _ in range(60):
        es = rng.choice([4, 8, 16])
        ti, tj = {4: (64, 128), 8: (64, 64), 16: (32, 32)}[es]
        ei, ej, ek = rng.choice([32, 64, 100, 130, 200, 330]), rng.choice([160, 200, 257, 300]), rng.choice([2, 3, 5, 9])
        gap = rng.choice([1, 2, 3, 4, 6])
        dk = ej + gap
        di = dk * (ek + rng.choice([0, 1, 2])) + rng.choice([0, 1, 5])
        run, group = rng.choice([0, 1, 2, 3, 7]), rng.choice([0, 0, 1, 2, 3])
        cover = lines_walk_reference(ei, ej, ek, di, dk, rng.randrange(0, 64), es, ti, tj, run, group)
        assert (cover == 1).all(), (es, ei, ej, ek, gap, di, run, group)


# ---- single-rank in-place transposes of cubic grids: the in-place rotation (csrc/kernels_rotate.hip) ---------------------------
def _orders(ac):
    return [[(ax + i) % 3 if ac[ax] else i for i in range(3)] for ax in range(3)]


def test_planner_offers_the_in_place_rotation_only_where_it_is_one():
    cube = cd.make_grid_spec((64, 64, 64), (1, 1), _orders((1, 1, 1)))
    want = {"XToY": 1, "YToZ": 1, "ZToY": -1, "YToX": -1}
    for op in cd.OPS:
        p = cd.cudecompExtPlanTranspose(cube, 0, op, inplace=True)
        assert p.rotate == want[op] and p.n_pack == 1 and p.n_unpack == 1, (op, p.rotate)   # the staged form stays in the plan
        assert cd.cudecompExtPlanTranspose(cube, 0, op, inplace=False).rotate == 0
        assert cd.cudecompExtPlanTranspose(cube, 0, op, (1, 0, 0), (1, 0, 0), inplace=True).rotate == 0       # halos
        assert cd.cudecompExtPlanTranspose(cube, 0, op, None, None, (0, 1, 0), (0, 1, 0), inplace=True).rotate == 0  # padding
    # not cubic / default layout (same order everywhere: in place is a no-op or a staged copy, never a rotation)
    for spec in (cd.make_grid_spec((64, 64, 32), (1, 1), _orders((1, 1, 1))), cd.make_grid_spec((64, 64, 64), (1, 1), _orders((0, 0, 0)))):
        for op in cd.OPS:
            assert cd.cudecompExtPlanTranspose(spec, 0, op, inplace=True).rotate == 0, (list(spec.gdims), op)
    # mixed layouts: the hops between two orders that ARE a rotation of each other still qualify (X and Y x-fastest, Z z-fastest)
    mixed = cd.make_grid_spec((64, 64, 64), (1, 1), _orders((1, 0, 1)))
    assert [cd.cudecompExtPlanTranspose(mixed, 0, op, inplace=True).rotate for op in cd.OPS] == [0, -1, 1, 0]
    two = cd.make_grid_spec((64, 64, 64), (2, 1), _orders((1, 1, 1)))
    assert all(cd.cudecompExtPlanTranspose(two, r, op, inplace=True).rotate == 0 for r in (0, 1) for op in cd.OPS)


def test_rotation_restatement_closes_every_orbit():
    """numpy restatement of rotate_kernel's decomposition (owner = smallest of the three block triples, three tiles loaded,
    three stored, in-tile transposition of the linear index space) on a small cube, both directions, against the index map
    of the transposes it replaces: out[y + N*(z + N*x)] = in[x + N*(y + N*z)]."""
    n, t = 12, 4
    nb = n // t
    a = np.arange(n ** 3, dtype=np.int64)
    old = a.reshape(n, n, n)   # [p2][p1][p0]
    for fwd in (True, False):
        new = np.full_like(old, -1)
        owners = 0
        for b2 in range(nb):
            for b1 in range(nb):
                for b0 in range(nb):
                    keys = [((b2 * nb + b1) * nb + b0), ((b1 * nb + b0) * nb + b2), ((b0 * nb + b2) * nb + b1)]
                    if keys[0] > keys[1] or keys[0] > keys[2]:
                        continue
                    owners += 1
                    orbit = [(b0, b1, b2), (b2, b0, b1), (b1, b2, b0)] if fwd else [(b0, b1, b2), (b1, b2, b0), (b2, b0, b1)]
                    nt = 1 if b0 == b1 == b2 else 3
                    tiles = [old[c2 * t:(c2 + 1) * t, c1 * t:(c1 + 1) * t, c0 * t:(c0 + 1) * t].copy() for c0, c1, c2 in orbit[:nt]]
                    for k in range(nt):
                        src = tiles[(k + 1) % nt].reshape(-1)          # old linear order of the source tile
                        if fwd:    # old linear = x + T*y -> new linear = y + T^2*x
                            out = src.reshape(t * t, t).T.reshape(-1)
                        else:      # old linear = y + T^2*x -> new linear = x + T*y
                            out = src.reshape(t, t * t).T.reshape(-1)
                        c0, c1, c2 = orbit[k]
                        assert (new[c2 * t:(c2 + 1) * t, c1 * t:(c1 + 1) * t, c0 * t:(c0 + 1) * t] == -1).all()
                        new[c2 * t:(c2 + 1) * t, c1 * t:(c1 + 1) * t, c0 * t:(c0 + 1) * t] = out.reshape(t, t, t)
        assert owners == (nb ** 3 - nb) // 3 + nb
        # forward: new[p2][p1][p0] = old at position (p2', p1', p0') = (p1, p0, p2) i.e. new[p0,p1,p2] = old[p2,p0,p1]
        exp = old.transpose(1, 0, 2) if False else None
        p2, p1, p0 = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
        if fwd:
            exp = old[p1, p0, p2]   # old[(q0,q1,q2) = (p2,p0,p1)] -> index [q2][q1][q0] = [p1][p0][p2]
        else:
            exp = old[p0, p2, p1]   # old[(q0,q1,q2) = (p1,p2,p0)] -> index [p0][p2][p1]
        assert np.array_equal(new, exp), fwd
        # ... and that IS the transpose: X pencil (x,y,z) -> Y pencil (y,z,x): out[y + N*(z + N*x)] = in[x + N*(y + N*z)]
        if fwd:
            x, y, z = 3, 7, 10
            assert new.reshape(-1)[y + n * (z + n * x)] == a[x + n * (y + n * z)]


def test_rotation_walks_visit_every_block_triple_once():
    """The orbit walk of rotate_kernel (csrc/rotate_walk.h, the code the kernel runs, through cudecompExtRotateWalk): for every array
    size and walk -- the default per-XCD walk with its shears, the cube walks of tuning builds, more shears -- every block triple
    is taken by exactly one workgroup, the rest of the grid maps to none, and the padding stays small.  On the default walk every
    XCD (workgroup % 8) steps through ALL p0 blocks of all three tiles of its orbits within any nb consecutive workgroups of
    its own: what the walk is for (profiles/r06_tuning.md section 8)."""
    default = 15 | 1 << 4 | 3 << 8 | 2 << 12
    walks = [-1, default, 15, 0, 1, 2, 3, 5, 1 | 2 << 4 | 3 << 8 | 3 << 12, 15 | 15 << 4 | 15 << 8 | 15 << 12, 2 | 7 << 8]
    for nb in (1, 2, 3, 4, 5, 7, 8, 9, 13, 16, 24, 33, 64):
        for walk in walks:
            grid, blocks = cd.cudecompExtRotateWalk(nb, walk)
            assert blocks.shape == (grid, 3)
            valid = blocks[blocks[:, 0] >= 0]
            assert ((blocks >= 0).all(axis=1) | (blocks == -1).all(axis=1)).all()
            assert len(valid) == nb ** 3 and (valid < nb).all(), (nb, walk)
            keys = (valid[:, 2].astype(np.int64) * nb + valid[:, 1]) * nb + valid[:, 0]
            assert len(np.unique(keys)) == nb ** 3, (nb, walk)
            assert grid <= (nb + 31) ** 3 and (walk not in (-1, default) or grid <= nb ** 3 + 8 * nb), (nb, walk, grid)
    for nb in (16, 64):
        grid, blocks = cd.cudecompExtRotateWalk(nb)
        assert grid == nb ** 3 and np.array_equal(blocks, cd.cudecompExtRotateWalk(nb, default)[1])   # (the pinned default)
        for x in range(8):
            mine = blocks[x::8]
            for start in (0, nb, 5 * nb, len(mine) - nb):   # (windows of one value of s // nb)
                window = mine[start:start + nb]
                for col in range(3):   # p0 blocks of the three tiles of an orbit: b0, b2, b1
                    assert len(np.unique(window[:, col])) == nb, (nb, x, start, col)


# ---- permutations onto halo-carrying pencils whose adjacent rows are the tile's OWN rows: transpose_rowlines_kernel -------------
ROWLINES = 16  # walk bit


def test_rowlines_kernel_is_chosen_for_inverse_hops_onto_halo_pencils():
    n, h = 1024, 1
    p = n + 2 * h
    # Z->Y of the axis-contiguous cycle: source z fastest, destination y fastest, rows of consecutive z adjacent, x planes far apart
    d = cd.cudecompExtDescribeMove(SRC, DST + 8 * (h + p * (h + p * h)), 8, (n, n, n - 2), (1, n * n, n), (p, 1, p * p), flags=WHOLE)
    assert d["cls"] == 1 and d["walk"] & ROWLINES and not d["walk"] & LINES and d["access"] == 4, d
    assert d["tiles_i"] == 16 and d["tiles_j"] == (p - 1 + 15) // 64 + 1 and d["batch"] == n - 2, d
    # without the planner's word, with CUDECOMP_PRESERVE_OUTPUT_HALOS (flag 8), for wide gaps: the window kernel
    assert not cd.cudecompExtDescribeMove(SRC, DST + 8, 8, (n, n, 64), (1, n * 64, n), (p, 1, p * p))["walk"] & ROWLINES
    assert not cd.cudecompExtDescribeMove(SRC, DST + 8, 8, (n, n, 64), (1, n * 64, n), (p, 1, p * p), flags=WHOLE | 8)["walk"] & ROWLINES
    assert not cd.cudecompExtDescribeMove(SRC, DST + 8, 8, (n, n, 64), (1, n * 64, n), (2 * n, 1, 2 * n * n), flags=WHOLE)["walk"] & ROWLINES
    # short rows (less than two windows + a unit) keep the window kernel
    assert not cd.cudecompExtDescribeMove(SRC, DST + 8, 8, (n, 120, 64), (1, n * 64, n), (122, 1, 122 * (n + 2)), flags=WHOLE | 4)["walk"] & ROWLINES


def rowlines_reference(ei, ej, di, dst_phase, es, ti, tj, ub=128):
    """numpy restatement of transpose_rowlines_kernel's store rules for ONE plane: coverage count of the plane's global linear
    positions [0, (ei - 1) * di + ej) and of anything outside."""
    U = ub // es
    span = (ei - 1) * di + ej
    tw_n = (di - 1 + U - 1) // tj + 1
    cover = np.zeros(span + 4 * tj, dtype=np.int32)   # slack behind the span: must stay zero
    below = 0
    for i in range(ei):
        ph = (dst_phase + i * di) % U
        lo = 0
        if i > 0:
            php = (dst_phase + (i - 1) * di) % U
            lo = ((di - 1 + php) // tj + 1) * tj - php - di
        hi = ej if i == ei - 1 else ((di - 1 + ph) // tj + 1) * tj - ph
        for w in range(tw_n):
            a, b = w * tj - ph, w * tj - ph + tj
            assert (dst_phase + i * di + a) % U == 0
            a, b = max(a, lo), min(b, hi)
            if b > a:
                if i * di + a < 0:
                    below += 1
                cover[i * di + a:i * di + b] += 1
    return cover, span, below


def test_rowlines_windows_cover_every_cell_of_a_plane_exactly_once():
    rng = random.Random(13)
    for _ in range(200):
        es = rng.choice([4, 8, 16])
        ti, tj = {4: (64, 128), 8: (64, 64), 16: (32, 32)}[es]
        u = 128 // es
        ej = rng.choice([2 * (tj + u) + 1, 300, 513, 1026, 777])
        if ej <= 2 * (tj + u):
            continue
        gap = rng.choice([1, 2, 3, 4, 6])
        ei = rng.choice([2, 3, 17, 64, 65, 130])
        cover, span, below = rowlines_reference(ei, ej, ej + gap, rng.randrange(0, 64), es, ti, tj)
        assert below == 0 and (cover[:span] == 1).all() and (cover[span:] == 0).all(), (es, ei, ej, gap)
