"""All ranks in ONE process: the product's planner (`cudecompExtPlanTranspose` / `cudecompExtPlanHalo`, the same
`buildTransposePlan` / `buildHaloPlan` the executors use) is asked for the plan of every rank of a decomposition,
the plans are executed on host arrays with numpy and the exchange is simulated by copying chunks between the ranks'
buffers.  No GPU, no processes -- so hypothesis can draw hundreds of decompositions (ragged extents, uneven
splits, gdims_dist, both rank orders, arbitrary memory orders, halos, padding, in place, every transport trait)
and every one is checked against the reference's analytic oracle (interior of every output pencil for
transposes, the whole pencil for halo updates) and the oracle's restatement of the reference algorithm."""
import itertools

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import cudecomp_amd as cd
from oracle import oracle as orc
from tests.bodies import run_moves

PERMS = list(itertools.permutations((0, 1, 2)))
KIND = 1  # fp64 payload: global indices are exact
DT = orc.KINDS[KIND][0]


@st.composite
def decompositions(draw, max_ranks=12):
    pdims = draw(st.sampled_from([(a, b) for a in range(1, 7) for b in range(1, 7) if a * b <= max_ranks]))
    lo = max(pdims)  # no empty pencils: that case has its own test
    gdims = tuple(draw(st.integers(lo, lo + 9)) for _ in range(3))
    if draw(st.booleans()):
        mem_order = tuple(draw(st.sampled_from(PERMS)) for _ in range(3))
    else:
        ac = tuple(draw(st.booleans()) for _ in range(3))
        mem_order = tuple(tuple((ax + i) % 3 if ac[ax] else i for i in range(3)) for ax in range(3))
    gdims_dist = None
    if draw(st.booleans()):
        gdims_dist = tuple(draw(st.integers(max(lo, g - 3), g)) for g in gdims)
    return {"gdims": gdims, "pdims": pdims, "mem_order": mem_order, "gdims_dist": gdims_dist,
            "col_major": draw(st.booleans())}


small3 = st.tuples(st.integers(0, 2), st.integers(0, 2), st.integers(0, 2))


def _grids(d):
    spec = cd.make_grid_spec(d["gdims"], d["pdims"], d["mem_order"], d["gdims_dist"], d["col_major"])
    g = orc.Grid(d["gdims"], d["pdims"], gdims_dist=d["gdims_dist"], rank_order=2 if d["col_major"] else 1,
                 mem_order=d["mem_order"])
    return spec, g


def _regions_disjoint(a0, a1, b0, b1):
    return a1 <= b0 or b1 <= a0 or a0 == a1 or b0 == b1


def _cells(m, off_key, st_key):
    """flat indices (elements) a move touches on one side"""
    e = list(m.extent)
    st_ = list(getattr(m, st_key))
    k = np.indices(e).reshape(3, -1)
    return getattr(m, off_key) + k[0] * st_[0] + k[1] * st_[1] + k[2] * st_[2]


def check_rows_whole_moves(plan, out_len):
    """Move3D::dst_row_pitch (csrc/plan.h): the kernel layer may REWRITE the cells between destination rows of such a move that
    are one row pitch apart (rows_dense_kernel reads them and writes them back inside whole cache lines).  That is only
    harmless when nobody else writes them during the operation: here, no other move of the rank's plan (pack or unpack) that
    targets the output pencil -- and the spans stay inside the pencil."""
    moves = [plan.pack[i] for i in range(plan.n_pack)] + [plan.unpack[i] for i in range(plan.n_unpack)]
    moves = [m for m in moves if 0 not in list(m.extent)]
    writers = np.zeros(out_len, dtype=np.int32)
    for m in moves:
        if m.dst_buf == 1:
            writers[_cells(m, "dst_off", "ds")] += 1
    assert writers.max(initial=0) <= 1, "two moves of one plan write the same output cell"
    n_checked = 0
    for m in moves:
        if m.row_pitch <= 0:
            continue
        assert m.dst_buf == 1, "dst_row_pitch on a move that does not target the output pencil"
        e, ds = list(m.extent), list(m.ds)
        rows = [i for i in range(3) if ds[i] == 1]
        assert len(rows) == 1 and e[rows[0]] > 1, "whole rows, but no single unit-stride dim with more than one element"
        i0 = rows[0]
        assert m.row_pitch >= e[i0]
        r = [i for i in range(3) if i != i0 and ds[i] == m.row_pitch and e[i] > 1]
        if not r:
            continue  # one row per plane: nothing is a row pitch apart (the kernel layer keeps its row-by-row kernels)
        r = r[0]
        q = 3 - i0 - r
        span = (e[r] - 1) * ds[r] + e[i0]
        assert e[q] == 1 or span <= ds[q], "planes of a whole-rows move interleave"
        n_checked += 1
        for kq in range(e[q]):
            base = m.dst_off + kq * ds[q]
            assert 0 <= base and base + span <= out_len, "span of a whole-rows move leaves the pencil"
            in_rows = np.zeros(span, dtype=bool)
            for kr in range(e[r]):
                in_rows[kr * ds[r]:kr * ds[r] + e[i0]] = True
            gap = base + np.nonzero(~in_rows)[0]
            assert not writers[gap].any(), "cells between the rows of a whole-rows move are written by another move"
    return n_checked


def simulate_transpose(d, op, halos, pads, inplace, pipelined, symmetric, npergroup):
    spec, g = _grids(d)
    n = g.nranks
    ai, ao = orc.OP_AXES[op]
    pa = [g.pencil_info(r, ai, halos[0], pads[0]) for r in range(n)]
    pb = [g.pencil_info(r, ao, halos[1], pads[1]) for r in range(n)]
    wsz = g.transpose_workspace_size()
    plans = [cd.cudecompExtPlanTranspose(spec, r, op, halos[0], halos[1], pads[0], pads[1], inplace, pipelined,
                                         symmetric, npergroup) for r in range(n)]
    for r in range(n):
        check_rows_whole_moves(plans[r], max(pa[r].size, pb[r].size))
    bufs = []
    for r in range(n):
        nel = max(pa[r].size, pb[r].size)
        a = np.full(nel, -7, dtype=DT)
        a[:pa[r].size] = g.fill_pencil(pa[r], KIND)
        b = a if inplace else np.full(nel, -9, dtype=DT)
        bufs.append([a, b, np.full(wsz, -11, dtype=DT)])
    for r in range(n):
        if not plans[r].noop:
            run_moves(plans[r].pack, plans[r].n_pack, bufs[r])
    # the exchange: snapshot every chunk first (an all-to-all reads all sends before any receive is visible)
    flights = []
    for r in range(n):
        p = plans[r]
        if p.noop or not p.exchange:
            continue
        for di in range(p.nranks):
            gr = p.member_global_rank[di]
            q = plans[gr]
            so, sc = p.send_base + p.send_off[di], p.send_cnt[di]
            assert q.exchange and q.nranks == p.nranks and q.member_global_rank[p.comm_rank] == r
            assert q.recv_cnt[p.comm_rank] == sc, "send / receive counts of a pair differ"
            if symmetric:
                assert p.remote_recv_off[di] == q.recv_off[p.comm_rank], "one-sided slot offset differs from the receiver's"
            flights.append((gr, q.recv_buf, q.recv_base + q.recv_off[p.comm_rank], bufs[r][p.send_buf][so:so + sc].copy()))
        # a rank's send and receive areas must not overlap when they live in the same buffer
        if p.send_buf == p.recv_buf:
            s0 = p.send_base + min(p.send_off[i] for i in range(p.nranks))
            s1 = p.send_base + max(p.send_off[i] + p.send_cnt[i] for i in range(p.nranks))
            r0 = p.recv_base + min(p.recv_off[i] for i in range(p.nranks))
            r1 = p.recv_base + max(p.recv_off[i] + p.recv_cnt[i] for i in range(p.nranks))
            assert _regions_disjoint(s0, s1, r0, r1), "send and receive areas overlap"
    for gr, buf, off, data in flights:
        assert off + data.size <= bufs[gr][buf].size, "receive chunk runs past the buffer"
        bufs[gr][buf][off:off + data.size] = data
    for r in range(n):
        if not plans[r].noop:
            run_moves(plans[r].unpack, plans[r].n_unpack, bufs[r])
    for r in range(n):
        got = np.ascontiguousarray(bufs[r][1][:pb[r].size])
        exp = g.fill_pencil(pb[r], KIND)
        bad = orc.compare_pencil(pb[r], KIND, exp, got, True)
        assert bad == 0, "rank %d: output pencil wrong at element %d" % (r, bad - 1)


@settings(max_examples=300, deadline=None, suppress_health_check=list(HealthCheck))
@given(d=decompositions(), op=st.sampled_from(cd.OPS), in_halo=small3, out_halo=small3, in_pad=small3, out_pad=small3,
       inplace=st.booleans(), pipelined=st.booleans(), symmetric=st.booleans(), grouped=st.booleans())
def test_transpose_plans_random_decompositions(d, op, in_halo, out_halo, in_pad, out_pad, inplace, pipelined, symmetric,
                                               grouped):
    ai, ao = orc.OP_AXES[op]
    # in place means ONE buffer holding both pencils, which the API only defines when both views describe it
    # consistently; like the reference's tests, draw independent halos/padding for the two sides
    P = d["pdims"][0] if op in ("XToY", "YToX") else d["pdims"][1]
    npergroup = 0
    if grouped and P > 1:
        divisors = [k for k in range(1, P + 1) if P % k == 0]
        npergroup = divisors[len(divisors) // 2]
    simulate_transpose(d, op, (in_halo, out_halo), (in_pad, out_pad), inplace, pipelined, symmetric, npergroup)


@settings(max_examples=80, deadline=None, suppress_health_check=list(HealthCheck))
@given(d=decompositions(max_ranks=8), inplace=st.booleans(), symmetric=st.booleans())
def test_transpose_cycle_returns_the_input(d, inplace, symmetric):
    zero = ((0, 0, 0), (0, 0, 0))
    for op in cd.OPS:
        simulate_transpose(d, op, zero, zero, inplace, False, symmetric, 0)


from tests.bodies import stage_of_move as _stage_of  # noqa: E402


def simulate_staged_transpose(d, op, halos, pads, inplace, stages, npergroup):
    """The staged pipeline of the one-sided pipelined transports (csrc/transport.cc peerStagedExchange) on host arrays:
    stage by stage, pack the stage's range of every chunk, deliver ONLY the matching contiguous sub-chunks, unpack the
    stage's range from every source.  The receive areas start poisoned, so an unpack stage that touched a range
    belonging to a later stage would spoil the output."""
    spec, g = _grids(d)
    n = g.nranks
    ai, ao = orc.OP_AXES[op]
    pa = [g.pencil_info(r, ai, halos[0], pads[0]) for r in range(n)]
    pb = [g.pencil_info(r, ao, halos[1], pads[1]) for r in range(n)]
    wsz = g.transpose_workspace_size()
    plans = [cd.cudecompExtPlanTranspose(spec, r, op, halos[0], halos[1], pads[0], pads[1], inplace, True, True, npergroup)
             for r in range(n)]
    bufs = []
    for r in range(n):
        nel = max(pa[r].size, pb[r].size)
        a = np.full(nel, -7, dtype=DT)
        a[:pa[r].size] = g.fill_pencil(pa[r], KIND)
        b = a if inplace else np.full(nel, -9, dtype=DT)
        bufs.append([a, b, np.full(wsz, -11, dtype=DT)])
    exchanging = [p for p in plans if not p.noop and p.exchange]
    if not exchanging:
        return simulate_transpose(d, op, halos, pads, inplace, True, True, npergroup)
    by_comm = {}
    for r, p in enumerate(plans):  # every member of one communicator must arrive at the same stage axis and limit
        if p.exchange:
            key = tuple(p.member_global_rank[i] for i in range(p.nranks))
            by_comm.setdefault(key, set()).add((p.stage_axis, p.stage_limit))
    assert all(len(v) == 1 for v in by_comm.values()), by_comm
    K_of = [max(1, min(stages, p.stage_limit)) if (not p.noop and p.exchange) else 1 for p in plans]
    Kmax = max(K_of)
    # the executor issues ALL pack stages before the first unpack stage on the caller's stream (an in-place unpack may
    # overwrite input that a later pack stage still needs); transfers of stage k may start as soon as pack k is done
    flights = [[] for _ in range(Kmax)]
    for k in range(Kmax):
        for r in range(n):
            p, K = plans[r], K_of[r]
            if p.noop or not p.exchange or k >= K:
                continue
            moves = (cd.ExtMove * max(p.n_pack, 1))(*[_stage_of(p.pack[i], p.stage_axis, k, K) for i in range(p.n_pack)])
            run_moves(moves, p.n_pack, bufs[r])
            for di in range(p.nranks):
                gr = p.member_global_rank[di]
                q = plans[gr]
                assert K_of[gr] == K, "members of one communicator disagree on the number of stages"
                nn = p.send_n[di]
                assert nn == q.recv_n[p.comm_rank] and p.send_cnt[di] % nn == 0
                per = p.send_cnt[di] // nn
                lo, hi = nn * k // K, nn * (k + 1) // K
                so = p.send_base + p.send_off[di] + lo * per
                flights[k].append((gr, q.recv_buf, q.recv_base + q.recv_off[p.comm_rank] + lo * per,
                                   bufs[r][p.send_buf][so:so + (hi - lo) * per].copy()))
    for k in range(Kmax):
        for gr, buf, off, data in flights[k]:
            bufs[gr][buf][off:off + data.size] = data
        for r in range(n):
            p, K = plans[r], K_of[r]
            if p.noop or not p.exchange or k >= K:
                continue
            moves = (cd.ExtMove * max(p.n_unpack, 1))(*[_stage_of(p.unpack[i], p.stage_axis, k, K) for i in range(p.n_unpack)])
            run_moves(moves, p.n_unpack, bufs[r])
    for r in range(n):
        got = np.ascontiguousarray(bufs[r][1][:pb[r].size])
        exp = g.fill_pencil(pb[r], KIND)
        bad = orc.compare_pencil(pb[r], KIND, exp, got, True)
        assert bad == 0, "rank %d: output pencil wrong at element %d (staged, %d stages)" % (r, bad - 1, stages)


@settings(max_examples=200, deadline=None, suppress_health_check=list(HealthCheck))
@given(d=decompositions(), op=st.sampled_from(cd.OPS), in_halo=small3, out_halo=small3, in_pad=small3, out_pad=small3,
       inplace=st.booleans(), stages=st.integers(1, 6), grouped=st.booleans())
def test_staged_exchange_random_decompositions(d, op, in_halo, out_halo, in_pad, out_pad, inplace, stages, grouped):
    P = d["pdims"][0] if op in ("XToY", "YToX") else d["pdims"][1]
    npergroup = 0
    if grouped and P > 1:
        divisors = [k for k in range(1, P + 1) if P % k == 0]
        npergroup = divisors[len(divisors) // 2]
    simulate_staged_transpose(d, op, (in_halo, out_halo), (in_pad, out_pad), inplace, stages, npergroup)


def simulate_direct_put(d, op, halos, pads, npergroup):
    """The direct-to-destination form of the one-sided plans: every rank's `direct` moves read its own input pencil
    and write the owners' OUTPUT pencils in their final layout -- no workspace, no unpack.  Destinations written by
    different ranks must not collide, and every output interior must equal the analytic expectation."""
    spec, g = _grids(d)
    n = g.nranks
    ai, ao = orc.OP_AXES[op]
    pa = [g.pencil_info(r, ai, halos[0], pads[0]) for r in range(n)]
    pb = [g.pencil_info(r, ao, halos[1], pads[1]) for r in range(n)]
    plans = [cd.cudecompExtPlanTranspose(spec, r, op, halos[0], halos[1], pads[0], pads[1], False, False, True,
                                         npergroup) for r in range(n)]
    inp = [g.fill_pencil(pa[r], KIND) for r in range(n)]
    out = [np.full(pb[r].size, -9, dtype=DT) for r in range(n)]
    hits = [np.zeros(pb[r].size, dtype=np.int32) for r in range(n)]
    for r in range(n):
        p = plans[r]
        if not p.exchange:
            assert p.n_direct == 0
            run_moves(p.pack, p.n_pack, [inp[r], out[r], None])
            continue
        assert p.n_direct == p.nranks, "one direct move per member"
        seen = set()
        for j in range(p.n_direct):
            m = p.direct[j]
            seen.add(m.peer)
            gr = p.member_global_rank[m.peer]
            assert m.src_buf == 0 and m.dst_buf == 1
            run_moves([m], 1, [inp[r], out[gr], None])
            run_moves([m], 1, [np.ones(pa[r].size, dtype=np.int32), hits[gr], None])
        assert seen == set(range(p.nranks))
    for r in range(n):
        if plans[r].exchange:
            assert hits[r].max() <= 1, "two ranks wrote the same output cell of rank %d" % r
        exp = g.fill_pencil(pb[r], KIND)
        bad = orc.compare_pencil(pb[r], KIND, exp, out[r], True)
        assert bad == 0, "rank %d: output pencil wrong at element %d" % (r, bad - 1)


@settings(max_examples=200, deadline=None, suppress_health_check=list(HealthCheck))
@given(d=decompositions(), op=st.sampled_from(cd.OPS), in_halo=small3, out_halo=small3, in_pad=small3, out_pad=small3,
       grouped=st.booleans())
def test_direct_put_plans_random_decompositions(d, op, in_halo, out_halo, in_pad, out_pad, grouped):
    P = d["pdims"][0] if op in ("XToY", "YToX") else d["pdims"][1]
    npergroup = 0
    if grouped and P > 1:
        divisors = [k for k in range(1, P + 1) if P % k == 0]
        npergroup = divisors[len(divisors) // 2]
    simulate_direct_put(d, op, (in_halo, out_halo), (in_pad, out_pad), npergroup)


def simulate_halos(d, axis, halo, periods, padding, force_packed):
    """dims 0, 1, 2 in sequence (so edges and corners fill), compared with the oracle's restatement after every dim
    and with the analytic reference at the end; returns False if the configuration is not supported."""
    spec, g = _grids(d)
    n = g.nranks
    pin = [g.pencil_info(r, axis, halo, padding) for r in range(n)]
    ws = max(max(g.halo_workspace_size(r, axis, halo) for r in range(n)), 1)
    data = [g.fill_pencil(pin[r], KIND, halo_style=True) for r in range(n)]
    ref = [a.copy() for a in data]
    ref_work = [np.zeros(ws, dtype=DT) for _ in range(n)]
    work = [np.full(ws, -13, dtype=DT) for _ in range(n)]
    for dim in range(3):
        rc = g.update_halos(axis, KIND, ref, ref_work, halo, periods, dim, padding)
        try:
            plans = [cd.cudecompExtPlanHalo(spec, r, axis, halo, periods, dim, padding, force_packed) for r in range(n)]
        except cd.CudecompError as e:
            assert rc != orc.OK and e.code == rc, "product refused (%d) what the oracle accepts (%d)" % (e.code, rc)
            return False
        assert rc == orc.OK, "oracle refused what the product accepts"
        if any(pl.kind == 1 for pl in plans) and halo[dim] > d["gdims"][dim]:
            # periodic self copy of a halo wider than the whole (single-rank) dimension: source and destination of
            # the two wrap copies overlap, in the reference too (halo.h:165-193 issues both in one kernel) -- the
            # result is whatever the hardware's ordering makes it, so there is nothing to compare
            return None
        for r in range(n):
            bufs = [data[r], data[r], work[r]]
            if plans[r].kind != 0:
                run_moves(plans[r].pre, plans[r].n_pre, bufs)
        flights = []
        for r in range(n):
            p = plans[r]
            if p.kind in (0, 1):
                continue
            src = [data[r], data[r], work[r]][p.xbuf]
            for i in range(2):  # my low (0) / high (1) face goes to that neighbour's opposite receive slot
                nb = p.neighbor[i]
                if nb < 0:
                    continue
                q = plans[nb]
                assert q.kind == p.kind and q.neighbor[1 - i] == r and q.face_elements == p.face_elements
                flights.append((nb, q.xbuf, q.recv_off[1 - i], src[p.send_off[i]:p.send_off[i] + p.face_elements].copy()))
        for nb, xbuf, off, face in flights:
            dst = [data[nb], data[nb], work[nb]][xbuf]
            dst[off:off + face.size] = face
        for r in range(n):
            if plans[r].kind != 0:
                run_moves(plans[r].post, plans[r].n_post, [data[r], data[r], work[r]])
        for r in range(n):
            assert np.array_equal(data[r], ref[r]), "rank %d differs from the oracle after dim %d" % (r, dim)
    for r in range(n):
        assert np.array_equal(data[r], g.fill_halo_reference(pin[r], KIND, periods)), "rank %d: analytic halo check" % r
    return True


@settings(max_examples=300, deadline=None, suppress_health_check=list(HealthCheck))
@given(d=decompositions(), axis=st.integers(0, 2), halo=small3, periods=st.tuples(st.booleans(), st.booleans(), st.booleans()),
       padding=small3, force_packed=st.booleans())
def test_halo_plans_random_decompositions(d, axis, halo, periods, padding, force_packed):
    from hypothesis import assume
    assume(simulate_halos(d, axis, halo, periods, padding, force_packed) is not None)


def test_halo_wider_than_the_neighbours_slab_is_refused_like_the_oracle():
    d = {"gdims": (4, 4, 4), "pdims": (2, 2), "mem_order": ((0, 1, 2),) * 3, "gdims_dist": None, "col_major": False}
    assert simulate_halos(d, 0, (0, 3, 0), (True, True, True), (0, 0, 0), False) is False
    assert simulate_halos(d, 0, (0, 2, 0), (True, True, True), (0, 0, 0), False) is True


def test_empty_pencils_are_not_supported():
    spec = cd.make_grid_spec((3, 8, 8), (4, 1), ((0, 1, 2),) * 3)  # 3 planes over 4 ranks: one Y pencil is empty
    with pytest.raises(cd.CudecompError) as e:
        cd.cudecompExtPlanTranspose(spec, 0, "XToY")
    assert e.value.code == 2  # CUDECOMP_RESULT_NOT_SUPPORTED
    with pytest.raises(cd.CudecompError) as e:
        cd.cudecompExtPlanHalo(spec, 0, 1, (1, 1, 1), (True, True, True), 0)
    assert e.value.code == 2


@settings(max_examples=300, deadline=None, suppress_health_check=list(HealthCheck))
@given(d=decompositions(max_ranks=24), halo=small3, padding=small3, disp=st.integers(-3, 3), periodic=st.booleans())
def test_index_maps_random_decompositions(d, halo, padding, disp, periodic):
    """Pencil geometry, neighbour ranks and workspace sizes of EVERY rank of random decompositions (the functions
    behind cudecompGetPencilInfo / GetShiftedRank / Get*WorkspaceSize) against the oracle, whose index maps are pinned
    to the reference's golden vectors."""
    spec, g = _grids(d)
    for rank in range(g.nranks):
        for axis in range(3):
            got = cd.cudecompExtPencilInfo(spec, rank, axis, halo, padding).as_dict()
            assert got == g.pencil_info(rank, axis, halo, padding).as_dict(), (rank, axis)
            tws, hws = cd.cudecompExtWorkspaceSizes(spec, rank, axis, halo)
            assert tws == g.transpose_workspace_size() and hws == g.halo_workspace_size(rank, axis, halo), (rank, axis)
            for dim in range(3):
                assert cd.cudecompExtShiftedRank(spec, rank, axis, dim, disp, periodic) == \
                    g.shifted_rank(rank, axis, dim, disp, periodic), (rank, axis, dim, disp, periodic)


def test_index_maps_reject_bad_arguments():
    spec = cd.make_grid_spec((8, 8, 8), (2, 2), ((0, 1, 2),) * 3)
    for call in (lambda: cd.cudecompExtPencilInfo(spec, 4, 0), lambda: cd.cudecompExtPencilInfo(spec, 0, 3),
                 lambda: cd.cudecompExtPencilInfo(spec, 0, 0, (-1, 0, 0)), lambda: cd.cudecompExtShiftedRank(spec, 0, 0, 3, 1, False)):
        with pytest.raises(cd.CudecompError) as e:
            call()
        assert e.value.code == 1  # CUDECOMP_RESULT_INVALID_USAGE
    bad = cd.make_grid_spec((8, 8, 8), (2, 2), ((0, 1, 1), (0, 1, 2), (0, 1, 2)))
    with pytest.raises(cd.CudecompError):
        cd.cudecompExtPencilInfo(bad, 0, 0)


# ---- two-hop relay of low-fan-out exchanges (csrc/plan.h RelayPlan, csrc/transport.cc peerRelayAlltoall) --------------------
def simulate_relayed_transpose(d, op, halos, pads, inplace):
    """The relayed exchange on host arrays, rank by rank as the executor orders it: pack; step 1 -- every rank scatters the
    slices of its chunks into the relay regions of all ranks (two slices per chunk straight into the destination's
    receive area); step 2 -- every rank forwards what sits in ITS relay region; self chunk; unpack.  Relay regions and
    receive areas start poisoned, slots must not overlap, and every output pencil must match the analytic oracle.
    Returns False when the planner says the exchange is not worth relaying."""
    spec, g = _grids(d)
    n = g.nranks
    ai, ao = orc.OP_AXES[op]
    pa = [g.pencil_info(r, ai, halos[0], pads[0]) for r in range(n)]
    pb = [g.pencil_info(r, ao, halos[1], pads[1]) for r in range(n)]
    wsz = g.transpose_workspace_size()
    plans = [cd.cudecompExtPlanTranspose(spec, r, op, halos[0], halos[1], pads[0], pads[1], inplace, False, True, 0) for r in range(n)]
    relays = [cd.cudecompExtPlanRelay(spec, r, op, halos[0], halos[1], pads[0], pads[1], inplace) for r in range(n)]
    assert len({rp.applies for rp in relays}) == 1, "ranks disagree on whether the exchange is relayed"
    if not relays[0].applies:
        return False
    assert len({(rp.slot_elements, rp.relay_elements) for rp in relays}) == 1, "ranks disagree on the relay region's size"
    bufs, relay = [], []
    for r in range(n):
        nel = max(pa[r].size, pb[r].size)
        a = np.full(nel, -7, dtype=DT)
        a[:pa[r].size] = g.fill_pencil(pa[r], KIND)
        b = a if inplace else np.full(nel, -9, dtype=DT)
        bufs.append([a, b, np.full(wsz, -11, dtype=DT)])
        relay.append(np.full(relays[r].relay_elements, -13, dtype=DT))
    for r in range(n):
        run_moves(plans[r].pack, plans[r].n_pack, bufs[r])
    written = [np.zeros(relays[r].relay_elements, dtype=bool) for r in range(n)]
    landed = [np.zeros(bufs[r][plans[r].recv_buf].size, dtype=np.int32) for r in range(n)]
    # step 1 (all sends read packed data; nothing of step 2 can have happened: it waits for every scatter)
    for r in range(n):
        p, rp = plans[r], relays[r]
        send = bufs[r][p.send_buf]
        for k in range(rp.n_scatter):
            m = rp.scatter[k]
            data = send[p.send_base + m.src_off:p.send_base + m.src_off + m.count].copy()
            assert data.size == m.count and m.dst_rank != r
            if m.to_relay:
                assert not written[m.dst_rank][m.dst_off:m.dst_off + m.count].any(), "relay slots overlap"
                written[m.dst_rank][m.dst_off:m.dst_off + m.count] = True
                relay[m.dst_rank][m.dst_off:m.dst_off + m.count] = data
            else:
                q = plans[m.dst_rank]
                off = q.recv_base + m.dst_off
                bufs[m.dst_rank][q.recv_buf][off:off + m.count] = data
                landed[m.dst_rank][off:off + m.count] += 1
    # step 2
    for r in range(n):
        rp = relays[r]
        for k in range(rp.n_forward):
            m = rp.forward[k]
            assert not m.to_relay and m.dst_rank != r
            assert written[r][m.src_off:m.src_off + m.count].all(), "forwarding something nobody sent"
            q = plans[m.dst_rank]
            off = q.recv_base + m.dst_off
            bufs[m.dst_rank][q.recv_buf][off:off + m.count] = relay[r][m.src_off:m.src_off + m.count]
            landed[m.dst_rank][off:off + m.count] += 1
    # self chunks, then every element of every remote chunk must have arrived exactly once
    for r in range(n):
        p = plans[r]
        me = p.comm_rank
        so, ro, c = p.send_base + p.send_off[me], p.recv_base + p.recv_off[me], p.send_cnt[me]
        bufs[r][p.recv_buf][ro:ro + c] = bufs[r][p.send_buf][so:so + c].copy()
        for s in range(p.nranks):
            if s != me:
                lo = p.recv_base + p.recv_off[s]
                assert (landed[r][lo:lo + p.recv_cnt[s]] == 1).all(), "a received chunk has holes or double deliveries"
    for r in range(n):
        run_moves(plans[r].unpack, plans[r].n_unpack, bufs[r])
        got = np.ascontiguousarray(bufs[r][1][:pb[r].size])
        exp = g.fill_pencil(pb[r], KIND)
        bad = orc.compare_pencil(pb[r], KIND, exp, got, True)
        assert bad == 0, "rank %d: output pencil wrong at element %d after the relayed exchange" % (r, bad - 1)
    return True


@st.composite
def relay_decompositions(draw):
    # grids whose X<->Y or Y<->Z exchange has two members while the node has at least four ranks
    pdims = draw(st.sampled_from([(2, 2), (2, 4), (4, 2), (2, 3), (3, 2), (2, 5), (2, 6), (6, 2), (2, 8)]))
    lo = max(pdims)
    gdims = tuple(draw(st.integers(lo, lo + 9)) for _ in range(3))
    if draw(st.booleans()):
        mem_order = tuple(draw(st.sampled_from(PERMS)) for _ in range(3))
    else:
        ac = tuple(draw(st.booleans()) for _ in range(3))
        mem_order = tuple(tuple((ax + i) % 3 if ac[ax] else i for i in range(3)) for ax in range(3))
    gdims_dist = tuple(draw(st.integers(max(lo, g - 3), g)) for g in gdims) if draw(st.booleans()) else None
    return {"gdims": gdims, "pdims": pdims, "mem_order": mem_order, "gdims_dist": gdims_dist, "col_major": draw(st.booleans())}


@settings(max_examples=150, deadline=None, suppress_health_check=list(HealthCheck))
@given(d=relay_decompositions(), in_halo=small3, out_halo=small3, in_pad=small3, out_pad=small3, inplace=st.booleans())
def test_relayed_exchange_random_decompositions(d, in_halo, out_halo, in_pad, out_pad, inplace):
    relayed = 0
    for op in cd.OPS:
        relayed += simulate_relayed_transpose(d, op, (in_halo, out_halo), (in_pad, out_pad), inplace)
    # a two-member exchange on a grid of >= 4 ranks is always relayed: at least the two ops of the 2-wide grid dim
    assert relayed >= 2, (d["pdims"], relayed)


def test_relay_is_not_offered_where_it_cannot_pay():
    """Slab grids use every link already, two ranks have no relays; an exchange is relayed when it drives at most a third of
    the links a rank has (plan.h relayWorthwhile): the two-member exchanges of 2x4 / 4x2, every exchange of a 4x4 grid."""
    zero = (0, 0, 0)
    for pdims, ops in (((1, 8), ()), ((8, 1), ()), ((2, 1), ()), ((4, 4), tuple(cd.OPS)), ((2, 4), ("XToY", "YToX")), ((4, 2), ("YToZ", "ZToY"))):
        spec = cd.make_grid_spec((16, 16, 16), pdims, [(0, 1, 2)] * 3)
        for op in cd.OPS:
            rp = cd.cudecompExtPlanRelay(spec, 0, op, zero, zero, zero, zero, False)
            assert bool(rp.applies) == (op in ops), (pdims, op)
