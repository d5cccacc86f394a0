"""Per-rank test bodies run by tests/mp.py (and directly, for single-rank cases)."""
import ctypes as C

import numpy as np

import cudecomp_amd as cd
from oracle import oracle as orc


def _grid_and_handle(args):
    h = cd.cudecompInit()
    cfg = cd.make_config(args["gdims"], args["pdims"], gdims_dist=args.get("gdims_dist"),
                         rank_order=args.get("rank_order", 0), axis_contiguous=args.get("ac", (0, 0, 0)),
                         mem_order=args.get("mem_order"), transpose_backend=args.get("transpose_backend"),
                         halo_backend=args.get("halo_backend"))
    gd = cd.cudecompGridDescCreate(h, cfg)
    return h, gd


def index_queries(rank, nranks, args):
    """Pencil infos, shifted ranks and workspace sizes of this rank through the C ABI."""
    h, gd = _grid_and_handle(args)
    out = {"rank": rank, "pencil": [], "shifted": [], "config_pdims": None}
    for axis in range(3):
        out["pencil"].append(cd.cudecompGetPencilInfo(h, gd, axis, args.get("halo"), args.get("padding")).as_dict())
    for q in args.get("shifted_queries", []):
        out["shifted"].append(cd.cudecompGetShiftedRank(h, gd, q["axis"], q["dim"], q["displacement"], q["periodic"]))
    out["transpose_ws"] = cd.cudecompGetTransposeWorkspaceSize(h, gd)
    if args.get("halo"):
        out["halo_ws"] = [cd.cudecompGetHaloWorkspaceSize(h, gd, a, args["halo"]) for a in range(3)]
    c = cd.cudecompGetGridDescConfig(h, gd)
    out["config_pdims"] = [c.pdims[0], c.pdims[1]]
    out["config_rank_order"] = c.rank_order
    cd.cudecompGridDescDestroy(h, gd)
    cd.cudecompFinalize(h)
    return out


def two_handles(rank, nranks, args):
    """Reference tests/ctest/api_tests.cc:575-656 on every rank of a distributed job: two LIVE handles over the same
    communicator with independent descriptors (row-major on the first, column-major on the second), each descriptor
    only valid with its own handle, both finalised in creation order."""
    L = cd.lib()
    h1 = cd.cudecompInit()
    h2 = cd.cudecompInit()
    assert h1.value != h2.value
    pd = tuple(args["pdims"])
    gd1 = cd.cudecompGridDescCreate(h1, cd.make_config(args["gdims"], pd, rank_order=cd.RANK_ORDER_ROW_MAJOR))
    gd2 = cd.cudecompGridDescCreate(h2, cd.make_config(args["gdims"], pd, rank_order=cd.RANK_ORDER_COL_MAJOR))
    out = {"rank": rank, "cross": [], "pencil_row_major": [], "pencil_col_major": []}
    out["cross"].append(L.cudecompGetGridDescConfigVersioned(h2, gd1, C.byref(cd.GridDescConfig()), 104, 1))
    out["cross"].append(L.cudecompGridDescDestroy(h2, gd1))
    unused = C.c_void_p()
    out["cross"].append(L.cudecompMalloc(h2, gd1, C.byref(unused), 1024))
    out["unused_is_null"] = not unused.value
    for axis in range(3):
        out["pencil_row_major"].append(cd.cudecompGetPencilInfo(h1, gd1, axis, args.get("halo"), args.get("padding")).as_dict())
        out["pencil_col_major"].append(cd.cudecompGetPencilInfo(h2, gd2, axis, args.get("halo"), args.get("padding")).as_dict())
    out["rank_orders"] = [cd.cudecompGetGridDescConfig(h1, gd1).rank_order, cd.cudecompGetGridDescConfig(h2, gd2).rank_order]
    if args.get("destroy_descriptors", True):
        cd.cudecompGridDescDestroy(h1, gd1)
        cd.cudecompGridDescDestroy(h2, gd2)
    cd.cudecompFinalize(h1)  # creation order (FinalizesMultipleHandlesInCreationOrder)
    # the second handle must be fully usable after the first is gone
    gd3 = cd.cudecompGridDescCreate(h2, cd.make_config(args["gdims"], pd))
    out["after_first_finalize"] = cd.cudecompGetPencilInfo(h2, gd3, 0, args.get("halo"), args.get("padding")).as_dict()
    cd.cudecompGridDescDestroy(h2, gd3)
    cd.cudecompFinalize(h2)
    return out


# ---- numpy execution of the PRODUCT's plans (host-logic check, no GPU) -------------------------------
def _view(buf, off, extent, strides):
    es = buf.itemsize
    return np.lib.stride_tricks.as_strided(buf[off:], shape=tuple(int(e) for e in extent)[::-1],
                                           strides=tuple(int(s) * es for s in strides)[::-1])


def run_moves(moves, n, bufs):
    for i in range(n):
        m = moves[i]
        if 0 in list(m.extent):
            continue
        src = _view(bufs[m.src_buf], m.src_off, m.extent, m.ss)
        dst = _view(bufs[m.dst_buf], m.dst_off, m.extent, m.ds)
        dst[...] = src.copy()


def stage_of_move(m, axis, k, K):
    """python twin of csrc/plan.cc stageOfMove: range k of K of the move's extent along global axis `axis`"""
    r = cd.ExtMove()
    for f in ("src_buf", "dst_buf", "src_off", "dst_off", "peer", "row_pitch"):
        setattr(r, f, getattr(m, f))
    n = m.extent[axis]
    lo, hi = n * k // K, n * (k + 1) // K
    for i in range(3):
        r.extent[i], r.ss[i], r.ds[i] = m.extent[i], m.ss[i], m.ds[i]
    r.extent[axis] = hi - lo
    r.src_off = m.src_off + lo * m.ss[axis]
    r.dst_off = m.dst_off + lo * m.ds[axis]
    return r


def staged_exchange_gloo(plan, bufs, stages, rank, itemsize, dt):
    """The staged pipeline of the one-sided pipelined transports (csrc/transport.cc peerStagedExchange) with a real
    multi-process exchange: all pack stages, then per stage the contiguous sub-chunks travel over gloo and the stage's
    range of every source is unpacked.  The receive area is poisoned first."""
    import torch
    import torch.distributed as dist
    P, K = plan.nranks, max(1, min(stages, plan.stage_limit))
    sendb, recvb = bufs[plan.send_buf], bufs[plan.recv_buf]
    if plan.recv_buf == 2:
        lo = plan.recv_base + min(plan.recv_off[i] for i in range(P))
        hi = plan.recv_base + max(plan.recv_off[i] + plan.recv_cnt[i] for i in range(P))
        recvb[lo:hi] = -12345
    for k in range(K):
        moves = (cd.ExtMove * max(plan.n_pack, 1))(*[stage_of_move(plan.pack[i], plan.stage_axis, k, K) for i in range(plan.n_pack)])
        run_moves(moves, plan.n_pack, bufs)
    for k in range(K):
        reqs, stage = [], {}
        for d in range(P):
            gr = plan.member_global_rank[d]
            n_s, n_r = plan.send_n[d], plan.recv_n[d]
            per_s, per_r = plan.send_cnt[d] // n_s, plan.recv_cnt[d] // n_r
            so = plan.send_base + plan.send_off[d] + (n_s * k // K) * per_s
            sc = (n_s * (k + 1) // K - n_s * k // K) * per_s
            ro = plan.recv_base + plan.recv_off[d] + (n_r * k // K) * per_r
            rc = (n_r * (k + 1) // K - n_r * k // K) * per_r
            if gr == rank:
                assert sc == rc
                recvb[ro:ro + rc] = sendb[so:so + sc].copy()
                continue
            s = torch.from_numpy(np.ascontiguousarray(sendb[so:so + sc]).view(np.uint8).copy())
            stage[d] = (torch.zeros(rc * itemsize, dtype=torch.uint8), ro, rc)
            if sc:
                reqs.append(dist.isend(s, gr))
            if rc:
                reqs.append(dist.irecv(stage[d][0], gr))
        for q in reqs:
            q.wait()
        for d, (t, ro, rc) in stage.items():
            recvb[ro:ro + rc] = t.numpy().view(dt)
        moves = (cd.ExtMove * max(plan.n_unpack, 1))(*[stage_of_move(plan.unpack[i], plan.stage_axis, k, K) for i in range(plan.n_unpack)])
        run_moves(moves, plan.n_unpack, bufs)


def relayed_exchange_gloo(plan, relays, bufs, rank, nranks, itemsize, dt):
    """The two-hop relay (csrc/plan.h RelayPlan, csrc/transport.cc peerRelayAlltoall) with a real multi-process exchange: every
    rank derives the relay moves of ALL ranks from the stateless planner (as the library does), so a receiver knows what
    each sender will send it in either step and in which order.  Step 1 scatters my slices (relay slots of the other ranks /
    two direct slices per chunk), step 2 forwards what landed in MY relay region; relay region and receive area start
    poisoned."""
    import torch
    import torch.distributed as dist
    mine = relays[rank]
    sendb, recvb = bufs[plan.send_buf], bufs[plan.recv_buf]
    relay = np.full(max(mine.relay_elements, 1), -777, dtype=dt)
    P = plan.nranks
    lo = plan.recv_base + min(plan.recv_off[i] for i in range(P))
    hi = plan.recv_base + max(plan.recv_off[i] + plan.recv_cnt[i] for i in range(P))
    me = plan.comm_rank
    self_chunk = sendb[plan.send_base + plan.send_off[me]:plan.send_base + plan.send_off[me] + plan.send_cnt[me]].copy()
    if plan.recv_buf == 2:
        recvb[lo:hi] = -12345

    def step(which, src_buf, src_base):
        reqs, landing = [], []
        for s in range(nranks):  # what the others send me, in the order they send it
            if s == rank:
                continue
            moves = getattr(relays[s], which)
            for k in range(getattr(relays[s], "n_" + which)):
                m = moves[k]
                if m.dst_rank == rank and m.count:
                    t = torch.zeros(m.count * itemsize, dtype=torch.uint8)
                    reqs.append(dist.irecv(t, s))
                    landing.append((t, bool(m.to_relay), m.dst_off, m.count))
        moves = getattr(mine, which)
        for k in range(getattr(mine, "n_" + which)):
            m = moves[k]
            assert m.dst_rank != rank
            data = np.ascontiguousarray(src_buf[src_base + m.src_off:src_base + m.src_off + m.count])
            assert data.size == m.count
            reqs.append(dist.isend(torch.from_numpy(data.view(np.uint8).copy()), m.dst_rank))
        for q in reqs:
            q.wait()
        for t, to_relay, off, cnt in landing:
            if to_relay:
                relay[off:off + cnt] = t.numpy().view(dt)
            else:
                recvb[plan.recv_base + off:plan.recv_base + off + cnt] = t.numpy().view(dt)

    step("scatter", sendb, plan.send_base)
    dist.barrier()  # (the library: wait "step 1 of everybody has landed")
    step("forward", relay, 0)
    recvb[plan.recv_base + plan.recv_off[me]:plan.recv_base + plan.recv_off[me] + plan.recv_cnt[me]] = self_chunk


def plan_transpose_gloo(rank, nranks, args):
    """Execute the product's transpose plans with numpy + gloo send/recv and check them against the analytic
    oracle, for a full X->Y->Z->Y->X chain (tests/cc/transpose_test.cc:516-559)."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=nranks)
    h, gd = _grid_and_handle(args)
    kind = args.get("kind", 1)
    dt = orc.KINDS[kind][0]
    g = orc.Grid(args["gdims"], args["pdims"], gdims_dist=args.get("gdims_dist"),
                 rank_order=args.get("rank_order", 0), axis_contiguous=args.get("ac", (0, 0, 0)),
                 mem_order=args.get("mem_order"))
    halos, pads = args.get("halos", [(0, 0, 0)] * 3), args.get("pads", [(0, 0, 0)] * 3)
    pin = [cd.cudecompGetPencilInfo(h, gd, ax, halos[ax], pads[ax]) for ax in range(3)]
    opin = [g.pencil_info(rank, ax, halos[ax], pads[ax]) for ax in range(3)]
    for ax in range(3):
        assert pin[ax].as_dict() == opin[ax].as_dict(), "pencil info differs from the oracle"
    wsz = cd.cudecompGetTransposeWorkspaceSize(h, gd)
    assert wsz == g.transpose_workspace_size()
    nel = max(p.size for p in pin)
    failures = []
    relayed = {"n": 0}
    for backend in args.get("backends", [cd.TRANSPOSE_COMM_NCCL]):
        for oop in (True, False):
            a = np.zeros(nel, dtype=dt)
            b = np.zeros(nel, dtype=dt) if oop else a
            a[:pin[0].size] = g.fill_pencil(opin[0], kind)
            cur, nxt = a, b
            for op in cd.OPS:
                ai, ao = orc.OP_AXES[op]
                work = np.zeros(wsz, dtype=dt)
                plan = cd.cudecompExtGetTransposePlan(h, gd, op, halos[ai], halos[ao], pads[ai], pads[ao],
                                                      inplace=not oop, backend_override=backend)
                bufs = [cur, nxt, work]
                if args.get("relay") and not plan.noop and plan.exchange:
                    mo = args.get("mem_order") or tuple(tuple((ax + i) % 3 if args.get("ac", (0, 0, 0))[ax] else i for i in range(3))
                                                        for ax in range(3))
                    spec = cd.make_grid_spec(args["gdims"], args["pdims"], mo, args.get("gdims_dist"), args.get("rank_order", 0) == 2)
                    relays = [cd.cudecompExtPlanRelay(spec, r, op, halos[ai], halos[ao], pads[ai], pads[ao], not oop)
                              for r in range(nranks)]
                    if relays[rank].applies:
                        relayed["n"] += 1
                        run_moves(plan.pack, plan.n_pack, bufs)
                        relayed_exchange_gloo(plan, relays, bufs, rank, nranks, a.itemsize, dt)
                        run_moves(plan.unpack, plan.n_unpack, bufs)
                        exp = g.fill_pencil(opin[ao], kind)
                        got = np.ascontiguousarray(nxt[:pin[ao].size])
                        bad = orc.compare_pencil(opin[ao], kind, exp, got, True)
                        if bad:
                            failures.append("relayed, oop %s %s: mismatch at %d" % (oop, op, bad - 1))
                        if oop:
                            cur, nxt = nxt, cur
                        continue
                staged = args.get("stages", 0) > 1 and backend in (cd.TRANSPOSE_COMM_NVSHMEM_PL, cd.TRANSPOSE_COMM_MPI_P2P_PL)
                if not plan.noop and plan.exchange and staged:
                    staged_exchange_gloo(plan, bufs, args["stages"], rank, a.itemsize, dt)
                elif not plan.noop:
                    run_moves(plan.pack, plan.n_pack, bufs)
                    if plan.exchange:
                        P = plan.nranks
                        sendb, recvb = bufs[plan.send_buf], bufs[plan.recv_buf]
                        # consistency of one-sided offsets: what I think member d's slot for me is must be what d thinks
                        mine = torch.tensor([plan.recv_off[i] for i in range(P)], dtype=torch.int64)
                        reqs, theirs = [], {}
                        for d in range(P):
                            gr = plan.member_global_rank[d]
                            if gr == rank:
                                continue
                            theirs[d] = torch.zeros(P, dtype=torch.int64)
                            reqs.append(dist.isend(mine.clone(), gr))
                            reqs.append(dist.irecv(theirs[d], gr))
                        for q in reqs:
                            q.wait()
                        for d, t in theirs.items():
                            if int(t[plan.comm_rank]) != plan.remote_recv_off[d]:
                                failures.append("%s: remote_recv_off[%d] mismatch" % (op, d))
                        reqs, stage = [], {}
                        for d in range(P):
                            gr = plan.member_global_rank[d]
                            so, sc = plan.send_base + plan.send_off[d], plan.send_cnt[d]
                            ro, rc = plan.recv_base + plan.recv_off[d], plan.recv_cnt[d]
                            if gr == rank:
                                assert sc == rc
                                recvb[ro:ro + rc] = sendb[so:so + sc].copy()
                                continue
                            s = torch.from_numpy(np.ascontiguousarray(sendb[so:so + sc]).view(np.uint8).copy())
                            stage[d] = (torch.zeros(rc * a.itemsize, dtype=torch.uint8), ro, rc)
                            if sc:
                                reqs.append(dist.isend(s, gr))
                            if rc:
                                reqs.append(dist.irecv(stage[d][0], gr))
                        for q in reqs:
                            q.wait()
                        for d, (t, ro, rc) in stage.items():
                            recvb[ro:ro + rc] = t.numpy().view(dt)
                    run_moves(plan.unpack, plan.n_unpack, bufs)
                exp = g.fill_pencil(opin[ao], kind)
                got = np.ascontiguousarray(nxt[:pin[ao].size])
                bad = orc.compare_pencil(opin[ao], kind, exp, got, True)
                if bad:
                    failures.append("backend %d oop %s %s: mismatch at %d" % (backend, oop, op, bad - 1))
                if oop:
                    cur, nxt = nxt, cur
    dist.barrier()
    cd.cudecompGridDescDestroy(h, gd)
    cd.cudecompFinalize(h)
    dist.destroy_process_group()
    if args.get("relay") and relayed["n"] < args.get("expect_relayed", 1):
        failures.append("only %d exchanges were relayed, expected at least %d" % (relayed["n"], args.get("expect_relayed", 1)))
    return failures
