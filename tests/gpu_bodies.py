"""GPU test bodies: the reference's transpose / halo test flows (tests/cc/transpose_test.cc:516-559,
tests/ctest/halo_tests.cc:330-380) driven through libcudecomp.so's C ABI, checked against the oracle's
restatement of the reference's analytic expected values.  Run in-process for one rank and under
tests/mp.py for several ranks (which then share the visible GPUs; the xGMI peer transport works between
processes on one device as well, RCCL needs one device per rank)."""
import os

import numpy as np
import torch

import cudecomp_amd as cd
from oracle import oracle as orc
from tests import gpu_util as G


_HANDLE = None


def _handle(rank):
    """One library handle per process, shared by every case the process runs."""
    global _HANDLE
    if _HANDLE is None:
        ndev = torch.cuda.device_count()
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % max(ndev, 1))
        torch.zeros(1, device="cuda")  # create the context before the library probes the device
        _HANDLE = cd.cudecompInit()
    return _HANDLE


def _setup(rank, nranks, args):
    h = _handle(rank)
    cfg = cd.make_config(args["gdims"], args["pdims"], gdims_dist=args.get("gdims_dist"),
                         rank_order=args.get("rank_order", 0), axis_contiguous=args.get("ac", (0, 0, 0)),
                         mem_order=args.get("mem_order"), transpose_backend=args.get("transpose_backend"),
                         halo_backend=args.get("halo_backend"))
    gd = cd.cudecompGridDescCreate(h, cfg)
    g = orc.Grid(args["gdims"], args["pdims"], gdims_dist=args.get("gdims_dist"),
                 rank_order=args.get("rank_order", 0), axis_contiguous=args.get("ac", (0, 0, 0)),
                 mem_order=args.get("mem_order"))
    return h, gd, g


def transpose_chain(rank, nranks, args):
    """X->Y->Z->Y->X with a check after every hop; returns a list of failure strings."""
    h, gd, g = _setup(rank, nranks, args)
    kind = args.get("kind", 0)
    dt, es = orc.KINDS[kind]
    halos, pads = args.get("halos", [(0, 0, 0)] * 3), args.get("pads", [(0, 0, 0)] * 3)
    pin = [cd.cudecompGetPencilInfo(h, gd, ax, halos[ax], pads[ax]) for ax in range(3)]
    opin = [g.pencil_info(rank, ax, halos[ax], pads[ax]) for ax in range(3)]
    failures = []
    for ax in range(3):
        if pin[ax].as_dict() != opin[ax].as_dict():
            failures.append("pencil info axis %d differs from the oracle" % ax)
    wsz = cd.cudecompGetTransposeWorkspaceSize(h, gd)
    nel = max(p.size for p in pin)
    use_malloc = args.get("work_alloc", "malloc") == "malloc"
    if use_malloc:
        work_ptr = cd.cudecompMalloc(h, gd, wsz * es)
    else:
        work_t = torch.zeros(wsz * es, dtype=torch.uint8, device="cuda")
        work_ptr = work_t.data_ptr()
    ops = args.get("ops", list(cd.OPS))
    for oop in args.get("out_of_place", [True, False]):
        a = torch.full((nel * es,), 0x5A, dtype=torch.uint8, device="cuda")
        b = torch.full((nel * es,), 0xA5, dtype=torch.uint8, device="cuda") if oop else a
        first_ax = orc.OP_AXES[ops[0]][0]
        init = np.full(nel, -7, dtype=dt)
        init[:pin[first_ax].size] = g.fill_pencil(opin[first_ax], kind)
        a.copy_(torch.from_numpy(init.view(np.uint8)))
        cur, nxt = a, b
        for op in ops:
            ai, ao = orc.OP_AXES[op]
            cd.cudecompTranspose(op, h, gd, cur.data_ptr(), nxt.data_ptr(), work_ptr, cd.DTYPE_OF_KIND[kind], halos[ai],
                                 halos[ao], pads[ai], pads[ao], G.stream_ptr())
            torch.cuda.synchronize()
            got = G.to_host(nxt).view(dt)[:pin[ao].size].copy()
            exp = g.fill_pencil(opin[ao], kind)
            bad = orc.compare_pencil(opin[ao], kind, exp, got, True)
            if bad:
                failures.append("rank %d oop=%s %s: mismatch at %d (exp %r got %r)" % (rank, oop, op, bad - 1,
                                                                                      exp[bad - 1], got[bad - 1]))
                break
            if oop:
                cur, nxt = nxt, cur
    if use_malloc:
        cd.cudecompFree(h, gd, work_ptr)
    cd.cudecompGridDescDestroy(h, gd)
    return failures


def single_transpose(rank, nranks, args):
    """One direct transpose, the ctest flow (tests/ctest/transpose_tests.cc:380-428)."""
    a = dict(args)
    ai, ao = orc.OP_AXES[args["op"]]
    halos, pads = [(0, 0, 0)] * 3, [(0, 0, 0)] * 3
    halos[ai], pads[ai] = tuple(args["in_halo"]), tuple(args["in_pad"])
    halos[ao], pads[ao] = tuple(args["out_halo"]), tuple(args["out_pad"])
    a.update(halos=halos, pads=pads, ops=[args["op"]], out_of_place=[bool(args["out_of_place"])])
    return transpose_chain(rank, nranks, a)


def halo_sweep(rank, nranks, args):
    """UpdateHalos{X,Y,Z} for dim 0,1,2 in sequence, whole-buffer compare."""
    h, gd, g = _setup(rank, nranks, args)
    kind = args.get("kind", 0)
    dt, es = orc.KINDS[kind]
    halo, periods, padding = args["halo"], args["periods"], args.get("padding", (0, 0, 0))
    failures = []
    for axis in args.get("axes", [0, 1, 2]):
        p = cd.cudecompGetPencilInfo(h, gd, axis, halo, padding)
        op = g.pencil_info(rank, axis, halo, padding)
        wsz = max(cd.cudecompGetHaloWorkspaceSize(h, gd, axis, halo), 1)
        if wsz != max(g.halo_workspace_size(rank, axis, halo), 1):
            failures.append("halo workspace size differs from the oracle")
        work_ptr = cd.cudecompMalloc(h, gd, wsz * es)
        data = G.to_device(g.fill_pencil(op, kind, halo_style=True).view(np.uint8))
        for dim in range(3):
            cd.cudecompUpdateHalos(axis, h, gd, data.data_ptr(), work_ptr, cd.DTYPE_OF_KIND[kind], halo, periods, dim,
                                   padding, G.stream_ptr())
        torch.cuda.synchronize()
        got = G.to_host(data).view(dt)
        exp = g.fill_halo_reference(op, kind, periods)
        bad = orc.compare_pencil(op, kind, exp, got, False)
        if bad:
            failures.append("rank %d axis %d: mismatch at %d (exp %r got %r)" % (rank, axis, bad - 1, exp[bad - 1],
                                                                                 got[bad - 1]))
        cd.cudecompFree(h, gd, work_ptr)
    cd.cudecompGridDescDestroy(h, gd)
    return failures


def many(rank, nranks, args):
    """Run a list of {"fn": name, "args": {...}} jobs in this process group; returns all failures."""
    out = []
    for job in args["jobs"]:
        fails = globals()[job["fn"]](rank, nranks, job["args"])
        out.extend("%s: %s" % (job.get("id", job["fn"]), f) for f in fails)
    return out


def autotune_then_cycle(rank, nranks, args):
    """Grid + backend autotuning (cudecompGridDescCreate with options), then a checked transpose cycle and a
    halo sweep with whatever was selected."""
    h = _handle(rank)
    cfg = cd.make_config(args["gdims"], args.get("pdims", (0, 0)), axis_contiguous=args.get("ac", (0, 0, 0)))
    opt = cd.cudecompGridDescAutotuneOptionsSetDefaults()
    opt.n_warmup_trials, opt.n_trials = 1, 2
    opt.dtype = cd.DTYPE_OF_KIND[args.get("kind", 1)]
    opt.autotune_transpose_backend = True
    opt.autotune_halo_backend = bool(args.get("halo_backend_too", True))
    opt.disable_nccl_backends = bool(args.get("disable_nccl", False))
    opt.skip_threshold = args.get("skip_threshold", 0.0)
    for i in range(3):
        opt.halo_extents[i] = 1
        opt.halo_periods[i] = True
    if args.get("grid_mode_halo"):
        opt.grid_mode = cd.AUTOTUNE_GRID_HALO
    gd = cd.cudecompGridDescCreate(h, cfg, opt)
    picked = {"pdims": [cfg.pdims[0], cfg.pdims[1]], "tb": cfg.transpose_comm_backend, "hb": cfg.halo_comm_backend}
    q = cd.cudecompGetGridDescConfig(h, gd)
    assert [q.pdims[0], q.pdims[1]] == picked["pdims"] and q.transpose_comm_backend == picked["tb"]
    cd.cudecompGridDescDestroy(h, gd)
    a = {"gdims": args["gdims"], "pdims": picked["pdims"], "ac": args.get("ac", (0, 0, 0)), "kind": args.get("kind", 1),
         "transpose_backend": picked["tb"], "halo_backend": picked["hb"]}
    fails = transpose_chain(rank, nranks, a)
    fails += halo_sweep(rank, nranks, dict(a, halo=(1, 1, 1), periods=(1, 1, 1), axes=[0]))
    return {"picked": picked, "failures": fails}
