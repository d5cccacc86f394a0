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
        if torch.cuda.is_available():  # (tests/test_rank_pool.py drives the geometry queries through here on the CPU)
            ndev = torch.cuda.device_count()
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % max(ndev, 1))
            torch.zeros(1, device="cuda")  # create the context before the library probes the device
        _HANDLE = cd.cudecompInit()
    return _HANDLE


def pool_probe(rank, nranks, args):
    """What a rank of tests/mp.py's pool looks like from inside: process id, the launcher variables of THIS job, selected
    switches, and -- through a library handle over the job's world -- every rank's pencil shape (geometry only: runs without
    a GPU).  args: {"raise_on": rank that fails, "sleep": seconds, "handle": bool}."""
    import time
    out = {"pid": os.getpid(), "rank": rank, "nranks": nranks, "env_rank": os.environ.get("RANK"),
           "world_size": os.environ.get("WORLD_SIZE"), "switch": os.environ.get(args.get("switch", "CUDECOMP_TEST_SWITCH")),
           "pool": os.environ.get("CUDECOMP_TEST_RANK_POOL")}
    if args.get("raise_on") == rank:
        raise RuntimeError("pool_probe: rank %d fails on purpose" % rank)
    if args.get("sleep"):
        time.sleep(args["sleep"])
    if args.get("handle", True):
        h = _handle(rank)
        gd = cd.cudecompGridDescCreate(h, cd.make_config(args.get("gdims", (12, 10, 14)), args.get("pdims", (nranks, 1))))
        out["shape"] = list(cd.cudecompGetPencilInfo(h, gd, 1).shape)
        out["handle_id"] = id(h) if not isinstance(h, int) else h
        cd.cudecompGridDescDestroy(h, gd)
    return out


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
    lib_data = args.get("data_alloc", "torch") == "malloc"  # pencils from cudecompMalloc: NVSHMEM_SM takes the direct put
    to_free = []
    for oop in args.get("out_of_place", [True, False]):
        if lib_data:
            a, pa = G.library_bytes(cd, h, gd, nel * es)
            b, pb = G.library_bytes(cd, h, gd, nel * es) if oop else (a, None)
            to_free += [p for p in (pa, pb) if p]
            a.fill_(0x5A)
            if oop:
                b.fill_(0xA5)
        else:
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
    if args.get("expect_path"):
        # the intended executor path really ran (e.g. "rccl" for the self-exchange tests: real librccl calls)
        counters = cd.cudecompExtGetCounters(h, gd)
        for name in args["expect_path"]:
            if counters[name] <= 0:
                failures.append("executor path %r did not run: %r" % (name, counters))
    if args.get("expect_counts"):   # exact counts of executor paths of this descriptor, e.g. {"rotations": 2}
        counters = cd.cudecompExtGetCounters(h, gd)
        for name, want in args["expect_counts"].items():
            if counters[name] != want:
                failures.append("executor path %r ran %d times, expected %d" % (name, counters[name], want))
    torch.cuda.synchronize()
    for p in to_free:
        cd.cudecompFree(h, gd, p)
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


def transpose_every_byte(rank, nranks, args):
    """Transposes onto halo-carrying / padded pencils with EVERY BYTE of the destination buffer compared: interior cells as the
    analytic oracle says, everything else (halo cells, padding, the tail of the buffer) exactly what it held before the call.
    The reference's tests compare the interior only (tests/ctest/transpose_tests.cc:356-378); the dense row copy
    (rows_dense_kernel) reads the cells between consecutive rows and writes them back, which this check would catch doing
    anything else.  args["expect_kernel"]: {op: substring of the name of the last kernel the op must have launched}."""
    h, gd, g = _setup(rank, nranks, args)
    kind = args.get("kind", 1)
    dt, es = orc.KINDS[kind]
    halos, pads = args["halos"], args["pads"]
    pin = [cd.cudecompGetPencilInfo(h, gd, ax, halos[ax], pads[ax]) for ax in range(3)]
    opin = [g.pencil_info(rank, ax, halos[ax], pads[ax]) for ax in range(3)]
    nel = max(p.size for p in pin) + 64
    wsz = cd.cudecompGetTransposeWorkspaceSize(h, gd)
    work_ptr = cd.cudecompMalloc(h, gd, wsz * es)
    failures = []
    rng = np.random.default_rng(1234 + rank)
    for oop in args.get("out_of_place", [True, False]):
        a = G.to_device(rng.integers(0, 256, nel * es, dtype=np.uint8))
        b = G.to_device(rng.integers(0, 256, nel * es, dtype=np.uint8)) if oop else a
        ops = args.get("ops", list(cd.OPS))
        first_ax = orc.OP_AXES[ops[0]][0]
        init = G.to_host(a).view(dt).copy()
        init[:pin[first_ax].size] = g.fill_pencil(opin[first_ax], kind)
        a.copy_(torch.from_numpy(init.view(np.uint8)))
        cur, nxt = a, b
        for op in ops:
            ai, ao = orc.OP_AXES[op]
            before = G.to_host(nxt).view(dt).copy()
            cd.cudecompTranspose(op, h, gd, cur.data_ptr(), nxt.data_ptr(), work_ptr, cd.DTYPE_OF_KIND[kind], halos[ai],
                                 halos[ao], pads[ai], pads[ao], G.stream_ptr())
            torch.cuda.synchronize()
            last = cd.cudecompExtLastKernelName()
            got = G.to_host(nxt).view(dt)
            exp = g.fill_pencil(opin[ao], kind)
            interior = np.real(exp) != -1  # (cells outside the interior are filled with -1 / (-1, -1))
            want = before.copy()
            want[:pin[ao].size][interior] = exp[interior]
            if not np.array_equal(got.view(np.uint8), want.view(np.uint8)):
                gb, wb = got.view(np.uint8).reshape(-1, es), want.view(np.uint8).reshape(-1, es)
                bad = np.nonzero((gb != wb).any(axis=1))[0]
                where = "interior" if (bad[0] < pin[ao].size and interior[bad[0]]) else "OUTSIDE the interior"
                failures.append("rank %d oop=%s %s: %d cells differ, first at %d (%s; want %r got %r); last kernel %s" %
                                (rank, oop, op, bad.size, bad[0], where, want[bad[0]], got[bad[0]], last))
                break
            need = (args.get("expect_kernel") or {}).get(op)
            if need and need not in last:
                failures.append("rank %d oop=%s %s: last kernel %r, expected %r" % (rank, oop, op, last, need))
            if oop:
                cur, nxt = nxt, cur
    torch.cuda.synchronize()
    cd.cudecompFree(h, gd, work_ptr)
    cd.cudecompGridDescDestroy(h, gd)
    return failures


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


def cycle_properties(rank, nranks, args):
    """Size-independent checks at full benchmark sizes: random 64/32-bit payload, X->Y->Z->Y->X; returns the
    per-hop wrapping sums of this rank's interior (the test adds them over ranks: every hop must preserve the
    global multiset) and whether the round trip reproduced this rank's input bit for bit."""
    h, gd, g = _setup(rank, nranks, args)
    kind = args.get("kind", 1)
    es = orc.KINDS[kind][1]
    idt = {4: torch.int32, 8: torch.int64, 16: torch.int64}[es]
    words = es // (4 if es == 4 else 8)
    pin = [cd.cudecompGetPencilInfo(h, gd, ax) for ax in range(3)]
    nel = max(p.size for p in pin)
    wsz = cd.cudecompGetTransposeWorkspaceSize(h, gd)
    work = cd.cudecompMalloc(h, gd, wsz * es)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99 + rank)
    a = torch.randint(-2**30, 2**30, (nel * words,), dtype=idt, device="cuda", generator=gen)
    b = torch.zeros_like(a)
    keep = a[:pin[0].size * words].clone()
    sums = [int(keep.sum(dtype=torch.int64))]
    cur, nxt = a, b
    for op in cd.OPS:
        ao = orc.OP_AXES[op][1]
        cd.cudecompTranspose(op, h, gd, cur.data_ptr(), nxt.data_ptr(), work, cd.DTYPE_OF_KIND[kind], stream=G.stream_ptr())
        torch.cuda.synchronize()
        sums.append(int(nxt[:pin[ao].size * words].sum(dtype=torch.int64)))
        cur, nxt = nxt, cur
    same = bool(torch.equal(cur[:pin[0].size * words], keep))
    cd.cudecompFree(h, gd, work)
    cd.cudecompGridDescDestroy(h, gd)
    return {"sums": sums, "round_trip_exact": same}


def expected_pencil_words(p, gdims, es):
    """Closed form of a halo-free pencil filled with the global linear index, as raw words on the device: cell
    (gx, gy, gz) holds lin = gx + X * (gy + Y * gz); 8-byte elements carry lin, 4-byte elements its low 31 bits,
    16-byte elements the pair (lin, ~lin).  Returns an int64 / int32 tensor of p.size * words entries."""
    shape, lo, order = list(p.shape), list(p.lo), list(p.order)
    coef = [1, gdims[0], gdims[0] * gdims[1]]
    v = None
    for m in range(3):
        idx = (torch.arange(shape[m], device="cuda", dtype=torch.int64) + lo[m]) * coef[order[m]]
        view = [1, 1, 1]
        view[2 - m] = -1
        v = idx.view(view) if v is None else v + idx.view(view)
    v = v.contiguous().view(-1)
    if es == 8:
        return v
    if es == 4:
        return (v & 0x7FFFFFFF).to(torch.int32)
    return torch.stack([v, ~v], dim=1).contiguous().view(-1)


def cycle_exact(rank, nranks, args):
    """Full-size check of EVERY cell: the X pencil is filled on the device with the global linear index, and after
    every hop of X->Y->Z->Y->X the whole output pencil is compared on the device with the closed form of that
    pencil (bit-exact; nothing is sampled).  Optionally in place, and with a host-asynchrony report: how long the
    library call kept the host vs how long the device needed."""
    import time
    h, gd, g = _setup(rank, nranks, args)
    kind = args.get("kind", 1)
    es = orc.KINDS[kind][1]
    idt = torch.int32 if es == 4 else torch.int64
    words = 2 if es == 16 else 1
    gdims = args["gdims"]
    pin = [cd.cudecompGetPencilInfo(h, gd, ax) for ax in range(3)]
    nel = max(p.size for p in pin)
    work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * es)
    inplace = bool(args.get("inplace", False))
    to_free = []
    if args.get("data_alloc", "torch") == "malloc":
        ta, pa = G.library_bytes(cd, h, gd, nel * es)
        a = ta.view(idt)
        a.zero_()
        to_free.append(pa)
        if inplace:
            b = a
        else:
            tb, pb = G.library_bytes(cd, h, gd, nel * es)
            b = tb.view(idt)
            b.fill_(-3)
            to_free.append(pb)
    else:
        a = torch.zeros(nel * words, dtype=idt, device="cuda")
        b = a if inplace else torch.full((nel * words,), -3, dtype=idt, device="cuda")
    a[:pin[0].size * words] = expected_pencil_words(pin[0], gdims, es)
    failures, host_ms, total_ms = [], [], []
    cur, nxt = a, b
    for it in range(args.get("cycles", 1)):
        for op in cd.OPS:
            ao = orc.OP_AXES[op][1]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cd.cudecompTranspose(op, h, gd, cur.data_ptr(), nxt.data_ptr(), work, cd.DTYPE_OF_KIND[kind], stream=G.stream_ptr())
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host_ms.append((t1 - t0) * 1e3)
            total_ms.append((t2 - t0) * 1e3)
            exp = expected_pencil_words(pin[ao], gdims, es)
            got = nxt[:pin[ao].size * words]
            if not torch.equal(got, exp):
                bad = int((got != exp).nonzero()[0])
                failures.append("rank %d cycle %d %s: %d cells differ, first cell %d holds %d, expected %d"
                                % (rank, it, op, int((got != exp).sum()), bad // words, int(got[bad]), int(exp[bad])))
                # (no early exit: the other ranks may not have failed, and the calls are collective)
                got.copy_(exp)
            del exp
            if not inplace:
                cur.fill_(-5)  # a hop that re-read stale input would show
                cur, nxt = nxt, cur
    burst = None
    if args.get("burst_cycles"):
        # back-to-back calls, no synchronisation in between: how long the host needs to ISSUE them vs how long the
        # device needs to RUN them (a host-ordered transport would make the two equal)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args["burst_cycles"]):
            for op in cd.OPS:
                cd.cudecompTranspose(op, h, gd, cur.data_ptr(), nxt.data_ptr(), work, cd.DTYPE_OF_KIND[kind], stream=G.stream_ptr())
                if not inplace:
                    cur, nxt = nxt, cur
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        burst = {"host_ms": (t1 - t0) * 1e3, "total_ms": (t2 - t0) * 1e3}
        exp = expected_pencil_words(pin[0], gdims, es)
        if not torch.equal(cur[:pin[0].size * words], exp):
            failures.append("rank %d: X pencil wrong after the burst of cycles" % rank)
        del exp
    counters = cd.cudecompExtGetCounters(h, gd)
    torch.cuda.synchronize()
    del a, b, cur, nxt
    for p in to_free:
        cd.cudecompFree(h, gd, p)
    cd.cudecompFree(h, gd, work)
    cd.cudecompGridDescDestroy(h, gd)
    return {"failures": failures, "host_ms": host_ms, "total_ms": total_ms, "counters": counters, "burst": burst}


def graph_cycle(rank, nranks, args):
    """The stream-ordered one-sided transposes inside a USER's hipGraph: one eager warm-up cycle, then the whole
    X->Y->Z->Y->X cycle (packs, per-peer copies on the library's copy streams, flag kernels, unpacks) is captured
    from the caller's stream and replayed `replays` times, each time on fresh input data, with every cell of the
    final and of an intermediate pencil checked on the device.  Works because the call counter of the exchanges
    lives in device memory (a replay advances it by itself)."""
    h, gd, g = _setup(rank, nranks, args)
    kind = args.get("kind", 1)
    es = orc.KINDS[kind][1]
    idt = torch.int32 if es == 4 else torch.int64
    words = 2 if es == 16 else 1
    gdims = args["gdims"]
    pin = [cd.cudecompGetPencilInfo(h, gd, ax) for ax in range(3)]
    nel = max(p.size for p in pin)
    work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * es)
    if args.get("torch_data"):  # (bisecting aid: data pencils from torch instead of cudecompMalloc)
        ta, tb, tz = (torch.zeros(nel * es, dtype=torch.uint8, device="cuda") for _ in range(3))
        pa = pb = pz = None
    else:
        ta, pa = G.library_bytes(cd, h, gd, nel * es)
        tb, pb = G.library_bytes(cd, h, gd, nel * es)
        tz, pz = G.library_bytes(cd, h, gd, nel * es)  # keeps a copy of the Z pencil of every cycle
    a, b, z = ta.view(idt), tb.view(idt), tz.view(idt)
    x0 = expected_pencil_words(pin[0], gdims, es)
    z0 = expected_pencil_words(pin[2], gdims, es)
    stream = torch.cuda.current_stream() if args.get("default_stream") else torch.cuda.Stream()
    # x0 / z0 (and the zero fills of torch buffers) were produced on the DEFAULT stream; a torch side stream is not ordered
    # behind it by itself.  Round 6: without this line the first assignment below could read x0 before its kernels had run --
    # it then copied what the block held before (the previous job's pencil, when the allocator reused the block), the rank's
    # whole contribution to the cycle was wrong, and the failure looked like a missed flag (profiles/r06_pooled_suite_failure.md).
    stream.wait_stream(torch.cuda.current_stream())
    failures = []

    def cycle(sptr):
        cur, nxt = (a, a) if args.get("in_place") else (a, b)   # in_place: every hop on the one buffer
        for op in cd.OPS:
            cd.cudecompTranspose(op, h, gd, cur.data_ptr(), nxt.data_ptr(), work, cd.DTYPE_OF_KIND[kind], stream=sptr)
            if op == "YToZ":
                z.copy_(nxt)
            cur, nxt = nxt, cur

    with torch.cuda.stream(stream):
        a[:x0.numel()] = x0
        cycle(stream.cuda_stream)  # warm-up: first-use allocations and mappings happen outside the capture
        stream.synchronize()
        if not torch.equal(a[:x0.numel()], x0) or not torch.equal(z[:z0.numel()], z0):
            bad_x, bad_z = (a[:x0.numel()] != x0).nonzero().flatten(), (z[:z0.numel()] != z0).nonzero().flatten()
            failures.append("rank %d: eager warm-up cycle wrong: X pencil %d cells (first %s), Z pencil %d cells (first %s, holds %s)"
                            % (rank, bad_x.numel(), bad_x[:3].tolist(), bad_z.numel(), bad_z[:3].tolist(),
                               z[bad_z[:3]].tolist() if bad_z.numel() else []))
    graph = torch.cuda.CUDAGraph()
    if not args.get("no_capture"):
        with torch.cuda.graph(graph, stream=None if args.get("default_stream") else stream, capture_error_mode="thread_local"):
            cycle(torch.cuda.current_stream().cuda_stream)
    for it in range(0 if args.get("no_capture") else args.get("replays", 3)):
        shift = 1000003 * (it + 1)
        with torch.cuda.stream(stream):
            a[:x0.numel()] = x0 + shift
            b.fill_(-1)
            z.fill_(-2)
            graph.replay()
            stream.synchronize()
            if not torch.equal(a[:x0.numel()], x0 + shift):
                failures.append("rank %d replay %d: the cycle did not return the input" % (rank, it))
            if not torch.equal(z[:z0.numel()], z0 + shift):
                failures.append("rank %d replay %d: Z pencil wrong" % (rank, it))
    counters = cd.cudecompExtGetCounters(h, gd)
    del graph
    torch.cuda.synchronize()
    del a, b, z, ta, tb, tz
    for p in (pa, pb, pz):
        if p is not None:
            cd.cudecompFree(h, gd, p)
    cd.cudecompFree(h, gd, work)
    cd.cudecompGridDescDestroy(h, gd)
    return {"failures": failures, "counters": counters}


def autotune_full_size(rank, nranks, args):
    """BASELINE config 3, "autotuned pgrid", at its size: the library's autotuner picks process grid and transport for
    a 1024^3 fp64 array on `nranks` ranks (all candidates measured), then one cycle with the selection is checked cell
    by cell."""
    h = _handle(rank)
    cfg = cd.make_config(args["gdims"], (0, 0), axis_contiguous=args.get("ac", (1, 1, 1)))
    opt = cd.cudecompGridDescAutotuneOptionsSetDefaults()
    opt.n_warmup_trials, opt.n_trials = 1, 2
    opt.dtype = cd.DTYPE_OF_KIND[args.get("kind", 1)]
    opt.autotune_transpose_backend = True
    opt.disable_nccl_backends = bool(args.get("disable_nccl", True))
    gd = cd.cudecompGridDescCreate(h, cfg, opt)
    picked = {"pdims": [cfg.pdims[0], cfg.pdims[1]], "tb": cfg.transpose_comm_backend}
    cd.cudecompGridDescDestroy(h, gd)
    res = cycle_exact(rank, nranks, dict(args, pdims=picked["pdims"], transpose_backend=picked["tb"]))
    return {"picked": picked, "failures": res["failures"]}


def absent_peer(rank, nranks, args):
    """Failure detection of the one-sided transport: rank 1 never enters the second transpose.  Rank 0 must get an
    error within the configured timeout instead of hanging -- from the host rendezvous (MPI enums) or, for the purely
    stream-ordered NVSHMEM enums, from the device-side wait that gives up and is reported by the next library call."""
    import time
    h, gd, g = _setup(rank, nranks, args)
    kind = 1
    es = 8
    pin = [cd.cudecompGetPencilInfo(h, gd, ax) for ax in range(3)]
    nel = max(p.size for p in pin)
    work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * es)
    a = torch.zeros(nel, dtype=torch.float64, device="cuda")
    b = torch.zeros(nel, dtype=torch.float64, device="cuda")
    cd.cudecompTranspose("XToY", h, gd, a.data_ptr(), b.data_ptr(), work, cd.DOUBLE, stream=G.stream_ptr())
    torch.cuda.synchronize()
    out = {"rank": rank, "error_code": None, "seconds": None}
    if rank == 0:
        t0 = time.perf_counter()
        try:
            cd.cudecompTranspose("YToX", h, gd, b.data_ptr(), a.data_ptr(), work, cd.DOUBLE, stream=G.stream_ptr())
            torch.cuda.synchronize()   # stream-ordered transports: the wait kernel gives up on the device ...
            cd.cudecompTranspose("XToY", h, gd, a.data_ptr(), b.data_ptr(), work, cd.DOUBLE, stream=G.stream_ptr())  # ... and this call reports it
        except cd.CudecompError as e:
            out["error_code"] = e.code
        out["seconds"] = time.perf_counter() - t0
    else:
        time.sleep(args.get("absent_for", 12.0))  # alive, but never calls
    return out


def halo_exact(rank, nranks, args):
    """Halo update checked in EVERY cell at any size (reference: tests/ctest/halo_tests.cc:229-272 compares the whole
    pencil): the interior is initialised on the device with the global linear index (-1 elsewhere), dims 0, 1, 2 are
    updated, and the whole halo-carrying pencil is compared on the device with its closed form -- the global coordinate of
    every cell wrapped modulo the global extent for periodic dims, -1 where a non-periodic domain ends."""
    h, gd, g = _setup(rank, nranks, args)
    halo, periods = args["halo"], args["periods"]
    gdims = args["gdims"]
    failures = []

    def bc(vec, m):  # memory position m varies fastest for m = 0: [shape2, shape1, shape0] tensors by broadcasting
        view = [1, 1, 1]
        view[2 - m] = -1
        return vec.view(view)

    for axis in args.get("axes", [0]):
        p = cd.cudecompGetPencilInfo(h, gd, axis, halo)
        shape, lo, order = list(p.shape), list(p.lo), list(p.order)
        pos = {order[m]: m for m in range(3)}
        idx = [torch.arange(shape[m], device="cuda", dtype=torch.int64) for m in range(3)]
        gcoord, interior, wrapped, valid = [None] * 3, [None] * 3, [None] * 3, [None] * 3
        for m in range(3):
            ax = order[m]
            gcoord[ax] = idx[m] + lo[m] - halo[ax]
            interior[ax] = (idx[m] >= halo[ax]) & (idx[m] < shape[m] - halo[ax])
            outside = (gcoord[ax] < 0) | (gcoord[ax] >= gdims[ax])
            wrapped[ax] = torch.remainder(gcoord[ax], gdims[ax]) if periods[ax] else gcoord[ax]
            valid[ax] = torch.ones_like(outside) if periods[ax] else ~outside
        minus1 = torch.full((), -1.0, dtype=torch.float64, device="cuda")

        def build(coord, mask):
            val = bc(coord[0], pos[0]) + gdims[0] * (bc(coord[1], pos[1]) + gdims[1] * bc(coord[2], pos[2]))
            ok = bc(mask[0], pos[0]) & bc(mask[1], pos[1]) & bc(mask[2], pos[2])
            return torch.where(ok, val.to(torch.float64), minus1).contiguous()

        data = build(gcoord, interior)
        wsz = max(cd.cudecompGetHaloWorkspaceSize(h, gd, axis, halo), 1)
        work = cd.cudecompMalloc(h, gd, wsz * 8)
        for dim in range(3):
            cd.cudecompUpdateHalos(axis, h, gd, data.data_ptr(), work, cd.DOUBLE, halo, periods, dim, None, G.stream_ptr())
        torch.cuda.synchronize()
        exp = build(wrapped, valid)
        if args.get("check_closed_form_against_oracle"):  # small grids: the closed form itself vs the oracle's reference
            ref = g.fill_halo_reference(g.pencil_info(rank, axis, halo), 1, periods)
            if not np.array_equal(exp.view(-1).cpu().numpy(), ref):
                failures.append("rank %d axis %d: the device-side closed form differs from the oracle" % (rank, axis))
        if not torch.equal(data, exp):
            ne = (data != exp).view(-1)
            first = int(ne.nonzero()[0])
            failures.append("rank %d axis %d: %d of %d cells differ (first at %d: expected %r, got %r)"
                            % (rank, axis, int(ne.sum()), ne.numel(), first, float(exp.view(-1)[first]), float(data.view(-1)[first])))
            del ne
        cd.cudecompFree(h, gd, work)
        del data, exp
        torch.cuda.empty_cache()
    cd.cudecompGridDescDestroy(h, gd)
    return failures


def perf_report(rank, nranks, args):
    """Repeat the transpose cycle and two halo updates on ONE descriptor with the performance report enabled
    (environment set by the launcher before cudecompInit); return the CSV files rank 0 finds after the
    descriptor was destroyed."""
    import glob
    outdir = os.environ["CUDECOMP_PERFORMANCE_REPORT_WRITE_DIR"]
    h, gd, g = _setup(rank, nranks, args)
    kind = args.get("kind", 1)
    dt, es = orc.KINDS[kind]
    halo = (1, 1, 1)
    pin = [cd.cudecompGetPencilInfo(h, gd, ax, halo) for ax in range(3)]
    nel = max(p.size for p in pin)
    work = cd.cudecompMalloc(h, gd, max(cd.cudecompGetTransposeWorkspaceSize(h, gd),
                                        max(cd.cudecompGetHaloWorkspaceSize(h, gd, ax, halo) for ax in range(3))) * es)
    a = torch.zeros(nel * es, dtype=torch.uint8, device="cuda")
    b = torch.zeros(nel * es, dtype=torch.uint8, device="cuda")
    for _ in range(args.get("repeat", 4)):
        cur, nxt = (a, a) if args.get("in_place") else (a, b)
        for op in cd.OPS:
            cd.cudecompTranspose(op, h, gd, cur.data_ptr(), nxt.data_ptr(), work, cd.DTYPE_OF_KIND[kind], None, None,
                                 None, None, G.stream_ptr())
            cur, nxt = nxt, cur
        for axis, dim in ((0, 1), (1, 2)):
            cd.cudecompUpdateHalos(axis, h, gd, a.data_ptr(), work, cd.DTYPE_OF_KIND[kind], halo, (True, True, True),
                                   dim, None, G.stream_ptr())
    torch.cuda.synchronize()
    counters = cd.cudecompExtGetCounters(h, gd)
    cd.cudecompFree(h, gd, work)
    cd.cudecompGridDescDestroy(h, gd)  # collective; rank 0 prints the report and writes the CSV files
    files = {}
    if rank == 0:
        for f in sorted(glob.glob(os.path.join(outdir, "*.csv"))):
            with open(f) as fh:
                files[os.path.basename(f)] = fh.read()
    return {"files": files, "counters": counters}


def repeated_cycle(rank, nranks, args):
    """The X->Y->Z->Y->X cycle several times on the SAME buffers with a check after every hop: cached plans, and with
    CUDECOMP_ENABLE_CUDA_GRAPHS=1 the captured pack loop of the pipelined backends, are replayed from the
    second iteration on."""
    h, gd, g = _setup(rank, nranks, args)
    kind = args.get("kind", 1)
    dt, es = orc.KINDS[kind]
    pin = [cd.cudecompGetPencilInfo(h, gd, ax) for ax in range(3)]
    opin = [g.pencil_info(rank, ax) for ax in range(3)]
    nel = max(p.size for p in pin)
    work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * es)
    a = torch.zeros(nel * es, dtype=torch.uint8, device="cuda")
    b = torch.zeros(nel * es, dtype=torch.uint8, device="cuda")
    failures = []
    for it in range(args.get("iterations", 3)):
        init = np.full(nel, -7, dtype=dt)
        init[:pin[0].size] = g.fill_pencil(opin[0], kind)
        a.copy_(torch.from_numpy(init.view(np.uint8)))
        b.fill_(0x5A)
        cur, nxt = a, b
        for k, op in enumerate(cd.OPS):
            ai, ao = orc.OP_AXES[op]
            if args.get("skew_ms"):
                # ranks enter the collective at very different times, in an order that changes from call to call
                import time
                order = (rank + it + k) % nranks if (it + k) % 2 else (nranks - 1 - rank)
                time.sleep(order * args["skew_ms"] * 1e-3)
            cd.cudecompTranspose(op, h, gd, cur.data_ptr(), nxt.data_ptr(), work, cd.DTYPE_OF_KIND[kind], None, None,
                                 None, None, G.stream_ptr())
            torch.cuda.synchronize()
            got = G.to_host(nxt).view(dt)[:pin[ao].size].copy()
            exp = g.fill_pencil(opin[ao], kind)
            bad = orc.compare_pencil(opin[ao], kind, exp, got, True)
            if bad:
                failures.append("rank %d iteration %d %s: mismatch at %d" % (rank, it, op, bad - 1))
            cur.fill_(0xA5)  # a stale replay that re-read the old input would show up in the next hop
            cur, nxt = nxt, cur
    counters = cd.cudecompExtGetCounters(h, gd)
    cd.cudecompFree(h, gd, work)
    cd.cudecompGridDescDestroy(h, gd)
    return {"failures": failures, "counters": counters}


def malloc_timing(rank, nranks, args):
    """Time cudecompMalloc + cudecompFree (collective: allocation, IPC export / import on every rank) per size."""
    import time
    h, gd, g = _setup(rank, nranks, args)
    out = {}
    for nbytes in args["sizes"]:
        ptrs = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.get("reps", 5)):
            ptrs.append(cd.cudecompMalloc(h, gd, nbytes))
        t1 = time.perf_counter()
        for p in ptrs:
            cd.cudecompFree(h, gd, p)
        t2 = time.perf_counter()
        out[str(nbytes)] = [round((t1 - t0) / len(ptrs) * 1e3, 3), round((t2 - t1) / len(ptrs) * 1e3, 3)]
    cd.cudecompGridDescDestroy(h, gd)
    return out


def two_handles_alternating(rank, nranks, args):
    """Two LIVE handles in one job (reference tests/ctest/api_tests.cc:575-656), each with its own descriptor (different
    rank order and backend), workspace and transport state, transposing ALTERNATELY hop by hop on the same stream; every
    cell of every output pencil is compared on the device with its closed form.  The first handle is finalised while
    the second keeps working."""
    h1, h2 = cd.cudecompInit(), cd.cudecompInit()
    gdims, pd, kind = args["gdims"], args["pdims"], args.get("kind", 1)
    es = orc.KINDS[kind][1]
    idt = torch.int32 if es == 4 else torch.int64
    words = 2 if es == 16 else 1
    torch.zeros(1, device="cuda")
    sides = []
    for h, order, backend in ((h1, cd.RANK_ORDER_ROW_MAJOR, args["backends"][0]), (h2, cd.RANK_ORDER_COL_MAJOR, args["backends"][1])):
        gd = cd.cudecompGridDescCreate(h, cd.make_config(gdims, pd, rank_order=order, axis_contiguous=args.get("ac", (0, 0, 0)),
                                                         transpose_backend=backend))
        pin = [cd.cudecompGetPencilInfo(h, gd, ax) for ax in range(3)]
        nel = max(p.size for p in pin)
        work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * es)
        a = torch.zeros(nel * words, dtype=idt, device="cuda")
        b = torch.full((nel * words,), -3, dtype=idt, device="cuda")
        a[:pin[0].size * words] = expected_pencil_words(pin[0], gdims, es)
        sides.append({"h": h, "gd": gd, "pin": pin, "work": work, "cur": a, "nxt": b})
    failures = []
    for it in range(args.get("cycles", 2)):
        for op in cd.OPS:
            ao = orc.OP_AXES[op][1]
            for s in sides:  # both issued before anything is synchronised: the two handles' exchanges interleave
                cd.cudecompTranspose(op, s["h"], s["gd"], s["cur"].data_ptr(), s["nxt"].data_ptr(), s["work"],
                                     cd.DTYPE_OF_KIND[kind], stream=G.stream_ptr())
            torch.cuda.synchronize()
            for k, s in enumerate(sides):
                exp = expected_pencil_words(s["pin"][ao], gdims, es)
                got = s["nxt"][:s["pin"][ao].size * words]
                if not torch.equal(got, exp):
                    failures.append("rank %d handle %d cycle %d %s: %d cells differ" % (rank, k, it, op, int((got != exp).sum())))
                    got.copy_(exp)
                s["cur"].fill_(-5)
                s["cur"], s["nxt"] = s["nxt"], s["cur"]
    # first handle goes away (creation order), the second runs one more cycle on its own
    s = sides[0]
    cd.cudecompFree(s["h"], s["gd"], s["work"])
    cd.cudecompGridDescDestroy(s["h"], s["gd"])
    cd.cudecompFinalize(s["h"])
    s = sides[1]
    for op in cd.OPS:
        ao = orc.OP_AXES[op][1]
        cd.cudecompTranspose(op, s["h"], s["gd"], s["cur"].data_ptr(), s["nxt"].data_ptr(), s["work"], cd.DTYPE_OF_KIND[kind],
                             stream=G.stream_ptr())
        torch.cuda.synchronize()
        exp = expected_pencil_words(s["pin"][ao], gdims, es)
        if not torch.equal(s["nxt"][:s["pin"][ao].size * words], exp):
            failures.append("rank %d second handle alone %s: cells differ" % (rank, op))
        s["cur"], s["nxt"] = s["nxt"], s["cur"]
    cd.cudecompFree(s["h"], s["gd"], s["work"])
    cd.cudecompGridDescDestroy(s["h"], s["gd"])
    cd.cudecompFinalize(s["h"])
    return {"failures": failures}


def link_info(rank, nranks, args):
    """What the start-up link probe of the one-sided transport measured (it runs when the transport comes up, i.e. with
    the first descriptor that selects a one-sided backend)."""
    h = _handle(rank)
    gd = cd.cudecompGridDescCreate(h, cd.make_config((64, 64, 64), (1, nranks), transpose_backend=cd.TRANSPOSE_COMM_NVSHMEM))
    work = cd.cudecompMalloc(h, gd, 1 << 20)
    info = cd.cudecompExtGetLinkInfo(h)
    cd.cudecompFree(h, gd, work)
    cd.cudecompGridDescDestroy(h, gd)
    return info


def pool_behaviour(rank, nranks, args):
    """cudecompMalloc / cudecompFree with the pool of released workspaces: a freed workspace is handed out again for a
    request it fits (same pointer, no new IPC mappings), a larger request gets a new one, and transposes are exact on
    recycled workspaces of several descriptors in a row (every cell)."""
    h, gd, g = _setup(rank, nranks, args)
    es = 8
    ws = cd.cudecompGetTransposeWorkspaceSize(h, gd) * es
    c0 = cd.cudecompExtGetCounters(h, gd)
    p1 = cd.cudecompMalloc(h, gd, ws)
    cd.cudecompFree(h, gd, p1)
    p2 = cd.cudecompMalloc(h, gd, ws)          # same size: must come from the pool when it is on
    cd.cudecompFree(h, gd, p2)
    p3 = cd.cudecompMalloc(h, gd, 3 * ws + (1 << 20))  # too large for the parked one
    p4 = cd.cudecompMalloc(h, gd, ws)          # the parked one again, while p3 is live
    c1 = cd.cudecompExtGetCounters(h, gd)
    out = {"same_pointer": p1 == p2 == p4, "distinct_large": p3 not in (p1, p2, p4),
           "pool_hits": c1["workspace_pool_hits"] - c0["workspace_pool_hits"], "stale": c1["stale_ipc_mappings"]}
    cd.cudecompFree(h, gd, p3)
    cd.cudecompFree(h, gd, p4)
    cd.cudecompGridDescDestroy(h, gd)
    failures = []
    for k in range(args.get("descriptors", 3)):  # fresh descriptor + "fresh" (recycled) workspace each time
        pd = args["pdims"] if k % 2 == 0 else (args["pdims"][1], args["pdims"][0])
        r = cycle_exact(rank, nranks, dict(args, pdims=pd))
        failures += r["failures"]
    out["failures"] = failures
    return out


def pool_pressure(rank, nranks, args):
    """The pool under memory pressure (round-3 advice): (1) cudecompExtTrimWorkspacePool really releases what cudecompFree
    parked; (2) a cudecompMalloc that does not fit beside the parked workspaces releases them and succeeds; (3) an
    impossible request fails on EVERY rank with the same result code, nobody hangs, and the library stays usable."""
    h, gd, g = _setup(rank, nranks, args)
    out = {"failures": []}
    mib = 1 << 20
    free0 = torch.cuda.mem_get_info()[0]
    p = cd.cudecompMalloc(h, gd, 512 * mib)
    cd.cudecompFree(h, gd, p)
    c = cd.cudecompExtGetCounters(h, gd)
    out["parked_bytes"] = c["workspace_pool_bytes"]
    cd.cudecompExtTrimWorkspacePool(h)
    torch.cuda.synchronize()
    c = cd.cudecompExtGetCounters(h, gd)
    out["parked_after_trim"] = c["workspace_pool_bytes"]
    out["freed_by_trim_mib"] = (torch.cuda.mem_get_info()[0] - free0) // mib
    # (2) park a workspace, then ask for more than what is free without it: only draining the pool makes room.  All ranks
    # share the device here, so the request is sized from what is free for ALL of them together.
    park = cd.cudecompMalloc(h, gd, int(args.get("park_gib", 8)) << 30)
    cd.cudecompFree(h, gd, park)
    out["parked_before_big"] = cd.cudecompExtGetCounters(h, gd)["workspace_pool_bytes"]
    # (3) an impossible request (more than the device has): the same error everywhere
    try:
        cd.cudecompMalloc(h, gd, int(args.get("impossible_gib", 400)) << 30)
        out["failures"].append("a 400-GiB workspace was granted")
    except cd.CudecompError as e:
        out["impossible_code"] = e.code
    out["parked_after_impossible"] = cd.cudecompExtGetCounters(h, gd)["workspace_pool_bytes"]  # drained on the way
    # still usable
    r = cycle_exact(rank, nranks, args)
    out["failures"] += r["failures"]
    cd.cudecompGridDescDestroy(h, gd)
    return out


def queue_census(rank, nranks, args):
    """The library's census of the compute queues on its GPU (cudecompExtQueueCensus) after a cycle on every rank."""
    r = cycle_exact(rank, nranks, args)
    h = _handle(rank)
    torch.cuda.synchronize()
    compute, slots = cd.cudecompExtQueueCensus(h)
    return {"failures": r["failures"], "compute": compute, "slots": slots}


def halo_timed(rank, nranks, args):
    """cudecompUpdateHalos{X,Y,Z} timing per pencil axis and dim on a multi-rank grid (BASELINE config 5 when called with
    its sizes): K timed updates per dim bracketed by device events, and the library's own per-phase samples (pack /
    exchange / unpack; needs CUDECOMP_ENABLE_PERFORMANCE_REPORT=1 in the environment)."""
    h, gd, g = _setup(rank, nranks, args)
    halo, periods = args["halo"], args["periods"]
    es = 8
    out = {}
    for axis in args.get("axes", [0, 1, 2]):
        p = cd.cudecompGetPencilInfo(h, gd, axis, halo)
        data = torch.zeros(p.size, dtype=torch.float64, device="cuda")
        wsz = max(cd.cudecompGetHaloWorkspaceSize(h, gd, axis, halo), 1)
        work = cd.cudecompMalloc(h, gd, wsz * es)
        st = G.stream_ptr()
        rec = {"pencil_shape": list(p.shape), "workspace_MiB": round(wsz * es / 2**20, 1)}
        for dim in range(3):
            for _ in range(args.get("warmup", 3)):
                cd.cudecompUpdateHalos(axis, h, gd, data.data_ptr(), work, cd.DOUBLE, halo, periods, dim, None, st)
            torch.cuda.synchronize()
            reps = args.get("reps", 10)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                cd.cudecompUpdateHalos(axis, h, gd, data.data_ptr(), work, cd.DOUBLE, halo, periods, dim, None, st)
            e1.record()
            torch.cuda.synchronize()
            t = cd.cudecompExtGetHaloTimings(h, gd, axis, dim)
            rec["dim%d" % dim] = {"ms": round(e0.elapsed_time(e1) / reps, 4), "pack_ms": round(t["pack_ms"], 4),
                                  "exchange_ms": round(t["exchange_ms"], 4), "unpack_ms": round(t["unpack_ms"], 4),
                                  "samples": t["samples"], "wire_MiB": round(t["pencil_bytes"] / 2**20, 2)}
        out["XYZ"[axis]] = rec
        cd.cudecompFree(h, gd, work)
        del data
        torch.cuda.empty_cache()
    cd.cudecompGridDescDestroy(h, gd)
    return out


def small_cycle_latency(rank, nranks, args):
    """Latency of the flag-ordered exchanges: a tiny grid, many back-to-back cycles, ms per transpose (the data is a few
    KiB, so the time is launches + flag round trips)."""
    import time
    h, gd, g = _setup(rank, nranks, args)
    es = 8
    pin = [cd.cudecompGetPencilInfo(h, gd, ax) for ax in range(3)]
    nel = max(p.size for p in pin)
    work = cd.cudecompMalloc(h, gd, max(cd.cudecompGetTransposeWorkspaceSize(h, gd), 1) * es)
    a = torch.zeros(nel, dtype=torch.float64, device="cuda")
    b = torch.zeros(nel, dtype=torch.float64, device="cuda")
    st = G.stream_ptr()

    def cycles(n):
        cur, nxt = a, b
        for _ in range(n):
            for op in cd.OPS:
                cd.cudecompTranspose(op, h, gd, cur.data_ptr(), nxt.data_ptr(), work, cd.DOUBLE, stream=st)
                cur, nxt = nxt, cur
    cycles(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = args.get("cycles", 200)
    cycles(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cd.cudecompFree(h, gd, work)
    cd.cudecompGridDescDestroy(h, gd)
    return {"us_per_transpose": dt / (4 * n) * 1e6}


def graph_cycle_failures(rank, nranks, args):
    """graph_cycle for the `many` runner (a list of failure strings)."""
    return graph_cycle(rank, nranks, args)["failures"]
