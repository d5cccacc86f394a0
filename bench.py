#!/usr/bin/env python3
"""bench.py -- the reference's headline measurement on MI355X: wall time and effective bandwidth of one
X->Y->Z->Y->X transpose cycle of a 1024^3 fp64 array (BASELINE.json), through libcudecomp.so's C ABI.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full cycle (4 transposes) on device-resident synthetic data.  The protocol is the reference
autotuner's (warm-up cycles, then timed cycles bracketed by barriers + device syncs, max over ranks;
reference src/autotune.cc:541-636).  Rank 0 prints ONE JSON line.

  value        effective GB/s = 4 * global array bytes / cycle time (whole job), as SURVEY.md section 8(d)
  roofline     dominant kernel of the cycle at this configuration vs the HBM roofline
  cpu_baseline the CPU oracle (oracle/, a port of the reference semantics; the reference has no CPU path)
               timed on one host core on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=1024, help="global grid is size^3")
    ap.add_argument("--layout", choices=["contiguous", "default"], default="contiguous",
                    help="contiguous: every pencil axis-contiguous (the layout of the reference's published "
                         "1024-class numbers, every hop permutes); default: X fastest everywhere")
    ap.add_argument("--backend", default="auto", help="auto | nccl | nccl_pl | peer | peer_pl")
    ap.add_argument("--pdims", type=int, nargs=2, default=None)
    ap.add_argument("--inplace", action="store_true")
    ap.add_argument("--watchdog", type=int, default=900, help="multi-GPU runs: give up after this many seconds (0 = never)")
    ap.add_argument("--cpu-sample", type=int, default=512, help="grid edge of the CPU-baseline sample (0 = skip)")
    return ap.parse_args()


def measured_traffic(layout):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 on
    gfx950 + WRITE_SIZE, see profiles/*_pmc_summary.json); PMC collection cannot run inside the timed loop."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            return int(json.load(f)["layouts"][layout]["hbm_traffic_bytes_per_launch"])
    except (KeyError, ValueError, OSError):
        return None


class c_stdout_to_stderr:
    """Route C-level stdout (the library's autotune log) to stderr for the duration of the block."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import ctypes
        ctypes.CDLL(None).fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)


def cpu_baseline(sample, layout):
    """The CPU path timed on this host's cores, on a sample^3 fp64 array with the same layout: the host-MPI path
    (oracle/cpu_mpi_cycle: pack -> MPI_Alltoallv -> unpack with one rank per core, up to 64) when an MPI installation
    is present, else the single-process oracle on one core."""
    mpi = cpu_baseline_mpi(sample, layout)
    if mpi is not None:
        return mpi
    import numpy as np
    from oracle import oracle as orc
    ac = (1, 1, 1) if layout == "contiguous" else (0, 0, 0)
    g = orc.Grid((sample,) * 3, (1, 1), axis_contiguous=ac)
    n = sample ** 3
    rng = np.random.default_rng(0)
    a, b = [rng.random(n)], [np.zeros(n)]
    w = [np.zeros(g.transpose_workspace_size())]
    t0 = time.perf_counter()
    for op in ("XToY", "YToZ", "ZToY", "YToX"):
        assert g.transpose(op, 1, a, b, w) == orc.OK
        a, b = b, a
    dt = time.perf_counter() - t0
    return {"value": round(4 * n * 8 / dt / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": "%d^3 fp64 X->Y->Z->Y->X cycle, 1x1 grid, %s layout, out-of-place, %.1f s on one core"
                      % (sample, layout, dt)}


def cpu_baseline_mpi(sample, layout):
    """mpirun -np R oracle/cpu_mpi_cycle ... ; None if there is no MPI here or anything goes wrong."""
    import shutil
    import subprocess
    try:
        mpirun = shutil.which("mpirun") or "/opt/conda/bin/mpirun"
        if not os.path.exists(mpirun) or not os.path.exists("/opt/conda/include/mpi.h"):
            return None
        exe = os.path.join(ROOT, "oracle", "cpu_mpi_cycle")
        if not os.path.exists(exe):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "cpu_mpi_cycle"], stdout=sys.stderr,
                                  stderr=sys.stderr)
        cores = os.cpu_count() or 1
        # one rank per core, up to 64 (power of two, near-square process grid)
        ranks = 1
        while ranks * 2 <= min(cores, 64):
            ranks *= 2
        pr = 1
        while pr * pr * 2 <= ranks:
            pr *= 2
        pc = ranks // pr
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        t0 = time.perf_counter()
        out = subprocess.run([mpirun, "-np", str(ranks), exe, str(sample), str(pr), str(pc),
                              "1" if layout == "contiguous" else "0", "1", "3"], env=env, capture_output=True, text=True,
                             timeout=240)
        dt = time.perf_counter() - t0
        rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        if not rec["round_trip_ok"]:
            return None
        return {"value": round(rec["gbps"], 4), "unit": "GB/s", "cores": ranks, "kind": "port",
                "sample": "%d^3 fp64 X->Y->Z->Y->X cycle on host memory, %d MPI ranks (%dx%d grid, one per core; MPICH "
                          "shared-memory MPI_Alltoallv), %s layout, out-of-place, %.3f s per cycle, 1 warm-up + 3 timed, "
                          "%.1f s in total" % (sample, ranks, pr, pc, layout, rec["cycle_s"], dt)}
    except Exception as e:  # the baseline is a courtesy number: never let it take the benchmark down
        sys.stderr.write("bench.py: host-MPI CPU baseline unavailable (%s); using the single-core oracle\n" % e)
        return None


def main():
    args = parse()
    # dmabuf IPC (the only mode the pool's hosts support) must be selected before the HIP runtime starts
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    if not os.path.exists(os.path.join(ROOT, "cudecomp_amd", "lib", "libcudecomp.so")):
        # binaries are git-ignored; build once (rank 0 of a multi-GPU launch, the others wait for the file)
        import subprocess
        if int(os.environ.get("RANK", "0")) == 0:
            subprocess.check_call(["make", "-s", "-j", "8", "-C", os.path.join(ROOT, "cudecomp_amd")], stdout=sys.stderr)
        else:
            t_wait = time.time()
            while not os.path.exists(os.path.join(ROOT, "cudecomp_amd", "lib", "libcudecomp.so")) and time.time() - t_wait < 600:
                time.sleep(1.0)
            time.sleep(2.0)
    import cudecomp_amd as cd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and args.watchdog > 0:
        # A multi-GPU run that wedges (a dead peer, a link that never completes) must not hold the node: after
        # --watchdog seconds every rank reports where it was and exits non-zero.
        import faulthandler
        import threading

        def _expired():
            sys.stderr.write("bench.py: watchdog expired after %d s on rank %d\n" % (args.watchdog, rank))
            faulthandler.dump_traceback(file=sys.stderr)
            sys.stderr.flush()
            os._exit(3)

        wd = threading.Timer(args.watchdog, _expired)
        wd.daemon = True
        wd.start()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    torch.zeros(1, device="cuda")
    if world > 1:
        # control plane of the harness only (barrier, max over ranks); the library moves data itself over
        # RCCL / xGMI and bootstraps from the same RANK / WORLD_SIZE / MASTER_* environment
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()

    n = args.size
    ac = (1, 1, 1) if args.layout == "contiguous" else (0, 0, 0)
    backends = {"nccl": cd.TRANSPOSE_COMM_NCCL, "nccl_pl": cd.TRANSPOSE_COMM_NCCL_PL,
                "peer": cd.TRANSPOSE_COMM_NVSHMEM, "peer_pl": cd.TRANSPOSE_COMM_NVSHMEM_PL,
                "peer_sm": cd.TRANSPOSE_COMM_NVSHMEM_SM, "mpi": cd.TRANSPOSE_COMM_MPI_P2P}
    names = {v: k for k, v in backends.items()}

    if world > 1:
        os.environ["CUDECOMP_ENABLE_PERFORMANCE_REPORT"] = "1"  # per-op local / exchange split, reported below
        # keep exactly the timed calls: the library skips its first WARMUP_SAMPLES calls per op (our warm-up steps;
        # the autotuner's trials are dropped by the library itself) and retains the last SAMPLES
        os.environ["CUDECOMP_PERFORMANCE_REPORT_WARMUP_SAMPLES"] = str(args.warmup)
        os.environ["CUDECOMP_PERFORMANCE_REPORT_SAMPLES"] = str(max(args.steps, 1))
    h = cd.cudecompInit()
    pin_backend, pin_pdims, fallback = args.backend, args.pdims, None
    for attempt in (0, 1):
        autotuned = None
        if world == 1:
            pdims = (1, 1)
            cfg = cd.make_config((n, n, n), pdims, axis_contiguous=ac,
                                 transpose_backend=backends.get(args.backend, cd.TRANSPOSE_COMM_NCCL))
            gd = cd.cudecompGridDescCreate(h, cfg)
        else:
            # BASELINE config 3: "autotuned pgrid".  The library's own autotuner (cudecompGridDescCreate with options)
            # times every process grid x transport through the public transposes and keeps the fastest; a transport that
            # cannot run on this system is dropped by the sweep.  --pdims / --backend pin either choice.
            cfg = cd.make_config((n, n, n), tuple(pin_pdims) if pin_pdims else (0, 0), axis_contiguous=ac,
                                 transpose_backend=backends.get(pin_backend, cd.TRANSPOSE_COMM_NCCL))
            opt = cd.cudecompGridDescAutotuneOptionsSetDefaults()
            opt.dtype = cd.DOUBLE
            opt.n_warmup_trials, opt.n_trials = 2, 3
            opt.autotune_transpose_backend = (pin_backend == "auto")
            for i in range(4):
                opt.transpose_use_inplace_buffers[i] = bool(args.inplace)
            if pin_backend == "auto" or not pin_pdims:
                with c_stdout_to_stderr():  # the sweep logs "CUDECOMP: ..." lines on stdout; keep ours a single JSON line
                    gd = cd.cudecompGridDescCreate(h, cfg, opt)
                autotuned = {"pdims": pin_pdims is None, "backend": pin_backend == "auto"}
            else:
                gd = cd.cudecompGridDescCreate(h, cfg)
            pdims = (cfg.pdims[0], cfg.pdims[1])
        used = names.get(cfg.transpose_comm_backend, cd.cudecompTransposeCommBackendToString(cfg.transpose_comm_backend))

        es = 8
        pinfo = [cd.cudecompGetPencilInfo(h, gd, ax) for ax in range(3)]
        nel = max(p.size for p in pinfo)
        wsz = cd.cudecompGetTransposeWorkspaceSize(h, gd)
        gen = torch.Generator(device="cuda")
        gen.manual_seed(1234 + rank)
        # synthetic payload: random 64-bit patterns (a transpose only relocates bits)
        a = torch.randint(-2**62, 2**62, (nel,), dtype=torch.int64, device="cuda", generator=gen)
        b = a if args.inplace else torch.zeros_like(a)
        work = cd.cudecompMalloc(h, gd, wsz * es)
        stream = torch.cuda.current_stream().cuda_stream
        a0 = a[:pinfo[0].size].clone()  # every cycle must return the X pencil to `a` bit for bit

        def cycle():
            cur, nxt = a, b
            for op in cd.OPS:
                cd.cudecompTranspose(op, h, gd, cur.data_ptr(), nxt.data_ptr(), work, cd.DOUBLE, stream=stream)
                if not args.inplace:
                    cur, nxt = nxt, cur

        for _ in range(args.warmup):
            cycle()
        torch.cuda.synchronize()
        # per-op split (outside the timed region)
        op_ms = []
        cur, nxt = a, b
        for op in cd.OPS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            cd.cudecompTranspose(op, h, gd, cur.data_ptr(), nxt.data_ptr(), work, cd.DOUBLE, stream=stream)
            e1.record()
            torch.cuda.synchronize()
            op_ms.append(e0.elapsed_time(e1))
            if not args.inplace:
                cur, nxt = nxt, cur

        barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            cycle()
        ev1.record()
        torch.cuda.synchronize()
        barrier()
        wall = time.perf_counter() - t0
        dev_ms = ev0.elapsed_time(ev1)
        if world > 1:
            t = torch.tensor([wall, dev_ms], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall, dev_ms = float(t[0]), float(t[1])
        ok = bool(torch.equal(a[:pinfo[0].size], a0))
        del a0
        if world > 1:
            t = torch.tensor([int(ok)], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = bool(t[0])
        if not ok and world > 1 and attempt == 0 and used != "nccl":
            # A transport that measures well but returns wrong data must not be reported: fall back to RCCL on the
            # same process grid and measure again (the JSON line says so).
            if rank == 0:
                sys.stderr.write("bench.py: round-trip checksum FAILED with transport %s on %dx%d; re-running with RCCL\n"
                                 % (used, pdims[0], pdims[1]))
            fallback = "round-trip checksum failed with transport %s" % used
            pin_backend, pin_pdims = "nccl", list(pdims)
            del a, b
            cd.cudecompFree(h, gd, work)
            with c_stdout_to_stderr():
                cd.cudecompGridDescDestroy(h, gd)
            continue

        # where the time goes: per-op averages of [pack | exchange | unpack] recorded by the library (max over ranks)
        split = None
        if world > 1:
            rows = []
            for op in cd.OPS:
                t = cd.cudecompExtGetTransposeTimings(h, gd, op)
                rows.append([t["pack_ms"], t["exchange_ms"], t["unpack_ms"]])
            tt = torch.tensor(rows, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            split = {op: {"pack_ms": round(float(tt[i][0]), 4), "exchange_ms": round(float(tt[i][1]), 4),
                          "unpack_ms": round(float(tt[i][2]), 4)} for i, op in enumerate(cd.OPS)}

        # bytes this rank pushes across the half/half cut of the node per cycle (for the bisection fraction)
        cut = 0
        if world > 1:
            for op in cd.OPS:
                p = cd.cudecompExtGetTransposePlan(h, gd, op, inplace=args.inplace)
                if p.exchange:
                    for d in range(p.nranks):
                        peer = p.member_global_rank[d]
                        if (peer < world // 2) != (rank < world // 2):
                            cut += p.send_cnt[d] * es
            t = torch.tensor([cut], dtype=torch.int64)
            dist.all_reduce(t)
            cut = int(t[0])
        break

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        global_bytes = n ** 3 * es
        value = 4 * global_bytes / (ms_per_step * 1e-3) / 1e9
        # dominant kernel: at 1x1 every hop is ONE launch that reads and writes each element of the pencil once
        # (LDS-tiled permutation for the contiguous layout, streaming row copy for the default layout)
        launches = 4 * args.steps
        alg_bytes = 2 * pinfo[0].size * es
        avg_ms = dev_ms / launches
        roof = None
        if world == 1:
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4),
                    "traffic": measured_traffic(args.layout) if n == 1024 else None,
                    "kernel": "transpose_kernel<8,2,64,64,2,true>" if args.layout == "contiguous" else "rows_kernel<16,true>",
                    "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(avg_ms, 4)}
            if args.layout == "contiguous" and not args.inplace:
                # context for `frac`: what a plain copy of the same pencil reaches on this GPU right now (the library's
                # row-copy kernel on the same buffers: X->Y of a 1x1 grid in the default layout), outside the timed region
                gdc = cd.cudecompGridDescCreate(h, cd.make_config((n, n, n), (1, 1), axis_contiguous=(0, 0, 0)))
                for _ in range(2):
                    cd.cudecompTranspose("XToY", h, gdc, a.data_ptr(), b.data_ptr(), work, cd.DOUBLE, stream=stream)
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record()
                for _ in range(5):
                    cd.cudecompTranspose("XToY", h, gdc, a.data_ptr(), b.data_ptr(), work, cd.DOUBLE, stream=stream)
                c1.record()
                torch.cuda.synchronize()
                copy_rate = alg_bytes / (c0.elapsed_time(c1) / 5 * 1e-3) / 1e9
                cd.cudecompGridDescDestroy(h, gdc)
                roof["copy_rate"] = {"achieved": round(copy_rate, 1), "unit": "GB/s", "kernel": "rows_kernel<16,true>",
                                     "what": "dense copy of the same 8 GiB pencil, same buffers, measured in this run"}
                roof["frac_of_copy_rate"] = round(achieved / copy_rate, 4)
        out = {
            # BASELINE.json's metric, verbatim at its size; `value` is the effective GB/s (4 x global bytes / cycle time),
            # `ms_per_step` the cycle wall time, `xgmi` the bisection fraction (N > 1)
            "metric": ("transpose cycle wall time + effective GB/s (vs xGMI bisection), 1024\u00b3 fp64" if n == 1024 else
                       "transpose cycle wall time + effective GB/s (vs xGMI bisection), %d^3 fp64" % n),
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d^3 fp64 X->Y->Z->Y->X transpose cycle, %dx%d process grid, %s layout, %s"
                                   % (n, pdims[0], pdims[1],
                                      "all-axis-contiguous" if args.layout == "contiguous" else "default (X fastest)",
                                      "in-place" if args.inplace else "out-of-place"),
                       "pdims": list(pdims), "transport": used, "autotuned": autotuned,
                       "per_op_ms": [round(x, 4) for x in op_ms], "per_op_split": split,
                       "round_trip_checksum_ok": bool(ok), "fallback": fallback},
            "roofline": roof,
        }
        if world > 1:
            # nominal xGMI link: 153.6 GB/s counting both directions (task statement); (N/2)^2 links cross the cut
            link = 153.6
            bis = (world // 2) ** 2 * link
            out["xgmi"] = {"cut_bytes_per_cycle": cut, "cut_GBps": round(cut / (ms_per_step * 1e-3) / 1e9, 1),
                           "bisection_GBps_nominal": bis,
                           "frac_of_bisection": round(cut / (ms_per_step * 1e-3) / 1e9 / bis, 4)}
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.layout)
        print(json.dumps(out))

    cd.cudecompFree(h, gd, work)
    with c_stdout_to_stderr():  # the library prints its performance summary here when enabled
        cd.cudecompGridDescDestroy(h, gd)
    cd.cudecompFinalize(h)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
