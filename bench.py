#!/usr/bin/env python3
"""bench.py -- the reference's headline measurement on MI355X: wall time and effective bandwidth of one
X->Y->Z->Y->X transpose cycle of a 1024^3 fp64 array (BASELINE.json), through libcudecomp.so's C ABI.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full cycle (4 transposes) on device-resident synthetic data.  Protocol (the
reference autotuner's, src/autotune.cc:541-636): W warm-up cycles, then K timed cycles bracketed by barriers + device
syncs, every cycle also bracketed by its own HIP events -> min / max / avg / std over cycles and ranks; out of place
AND in place.  Rank 0 prints ONE JSON line.

  value        effective GB/s = 4 * global array bytes / cycle time, out of place (whole job)
  stats        per-cycle device times (HIP events on the library's stream), out of place and in place
  roofline     dominant kernel of the cycle (N = 1) vs the HBM roofline; its name comes from the library
  cpu_baseline the host-MPI CPU path (oracle/cpu_mpi_cycle: the oracle's pack / unpack around MPI_Alltoallv, one rank
               per core) on the same workload, rank 0, N = 1 only
  extra        N = 1 only: BASELINE config 4 (1024^3 complex<fp32> 3-D FFT forward + inverse, benchmark/fft3d_benchmark)
               and config 5 (halo update of its per-rank pencil), measured after the timed region; `dtypes`: the cycle at
               8-GiB pencils for fp32 / complex64 / complex128, both layouts, per-hop kernel and roofline fraction
  N > 1        the fixed 1024^3 problem on N ranks ("strong"); first over RCCL (process grid autotuned), then -- after
               a preflight that tries every one-sided transport on a small grid and reports pass/fail per transport
               to stderr -- with the library's autotuner choosing among all transports that passed; the faster valid
               result is reported.  If the one-sided phase hangs or fails the RCCL line is printed (watchdog).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy ceiling
LINK_GBPS_PER_DIRECTION = 76.8    # one direction of one xGMI link (153.6 GB/s counting both); replaced by the library's
                                  # start-up measurement when the ranks sit on different GPUs


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=1024, help="global grid is size^3")
    ap.add_argument("--layout", choices=["contiguous", "default"], default="contiguous",
                    help="contiguous: every pencil axis-contiguous (the layout of the reference's published "
                         "1024-class numbers, every hop permutes); default: X fastest everywhere")
    ap.add_argument("--backend", default="auto", help="auto | nccl | nccl_pl | peer | peer_pl | peer_sm | mpi")
    ap.add_argument("--pdims", type=int, nargs=2, default=None)
    ap.add_argument("--inplace", action="store_true", help="make the in-place cycle the headline value")
    ap.add_argument("--watchdog", type=int, default=600, help="multi-GPU runs: give up after this many seconds (0 = never)")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="grid edge of the CPU-baseline sample (0 = skip, -1 = the full size if host memory allows, else half)")
    ap.add_argument("--no-extras", action="store_true", help="skip the config-4 / config-5 records (N = 1)")
    return ap.parse_args()


def measured_traffic(layout):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 +
    WRITE_SIZE, see profiles/*_pmc_summary.json) and the file they come from: counters cannot be collected inside the
    timed loop, so the live line repeats the profile and says so."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    for f in reversed(files):
        try:
            with open(f) as fh:
                return int(json.load(fh)["layouts"][layout]["hbm_traffic_bytes_per_launch"]), os.path.relpath(f, ROOT)
        except (KeyError, ValueError, OSError):
            continue
    return None, None


class c_stdout_to_stderr:
    """Route C-level stdout (the library's autotune log) to stderr for the duration of the block; the text is kept in
    `.text` so that the sweep's per-candidate results can be reported (parse_sweep)."""

    def __enter__(self):
        import tempfile
        sys.stdout.flush()
        self.saved = os.dup(1)
        self.tmp = tempfile.TemporaryFile(mode="w+b")
        os.dup2(self.tmp.fileno(), 1)
        self.text = ""
        return self

    def __exit__(self, *exc):
        import ctypes
        ctypes.CDLL(None).fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)
        self.tmp.seek(0)
        self.text = self.tmp.read().decode(errors="replace")
        self.tmp.close()
        sys.stderr.write(self.text)
        sys.stderr.flush()


def parse_sweep(text):
    """Every candidate the library's autotuner measured, from its log (reference format, src/autotune.cc:639-668):
    [{"pdims": [r, c], "transport": name, "avg_ms": weighted average or None, "status": "measured"|"skipped"|"failed",
    "model_ms": the xGMI-mesh estimate when printed}]."""
    import re
    out, cur = [], None
    for line in text.splitlines():
        m = re.match(r"CUDECOMP:\s+grid: (\d+) x (\d+), (?:halo )?backend: (.*?)\s*$", line)
        if m:
            cur = {"pdims": [int(m.group(1)), int(m.group(2))], "transport": m.group(3), "avg_ms": None, "status": "measured"}
            out.append(cur)
            continue
        if cur is None:
            continue
        if "(failed, skipped)" in line:
            cur["status"] = "failed"
        elif "(skipped)" in line:
            cur["status"] = "skipped"
        m = re.match(r"CUDECOMP:\s+min/max/avg/std \[ms\]: ([\d.eE+-]+)/([\d.eE+-]+)/([\d.eE+-]+)/([\d.eE+-]+) \(weighted\)", line)
        if m:
            cur["avg_ms"] = round(float(m.group(3)), 4)
        m = re.match(r"CUDECOMP:\s+xGMI-mesh model estimate \[ms\]: ([\d.eE+-]+)", line)
        if m:
            cur["model_ms"] = round(float(m.group(1)), 4)
    return out


def gpus_on_this_host():
    """GPU agents the kernel driver lists (no runtime needed: this is asked before torch is imported)."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(f) as fh:
                for line in fh:
                    if line.startswith("simd_count") and int(line.split()[1]) > 0:
                        n += 1
        except (OSError, ValueError):
            pass
    return n


def stats_of(xs):
    xs = [float(x) for x in xs]
    avg = sum(xs) / len(xs)
    return {"min": round(min(xs), 4), "max": round(max(xs), 4), "avg": round(avg, 4),
            "std": round(math.sqrt(sum((x - avg) ** 2 for x in xs) / len(xs)), 4), "n": len(xs)}


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------------------------
def host_memory_gib():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / 2**20
    except OSError:
        pass
    return 0.0


def cpu_baseline(sample, size, layout):
    """The CPU path timed on this host's cores, fp64, same layout: the host-MPI path (oracle/cpu_mpi_cycle: pack ->
    MPI_Alltoallv -> unpack with one rank per core, up to 64) when an MPI installation is present, else the
    single-process oracle on one core.  sample < 0: the benchmark's own size when ~5x its array fits in host memory
    (input, output, workspace, MPI buffers), else half the edge (1/8 of the volume), which the record states."""
    if sample < 0:
        need_gib = 5 * size ** 3 * 8 / 2**30
        sample = size if host_memory_gib() > need_gib + 8 else size // 2
    mpi = cpu_baseline_mpi(sample, size, layout)
    if mpi is not None:
        return mpi
    import numpy as np
    from oracle import oracle as orc
    sample = min(sample, 512)  # one core: keep it to tens of seconds
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
            "ranks": 1, "grid": [1, 1], "cpu_model": cpu_model(), "host_cores": os.cpu_count(), "mpi_version": None,
            "pinning": "none (single process)", "sample_edge": sample,
            "sample": "%d^3 fp64 X->Y->Z->Y->X cycle (%s of the benchmark's volume), 1x1 grid, %s layout, out-of-place, "
                      "one cycle, %.1f s on one core" % (sample, "all" if sample == size else "1/%d" % (size // sample) ** 3,
                                                         layout, dt)}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def run_cpu_mpi_cycle(mpirun, exe, env, ranks, grid, n, contiguous, warm, timed, kind=1, timeout=300):
    """One launch of oracle/cpu_mpi_cycle with the ranks bound to cores; the record it prints, plus the wall time."""
    import subprocess
    t0 = time.perf_counter()
    cmd = [mpirun, "-bind-to", "core", "-np", str(ranks), exe, str(n), str(grid[0]), str(grid[1]), "1" if contiguous else "0",
           str(warm), str(timed), str(kind)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    rec["wall_s"] = time.perf_counter() - t0
    return rec


def cycle_stats(rec, scale=1.0, digits=4):
    return {"avg": round(rec["cycle_s"] * scale, digits), "min": round(rec.get("cycle_s_min", rec["cycle_s"]) * scale, digits),
            "max": round(rec.get("cycle_s_max", rec["cycle_s"]) * scale, digits), "std": round(rec.get("cycle_s_std", 0.0) * scale, digits)}


def cpu_baseline_mpi(sample, size, layout):
    """mpirun -np R oracle/cpu_mpi_cycle ... ; None if there is no MPI here or anything goes wrong.  Two records of the same
    workload: one rank per host core (up to 64; `value`) and -- BASELINE.md section 3 -- one rank per GPU SLOT of the node
    the 8-GPU configuration runs on (8 ranks, 2x4 grid, config 3's decomposition), both with the ranks bound to cores."""
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
        contiguous = layout == "contiguous"
        warm, timed = 2, 5  # (about 32 s at 1024^3 on 64 cores: the first cycle after start-up runs 20-30 % slow)
        rec = run_cpu_mpi_cycle(mpirun, exe, env, ranks, (pr, pc), sample, contiguous, warm, timed)
        if not rec["round_trip_ok"]:
            return None
        frac = "the benchmark's own size" if sample == size else "1/%d of the benchmark's volume" % (size // sample) ** 3
        # one rank per GPU slot (8 ranks, 2x4): a cycle takes several times longer, so 1 warm-up + 2 timed cycles
        slots = None
        if cores >= 8:
            try:
                r8 = run_cpu_mpi_cycle(mpirun, exe, env, 8, (2, 4), sample, contiguous, 1, 2, timeout=400)
                slots = {"value": round(r8["gbps"], 4), "unit": "GB/s", "ranks": 8, "grid": [2, 4], "cores": 8,
                         "cycle_s": cycle_stats(r8), "warmup": 1, "timed": 2, "sample_edge": sample,
                         "distinct_cores": r8.get("distinct_cpus"), "bytes_per_rank": r8.get("bytes_per_rank"),
                         "round_trip_ok": r8["round_trip_ok"], "wall_s": round(r8["wall_s"], 1),
                         "what": "the same %d^3 fp64 cycle with ONE host-MPI rank per GPU slot of the 8-GPU node "
                                 "(BASELINE.md section 3): 8 ranks, config 3's 2x4 grid" % sample}
            except Exception as e8:
                slots = {"unavailable": str(e8)[:120]}
        # BASELINE config 1 at its own shape: 256^3 fp32 slab decomposition on 2 host-MPI ranks, 3 + 5 cycles (< 1 s)
        config1 = {}
        for grid in ((2, 1), (1, 2)):
            try:
                r1 = run_cpu_mpi_cycle(mpirun, exe, env, 2, grid, 256, False, 3, 5, kind=0, timeout=120)
                config1["%dx%d" % grid] = {"GBps": round(r1["gbps"], 3), "cycle_ms": cycle_stats(r1, 1e3, 3),
                                           "round_trip_ok": r1["round_trip_ok"]}
            except Exception as e1:
                config1["%dx%d" % grid] = "unavailable: %s" % str(e1)[:80]
        return {"value": round(rec["gbps"], 4), "unit": "GB/s", "cores": ranks, "kind": "port",
                # provenance as structured keys (CPU model, core count, MPI version next to the number)
                "ranks": ranks, "grid": [pr, pc], "cpu_model": cpu_model(), "host_cores": cores,
                "mpi_version": rec.get("mpi_version"), "pinning": "mpirun -bind-to core",
                "distinct_cores": rec.get("distinct_cpus"), "bytes_per_rank": rec.get("bytes_per_rank"),
                "sample_edge": sample, "warmup": warm, "timed": timed,
                "cycle_s": cycle_stats(rec),
                "one_rank_per_gpu_slot": slots,
                "config1_256cube_fp32_2_ranks": dict(config1, what="BASELINE.json configs[0]: 256^3 fp32 slab, 2-rank host-MPI "
                                                     "a2a CPU path (oracle pack/unpack + MPI_Alltoallv), 3 warm-up + 5 timed cycles"),
                "sample": "%d^3 fp64 X->Y->Z->Y->X cycle on host memory (%s), %d MPI ranks (%dx%d grid, one per core, bound; %s "
                          "shared-memory MPI_Alltoallv), %s layout, out-of-place, %.3f s per cycle, %d warm-up + %d timed "
                          "cycles, %.1f s in total" % (sample, frac, ranks, pr, pc, rec.get("mpi_version", "MPI"), layout,
                                                       rec["cycle_s"], warm, timed, rec["wall_s"])}
    except Exception as e:  # the baseline is a courtesy number: never let it take the benchmark down
        sys.stderr.write("bench.py: host-MPI CPU baseline unavailable (%s); using the single-core oracle\n" % e)
        return None


# ---------------------------------------------------------------------------------------------------------------
# device helpers
# ---------------------------------------------------------------------------------------------------------------
class _Raw:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class Pencils:
    """Two data pencils and the workspace of one grid descriptor, all from cudecompMalloc (memory every rank of the
    node has mapped: the one-sided transports may then write straight into the output pencil)."""

    def __init__(self, cd, torch, h, gd, seed):
        self.cd, self.h, self.gd = cd, h, gd
        self.pinfo = [cd.cudecompGetPencilInfo(h, gd, ax) for ax in range(3)]
        nel = max(p.size for p in self.pinfo)
        self.ptrs = [cd.cudecompMalloc(h, gd, nel * 8) for _ in range(2)]
        self.a, self.b = [torch.as_tensor(_Raw(p, nel * 8), device="cuda").view(torch.int64) for p in self.ptrs]
        self.work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * 8)
        gen = torch.Generator(device="cuda")
        gen.manual_seed(seed)
        # synthetic payload: random 64-bit patterns (a transpose only relocates bits)
        self.a.copy_(torch.randint(-2**62, 2**62, (nel,), dtype=torch.int64, device="cuda", generator=gen))
        self.b.zero_()
        self.x0 = self.a[:self.pinfo[0].size].clone()  # every cycle must return the X pencil to `a` bit for bit

    def free(self):
        import torch
        torch.cuda.synchronize()
        self.a = self.b = self.x0 = None
        for p in self.ptrs + [self.work]:
            self.cd.cudecompFree(self.h, self.gd, p)


def run_cycles(cd, torch, dist, world, h, gd, pen, inplace, warmup, steps, stream):
    """W warm-up + K timed cycles.  Returns wall ms (barrier + sync bracket, max over ranks), per-cycle device ms of
    this rank, per-op ms of one extra cycle and whether the X pencil came back bit for bit."""
    def barrier():
        if world > 1:
            dist.barrier()

    def cycle():
        cur, nxt = pen.a, (pen.a if inplace else pen.b)
        for op in cd.OPS:
            cd.cudecompTranspose(op, h, gd, cur.data_ptr(), nxt.data_ptr(), pen.work, cd.DOUBLE, stream=stream)
            if not inplace:
                cur, nxt = nxt, cur

    for _ in range(warmup):
        cycle()
    torch.cuda.synchronize()
    op_ms, op_kernels = [], []
    cur, nxt = pen.a, (pen.a if inplace else pen.b)
    for op in cd.OPS:  # per-op split, outside the timed region
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        cd.cudecompTranspose(op, h, gd, cur.data_ptr(), nxt.data_ptr(), pen.work, cd.DOUBLE, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        op_ms.append(e0.elapsed_time(e1))
        op_kernels.append(cd.cudecompExtLastKernelName())
        if not inplace:
            cur, nxt = nxt, cur
    # the same split INSIDE a sustained sequence (no host synchronisation between the hops; outside the timed region): events
    # between the hops of 3 back-to-back cycles after one more warm-up cycle.  Isolated hops start on an idle memory system;
    # sustained ones inherit the previous hop's tail (dirty lines of the 256 MB memory-side cache still draining to HBM).
    op_sus = [0.0] * len(cd.OPS)
    sus_cycles = 3
    cycle()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(sus_cycles * len(cd.OPS) + 1)]
    marks[0].record()
    for c in range(sus_cycles):
        cur, nxt = pen.a, (pen.a if inplace else pen.b)
        for i, op in enumerate(cd.OPS):
            cd.cudecompTranspose(op, h, gd, cur.data_ptr(), nxt.data_ptr(), pen.work, cd.DOUBLE, stream=stream)
            marks[c * len(cd.OPS) + i + 1].record()
            if not inplace:
                cur, nxt = nxt, cur
    torch.cuda.synchronize()
    for c in range(sus_cycles):
        for i in range(len(cd.OPS)):
            op_sus[i] += marks[c * len(cd.OPS) + i].elapsed_time(marks[c * len(cd.OPS) + i + 1]) / sus_cycles
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(steps):
        cycle()
        ev[k + 1].record()
    torch.cuda.synchronize()
    barrier()
    wall_ms = (time.perf_counter() - t0) * 1e3
    cyc = [ev[k].elapsed_time(ev[k + 1]) for k in range(steps)]
    ok = bool(torch.equal(pen.a[:pen.pinfo[0].size], pen.x0))
    if world > 1:
        t = torch.tensor([wall_ms, -float(ok)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall_ms, ok = float(t[0]), float(t[1]) == -1.0
        allc = [None] * world
        dist.all_gather_object(allc, cyc)
        cyc_all = [x for c in allc for x in c]
    else:
        cyc_all = cyc
    return {"wall_ms": wall_ms, "cycle_ms": cyc, "cycle_ms_all_ranks": cyc_all, "op_ms": op_ms, "op_ms_sustained": op_sus, "op_kernels": op_kernels,
            "ok": ok}


def extras_single_gpu(cd, torch, h, stream):
    """BASELINE configs 4 and 5 on one GPU, after the timed region: (C4) 1024^3 complex<fp32> 3-D FFT, forward + inverse,
    hipFFT + the library's transposes (benchmark/fft3d_benchmark, the counterpart of the reference's benchmark.cu:
    GFLOP/s = 5 N log2 N per direction, max-abs round-trip residual); (C5) halo update (width 2, periodic) of the
    per-rank X pencil 2048 x 1024 x 256 fp64 of the 2x4 decomposition, per dim, against the algorithmic bytes
    2 (faces) x 2 (read + write) x face bytes."""
    import subprocess
    out = {}
    try:
        exe = os.path.join(ROOT, "benchmark", "fft3d_benchmark")
        if not os.path.exists(exe):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "benchmark")], stdout=sys.stderr, stderr=sys.stderr)
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        def fft_run(extra):
            r = subprocess.run([exe, "--gx", "1024", "--gy", "1024", "--gz", "1024", "--pr", "1", "--pc", "1", "--warmup", "3",
                                "--trials", "5", "-o", "--no-spectrum-check"] + extra, env=env, capture_output=True, text=True,
                               timeout=300)
            rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            return {"ms_per_direction": rec["ms_avg"], "ms_min": rec["ms_min"], "ms_max": rec["ms_max"], "gflops": rec["gflops"],
                    "roundtrip_max_abs_residual": rec["roundtrip_max_abs_err"], "tolerance": rec["tolerance"], "ok": rec["ok"],
                    "slab": rec.get("slab"), "mode": rec.get("mode")}
        # three 1-D passes + the library's four transposes (what a pencil grid runs; the figure of earlier rounds) ...
        passes = fft_run(["--no-slab-opt"])
        out["config4_fft"] = dict(passes, workload="1024^3 complex<fp32> 3-D FFT forward + inverse, 1x1 grid, axis-contiguous "
                                  "pencils, out of place, 3 warm-up + 5 timed; x / y / z line passes with the library's "
                                  "transposes in between (--no-slab-opt)")
        # ... and what the reference's benchmark runs on a 1x1 grid: its slab shortcut, ONE 3-D FFT and no transposes
        # (benchmark.cu:340-345); and the real-to-complex flavour (:238-330)
        try:
            out["config4_fft"]["single_3d_fft_shortcut"] = fft_run([])
            out["config4_fft"]["r2c_three_passes"] = fft_run(["--no-slab-opt", "--r2c"])
        except Exception as e2:
            out["config4_fft"]["single_3d_fft_shortcut"] = {"error": str(e2)[:200]}
    except Exception as e:
        out["config4_fft"] = {"error": str(e)[:200]}
    try:
        gdims, halo = (2048, 1024, 256), (2, 2, 2)
        gd = cd.cudecompGridDescCreate(h, cd.make_config(gdims, (1, 1)))
        p = cd.cudecompGetPencilInfo(h, gd, 0, halo)
        data = torch.zeros(p.size, dtype=torch.float64, device="cuda")
        work = cd.cudecompMalloc(h, gd, max(cd.cudecompGetHaloWorkspaceSize(h, gd, 0, halo), 1) * 8)
        shape, res = list(p.shape), {}
        for dim in range(3):
            face = halo[dim] * shape[(dim + 1) % 3] * shape[(dim + 2) % 3]
            for _ in range(3):
                cd.cudecompUpdateHalos(0, h, gd, data.data_ptr(), work, cd.DOUBLE, halo, (1, 1, 1), dim, stream=stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                cd.cudecompUpdateHalos(0, h, gd, data.data_ptr(), work, cd.DOUBLE, halo, (1, 1, 1), dim, stream=stream)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            res["dim%d" % dim] = {"ms": round(ms, 4), "face_MiB": round(face * 8 / 2**20, 2),
                                  "GBps": round(2 * 2 * face * 8 / ms / 1e6, 1), "kernel": cd.cudecompExtLastKernelName()}
        cd.cudecompFree(h, gd, work)
        cd.cudecompGridDescDestroy(h, gd)
        del data
        out["config5_halo"] = {"workload": "halo update, width 2, periodic, of the per-rank X pencil 2048x1024x256 fp64 "
                                           "(+halo: 2052x1028x260) of config 5 on a single rank (wrap-around face copies), "
                                           "3 warm-up + 5 timed per dim; GB/s = 2 faces x (read + write) x face bytes / time",
                               "per_dim": res}
    except Exception as e:
        out["config5_halo"] = {"error": str(e)[:200]}
    try:
        # transposes ONTO halo-carrying pencils (solvers that keep halos around their pencils): 1 x 1 grid, both layouts, 1024^3
        # fp64 with a halo of one cell on every pencil.  Default layout: every hop is one row copy of whole rows (rows_dense_kernel:
        # whole cache lines across the row ends); axis-contiguous: permutations onto rows off the 64-byte grid (window kernel).
        halo, n, res = (1, 1, 1), 1024, {}
        for layout, ac in (("default", (0, 0, 0)), ("contiguous", (1, 1, 1))):
            gd = cd.cudecompGridDescCreate(h, cd.make_config((n, n, n), (1, 1), axis_contiguous=ac))
            nel = max(cd.cudecompGetPencilInfo(h, gd, ax, halo).size for ax in range(3))
            a = torch.zeros(nel, dtype=torch.float64, device="cuda")
            b = torch.zeros(nel, dtype=torch.float64, device="cuda")
            work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * 8)
            ops = {}
            for op in cd.OPS:
                for _ in range(2):
                    cd.cudecompTranspose(op, h, gd, a.data_ptr(), b.data_ptr(), work, cd.DOUBLE, halo, halo, None, None, stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    cd.cudecompTranspose(op, h, gd, a.data_ptr(), b.data_ptr(), work, cd.DOUBLE, halo, halo, None, None, stream)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 5
                ops[op] = {"ms": round(ms, 4), "frac": round(2 * 8 * n**3 / ms / 1e6 / HBM_PEAK_GBPS, 4),
                           "kernel": cd.cudecompExtLastKernelName()}
            res[layout] = ops
            cd.cudecompFree(h, gd, work)
            cd.cudecompGridDescDestroy(h, gd)
            del a, b
        out["halo_pencil_transposes"] = {"workload": "transposes onto halo-carrying pencils: 1024^3 fp64, halo (1,1,1) on every pencil, "
                                                     "1x1 grid, out of place, 2 warm-up + 5 timed per op; frac = 2 x 8 GiB / ms / 8 TB/s",
                                         "per_layout": res}
        # the same on config 5's per-rank pencil shape (2048 x 1024 x 256 fp64, halo 2: 2052-wide rows), axis-contiguous
        gdims5, halo5 = (2048, 1024, 256), (2, 2, 2)
        gd = cd.cudecompGridDescCreate(h, cd.make_config(gdims5, (1, 1), axis_contiguous=(1, 1, 1)))
        nel = max(cd.cudecompGetPencilInfo(h, gd, ax, halo5).size for ax in range(3))
        a = torch.zeros(nel, dtype=torch.float64, device="cuda")
        b = torch.zeros(nel, dtype=torch.float64, device="cuda")
        work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * 8)
        ops, moved = {}, 2 * 8 * gdims5[0] * gdims5[1] * gdims5[2]
        for op in cd.OPS:
            for _ in range(2):
                cd.cudecompTranspose(op, h, gd, a.data_ptr(), b.data_ptr(), work, cd.DOUBLE, halo5, halo5, None, None, stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                cd.cudecompTranspose(op, h, gd, a.data_ptr(), b.data_ptr(), work, cd.DOUBLE, halo5, halo5, None, None, stream)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            ops[op] = {"ms": round(ms, 4), "frac": round(moved / ms / 1e6 / HBM_PEAK_GBPS, 4), "kernel": cd.cudecompExtLastKernelName()}
        cd.cudecompFree(h, gd, work)
        cd.cudecompGridDescDestroy(h, gd)
        del a, b
        out["halo_pencil_transposes"]["config5_pencil_contiguous"] = {
            "workload": "2048 x 1024 x 256 fp64, halo (2,2,2) on every pencil, 1x1 grid, axis-contiguous, out of place; frac = 2 x 4 GiB / ms / 8 TB/s",
            "per_op": ops}
    except Exception as e:
        out["halo_pencil_transposes"] = {"error": str(e)[:200]}
    return out


def dtype_table(cd, torch, h, stream, only=None):
    """The other three element types the reference instantiates (src/cudecomp_kernels.cu:29-46; its published sweeps
    are float and double) at the benchmark's pencil size: 8-GiB pencils on a 1x1 grid, out of place, both layouts,
    2 warm-up + 5 timed cycles with HIP events around every transpose.  Per hop: ms (median; min / max beside it), achieved GB/s = 2 x pencil bytes / ms,
    fraction of the 8 TB/s HBM peak and the kernel the library launched for it."""
    rows = []
    cases = [("fp32", cd.FLOAT, 4, (2048, 1024, 1024)), ("complex64", cd.FLOAT_COMPLEX, 8, (1024, 1024, 1024)),
             ("complex128", cd.DOUBLE_COMPLEX, 16, (1024, 1024, 512))]
    for name, dt, es, gdims in cases:
        if only and name not in only:
            continue
        for layout, ac in (("contiguous", (1, 1, 1)), ("default", (0, 0, 0))):
            row = {"dtype": name, "element_bytes": es, "gdims": list(gdims), "layout": layout}
            try:
                gd = cd.cudecompGridDescCreate(h, cd.make_config(gdims, (1, 1), axis_contiguous=ac))
                nbytes = gdims[0] * gdims[1] * gdims[2] * es
                gen = torch.Generator(device="cuda")
                gen.manual_seed(77)
                a = torch.randint(-2**62, 2**62, (nbytes // 8,), dtype=torch.int64, device="cuda", generator=gen)
                keep = a.clone()
                b = torch.zeros_like(a)
                work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * es)
                ms = {op: [] for op in cd.OPS}
                kernels = {}
                for it in range(2 + 5):
                    cur, nxt = a, b
                    for op in cd.OPS:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        cd.cudecompTranspose(op, h, gd, cur.data_ptr(), nxt.data_ptr(), work, dt, stream=stream)
                        e1.record()
                        kernels[op] = cd.cudecompExtLastKernelName()
                        torch.cuda.synchronize()
                        if it >= 2:
                            ms[op].append(e0.elapsed_time(e1))
                        cur, nxt = nxt, cur
                row["round_trip_ok"] = bool(torch.equal(a, keep))
                per_op = []
                for op in cd.OPS:
                    avg = sorted(ms[op])[len(ms[op]) // 2]  # median of the timed cycles (one hiccup in five must not halve a rate)
                    gbps = 2 * nbytes / (avg * 1e-3) / 1e9
                    per_op.append({"op": op, "ms": round(avg, 4), "ms_min": round(min(ms[op]), 4), "ms_max": round(max(ms[op]), 4),
                                   "GBps": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBPS, 4), "kernel": kernels[op]})
                row["per_op"] = per_op
                row["cycle_ms"] = round(sum(o["ms"] for o in per_op), 4)  # SUM OF HOPS timed one at a time, not a sustained cycle
                row["cycle_ms_is"] = "sum of the per-hop medians (each hop timed alone)"
                row["min_frac"] = min(o["frac"] for o in per_op)
                del a, b, keep
                cd.cudecompFree(h, gd, work)
                cd.cudecompGridDescDestroy(h, gd)
                torch.cuda.empty_cache()
            except Exception as e:
                row["error"] = str(e)[:200]
            rows.append(row)
    return {"workload": "X->Y->Z->Y->X cycle of an 8-GiB pencil per element type, 1x1 grid, out of place, 2 warm-up + 5 timed "
                        "cycles, HIP events around every transpose; ms = median per hop; frac = 2 x pencil bytes / ms / 8 TB/s", "rows": rows}


# ---------------------------------------------------------------------------------------------------------------
_RESULT_FD = [None]


def emit(record):
    """The ONE JSON line goes to the process's original stdout; everything else this process (or the library: autotune
    log, performance report) prints on stdout is routed to stderr so that the line is the only thing a parser sees."""
    os.write(_RESULT_FD[0], (json.dumps(record) + "\n").encode())


def launcher_command(argv, gpus, port):
    """`python bench.py --gpus N` without a launcher around it: the command that runs it the way the driver does -- one rank
    per GPU under torch.distributed.run on this node, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr",
            "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args):
    """Called as a plain command with --gpus N > 1 (no RANK / WORLD_SIZE in the environment): start the N ranks ourselves and
    hand their ONE JSON line through.  Returns the exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = launcher_command(sys.argv[1:], args.gpus, port)
    sys.stderr.write("bench.py: --gpus %d without a launcher: running %s\n" % (args.gpus, " ".join(cmd)))
    limit = (args.watchdog + 300) if args.watchdog > 0 else None
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, timeout=limit)
    except subprocess.TimeoutExpired as e:
        sys.stderr.write("bench.py: the launched ranks did not finish within %d s\n" % limit)
        out, rc = (e.stdout or b""), 3
    else:
        out, rc = r.stdout, r.returncode
    lines = [l for l in out.decode(errors="replace").splitlines() if l.startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
        return 0 if rc == 0 else rc
    return rc or 3


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))
    sys.stdout.flush()
    _RESULT_FD[0] = os.dup(1)
    os.dup2(2, 1)
    # dmabuf IPC (the only mode the pool's hosts support) must be selected before the HIP runtime starts
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1:
        # a peer that never shows up must cost a minute, not the library default of two, per attempt (device-side waits, host rendezvous)
        os.environ.setdefault("CUDECOMP_PEER_TIMEOUT", "60")
        os.environ.setdefault("CUDECOMP_BOOTSTRAP_TIMEOUT", "120")
        # pencils of this benchmark live in cudecompMalloc memory: let the autotuner measure NVSHMEM_SM's direct put
        os.environ.setdefault("CUDECOMP_AUTOTUNE_LIBRARY_BUFFERS", "1")
        # every candidate of the sweep is reported with the analytic prior next to its measurement (config.also_measured)
        os.environ.setdefault("CUDECOMP_AUTOTUNE_PRINT_MODEL", "1")
        # (no GPU_MAX_HW_QUEUES setting for a GPU per rank any more: the one-sided transport never parks a wait kernel on a
        # copy stream and, with kernel copies, uses ONE copy stream beside the caller's, so the runtime's default of 4
        # hardware queues per process is enough.)  Ranks that SHARE a device -- flow checks on a one-GPU box, never a
        # measurement -- must not exceed its hardware queue slots: two queues per process then.
        if gpus_on_this_host() * 5 < world:
            os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
    import torch
    import torch.distributed as dist

    if not os.path.exists(os.path.join(ROOT, "cudecomp_amd", "lib", "libcudecomp.so")):
        # binaries are git-ignored; build once (rank 0 of a multi-GPU launch, the others wait for the file)
        import subprocess
        if rank == 0:
            subprocess.check_call(["make", "-s", "-j", "8", "-C", os.path.join(ROOT, "cudecomp_amd")], stdout=sys.stderr)
        else:
            t_wait = time.time()
            while not os.path.exists(os.path.join(ROOT, "cudecomp_amd", "lib", "libcudecomp.so")) and time.time() - t_wait < 600:
                time.sleep(1.0)
            time.sleep(2.0)
    import cudecomp_amd as cd

    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    fallback_line = [None]  # multi-GPU: the RCCL result, printed by the watchdog if a later phase wedges
    if world > 1 and args.watchdog > 0:
        import faulthandler
        import threading

        def _expired():
            sys.stderr.write("bench.py: watchdog expired after %d s on rank %d\n" % (args.watchdog, rank))
            faulthandler.dump_traceback(file=sys.stderr)
            sys.stderr.flush()
            if rank == 0 and fallback_line[0] is not None:
                fallback_line[0]["config"]["fallback"] = "watchdog expired during the one-sided transports; RCCL result reported"
                emit(fallback_line[0])
                os._exit(0)
            os._exit(0 if fallback_line[0] is not None else 3)

        wd = threading.Timer(args.watchdog, _expired)
        wd.daemon = True
        wd.start()
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    torch.zeros(1, device="cuda")
    if world > 1:
        # control plane of the harness only (barrier, max over ranks); the library moves data itself over
        # RCCL / xGMI and bootstraps from the same RANK / WORLD_SIZE / MASTER_* environment
        dist.init_process_group("gloo", rank=rank, world_size=world)

    n, es = args.size, 8
    ac = (1, 1, 1) if args.layout == "contiguous" else (0, 0, 0)
    backends = {"nccl": cd.TRANSPOSE_COMM_NCCL, "nccl_pl": cd.TRANSPOSE_COMM_NCCL_PL,
                "peer": cd.TRANSPOSE_COMM_NVSHMEM, "peer_pl": cd.TRANSPOSE_COMM_NVSHMEM_PL,
                "peer_sm": cd.TRANSPOSE_COMM_NVSHMEM_SM, "mpi": cd.TRANSPOSE_COMM_MPI_P2P}
    names = {v: k for k, v in backends.items()}
    names.update({cd.TRANSPOSE_COMM_MPI_P2P_PL: "mpi_pl", cd.TRANSPOSE_COMM_MPI_A2A: "mpi_a2a"})
    lib_names = {"nccl": "NCCL", "nccl_pl": "NCCL_PL", "peer": "NVSHMEM", "peer_pl": "NVSHMEM_PL", "peer_sm": "NVSHMEM_SM",
                 "mpi": "MPI_P2P", "mpi_pl": "MPI_P2P_PL", "mpi_a2a": "MPI_A2A"}
    stream = torch.cuda.current_stream().cuda_stream

    if world > 1:
        os.environ["CUDECOMP_ENABLE_PERFORMANCE_REPORT"] = "1"  # per-op local / exchange split, reported below
        os.environ["CUDECOMP_PERFORMANCE_REPORT_WARMUP_SAMPLES"] = str(args.warmup + 1)
        os.environ["CUDECOMP_PERFORMANCE_REPORT_SAMPLES"] = str(max(args.steps, 1))
    h = cd.cudecompInit()

    def all_ok(flag):
        if world == 1:
            return bool(flag)
        t = torch.tensor([int(bool(flag))], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t[0])

    def make_desc(pin_backend, pin_pdims, sweep_backends):
        """(grid descriptor, config, what the autotuner decided).  sweep_backends: candidate list for the autotuner or None."""
        if world == 1:
            cfg = cd.make_config((n, n, n), (1, 1), axis_contiguous=ac,
                                 transpose_backend=backends.get(pin_backend, cd.TRANSPOSE_COMM_NCCL))
            return cd.cudecompGridDescCreate(h, cfg), cfg, None
        tune_backend = sweep_backends is not None
        # the RCCL phase must not touch the one-sided transport at all (it is the safety net): with both backends of
        # the descriptor on RCCL the library creates no IPC mappings, no shared board and runs no link probe
        rccl_only = not tune_backend and pin_backend in ("nccl", "nccl_pl")
        cfg = cd.make_config((n, n, n), tuple(pin_pdims) if pin_pdims else (0, 0), axis_contiguous=ac,
                             transpose_backend=backends.get(pin_backend, cd.TRANSPOSE_COMM_NCCL),
                             halo_backend=cd.HALO_COMM_NCCL if rccl_only else None)
        if not tune_backend and pin_pdims:
            return cd.cudecompGridDescCreate(h, cfg), cfg, None
        opt = cd.cudecompGridDescAutotuneOptionsSetDefaults()
        opt.dtype = cd.DOUBLE
        opt.n_warmup_trials, opt.n_trials = 2, 3
        opt.autotune_transpose_backend = tune_backend
        for i in range(4):
            opt.transpose_use_inplace_buffers[i] = bool(args.inplace)
        if tune_backend:
            os.environ["CUDECOMP_AUTOTUNE_TRANSPOSE_BACKENDS"] = ",".join(lib_names[b] for b in sweep_backends)
        with c_stdout_to_stderr() as log:  # the sweep logs "CUDECOMP: ..." lines on stdout; keep ours a single JSON line
            gd = cd.cudecompGridDescCreate(h, cfg, opt)
        return gd, cfg, {"pdims": pin_pdims is None, "backend": tune_backend,
                         "candidates": list(sweep_backends) if tune_backend else None,
                         "sweep": parse_sweep(log.text) if rank == 0 else None}

    def measure(pin_backend, pin_pdims, sweep_backends):
        """One complete measurement (out of place and in place) with one descriptor; returns the record pieces."""
        gd, cfg, autotuned = make_desc(pin_backend, pin_pdims, sweep_backends)
        pdims = (cfg.pdims[0], cfg.pdims[1])
        used = names.get(cfg.transpose_comm_backend, cd.cudecompTransposeCommBackendToString(cfg.transpose_comm_backend))
        pen = Pencils(cd, torch, h, gd, 1234 + rank)
        res = {}
        order = (True, False) if args.inplace else (False, True)
        for inplace in order[::-1]:  # the headline variant LAST, so that the library's per-op samples are its own
            res[inplace] = run_cycles(cd, torch, dist, world, h, gd, pen, inplace, args.warmup, args.steps, stream)
        head = res[bool(args.inplace)]
        kernel = cd.cudecompExtLastKernelName()
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
        cut = 0  # bytes this job pushes across the half/half cut of the node per cycle, both directions
        if world > 1:
            for op in cd.OPS:
                p = cd.cudecompExtGetTransposePlan(h, gd, op, inplace=bool(args.inplace))
                if p.exchange:
                    for d in range(p.nranks):
                        peer = p.member_global_rank[d]
                        if (peer < world // 2) != (rank < world // 2):
                            cut += p.send_cnt[d] * es
            t = torch.tensor([cut], dtype=torch.int64)
            dist.all_reduce(t)
            cut = int(t[0])
        counters = cd.cudecompExtGetCounters(h, gd)
        return {"gd": gd, "pen": pen, "pdims": pdims, "used": used, "autotuned": autotuned, "res": res, "head": head,
                "kernel": kernel, "split": split, "cut": cut, "counters": counters,
                "ok": head["ok"] and res[not args.inplace]["ok"]}

    def release(m):
        m["pen"].free()
        with c_stdout_to_stderr():  # the library prints its performance summary here when enabled
            cd.cudecompGridDescDestroy(h, m["gd"])

    def record(m, fallback=None, preflight=None):
        head = m["head"]
        ms_per_step = head["wall_ms"] / args.steps
        global_bytes = n ** 3 * es
        value = 4 * global_bytes / (ms_per_step * 1e-3) / 1e9
        pdims = m["pdims"]
        out = {
            # BASELINE.json's metric, verbatim at its size; `value` is the effective GB/s (4 x global bytes / cycle time),
            # `ms_per_step` the cycle wall time, `xgmi` the bisection fraction (N > 1)
            "metric": ("transpose cycle wall time + effective GB/s (vs xGMI bisection), 1024³ fp64" if n == 1024 else
                       "transpose cycle wall time + effective GB/s (vs xGMI bisection), %d^3 fp64" % n),
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d^3 fp64 X->Y->Z->Y->X transpose cycle, %dx%d process grid, %s layout, %s"
                                   % (n, pdims[0], pdims[1],
                                      "all-axis-contiguous" if args.layout == "contiguous" else "default (X fastest)",
                                      "in-place" if args.inplace else "out-of-place"),
                       "pdims": list(pdims), "transport": m["used"], "autotuned": m["autotuned"],
                       "per_op_ms": [round(x, 4) for x in head["op_ms"]], "per_op_split": m["split"],
                       "per_op_ms_sustained": [round(x, 4) for x in head.get("op_ms_sustained", [])],
                       "round_trip_checksum_ok": bool(m["ok"]), "fallback": fallback,
                       "direct_puts": m["counters"]["direct_puts"]},
            # protocol of SURVEY 8(d): per-cycle device times (HIP events on the library's stream) over the timed cycles of
            # all ranks, both variants
            "stats": {"protocol": "%d warm-up + %d timed cycles, barrier + device sync on both sides; per-cycle HIP events"
                                  % (args.warmup, args.steps),
                      "out_of_place_cycle_ms": stats_of(m["res"][False]["cycle_ms_all_ranks"]),
                      "in_place_cycle_ms": stats_of(m["res"][True]["cycle_ms_all_ranks"]),
                      "out_of_place_GBps": round(4 * global_bytes / (m["res"][False]["wall_ms"] / args.steps * 1e-3) / 1e9, 2),
                      "in_place_GBps": round(4 * global_bytes / (m["res"][True]["wall_ms"] / args.steps * 1e-3) / 1e9, 2),
                      "in_place_per_op_ms": [round(x, 4) for x in m["res"][True]["op_ms"]],
                      "in_place_round_trip_checksum_ok": bool(m["res"][True]["ok"])},
            "roofline": None,
        }
        if preflight is not None:
            out["config"]["preflight"] = preflight
        return out

    # ------------------------------------------------------------------------------------------------------ N = 1
    if world == 1:
        m = measure(args.backend, (1, 1), None)
        out = record(m)
        head = m["head"]
        pen = m["pen"]
        # dominant kernel: at 1x1 out of place every hop is ONE launch that reads and writes each element of the pencil
        # once (LDS-tiled permutation for the contiguous layout, streaming row copy for the default layout); in place it
        # is two launches per hop (permute into the workspace, copy back)
        launches_per_cycle = 4 if not args.inplace else (8 if args.layout == "contiguous" else 0)
        alg_bytes = 2 * pen.pinfo[0].size * es
        if launches_per_cycle:
            avg_ms = sum(head["cycle_ms"]) / len(head["cycle_ms"]) / launches_per_cycle
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
            traffic, src = measured_traffic(args.layout) if n == 1024 and not args.inplace else (None, None)
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "traffic_source": (src + " (rocprofv3 PMC passes of this command; counters cannot run inside the "
                                       "timed loop)") if src else None,
                    "kernel": m["kernel"], "kernel_source": "cudecompExtLastKernelName() after every hop of one extra cycle",
                    "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(avg_ms, 4),
                    "launches_per_cycle": launches_per_cycle,
                    "duration_source": "HIP events on the library's stream around each timed cycle"}
            # where a cycle's time is: the kernels one at a time (each hop alone between two host synchronisations), the
            # same hops inside a sustained sequence (events between them, no host synchronisation), and what is left
            iso, sus = sum(head["op_ms"]), sum(head.get("op_ms_sustained", []))
            cyc_ms = sum(head["cycle_ms"]) / len(head["cycle_ms"])
            roof["kernel_sum_ms"] = round(iso, 4)
            roof["sustained_hop_sum_ms"] = round(sus, 4)
            roof["gap_ms"] = round(cyc_ms - iso, 4)
            roof["gap_note"] = ("kernel_sum_ms = the hops of one cycle timed ONE AT A TIME (host synchronisation before each); "
                                "sustained_hop_sum_ms = the same hops inside back-to-back cycles (events between them, no host "
                                "synchronisation); gap_ms = timed cycle - kernel_sum_ms.  rocprofv3 begin / end timestamps of the same "
                                "command (profiles/r05_cycle_gaps.json): median idle time between one kernel's end and the next "
                                "one's begin 0 us, back-to-back kernels not slower than isolated ones")
            if not args.inplace and head.get("op_kernels"):
                # the hops of a cycle may run different instantiations (round 5: forward and inverse hops use different tiles /
                # tile walks); `achieved` is the average over the launches of a cycle, the per-kernel figures come from the
                # sustained per-hop events
                per_k = {}
                for name, ms_ in zip(head["op_kernels"], head.get("op_ms_sustained", [])):
                    per_k.setdefault(name, []).append(ms_)
                roof["kernel"] = " + ".join(sorted(per_k)) if per_k else roof["kernel"]
                roof["per_kernel"] = [{"kernel": k, "launches_per_cycle": len(v), "avg_launch_ms": round(sum(v) / len(v), 4),
                                       "achieved": round(alg_bytes / (sum(v) / len(v) * 1e-3) / 1e9, 1),
                                       "frac": round(alg_bytes / (sum(v) / len(v) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)} for k, v in sorted(per_k.items())]
            if args.layout == "contiguous" and not args.inplace:
                # context for `frac`: what a plain copy of the same pencil reaches on this GPU right now (the library's
                # row-copy kernel on the same buffers: X->Y of a 1x1 grid in the default layout), outside the timed region
                gdc = cd.cudecompGridDescCreate(h, cd.make_config((n, n, n), (1, 1), axis_contiguous=(0, 0, 0)))
                for _ in range(2):
                    cd.cudecompTranspose("XToY", h, gdc, pen.a.data_ptr(), pen.b.data_ptr(), pen.work, cd.DOUBLE, stream=stream)
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record()
                for _ in range(5):
                    cd.cudecompTranspose("XToY", h, gdc, pen.a.data_ptr(), pen.b.data_ptr(), pen.work, cd.DOUBLE, stream=stream)
                c1.record()
                torch.cuda.synchronize()
                copy_rate = alg_bytes / (c0.elapsed_time(c1) / 5 * 1e-3) / 1e9
                roof["copy_rate"] = {"achieved": round(copy_rate, 1), "unit": "GB/s", "kernel": cd.cudecompExtLastKernelName(),
                                     "what": "dense copy of the same 8 GiB pencil, same buffers, measured in this run"}
                roof["frac_of_copy_rate"] = round(achieved / copy_rate, 4)
                cd.cudecompGridDescDestroy(h, gdc)
            out["roofline"] = roof
        release(m)
        if not args.no_extras and n == 1024:
            out["extra"] = extras_single_gpu(cd, torch, h, stream)
            out["extra"]["dtypes"] = dtype_table(cd, torch, h, stream)
        if args.cpu_sample != 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, n, args.layout)
        emit(out)
        cd.cudecompFinalize(h)
        return

    # ------------------------------------------------------------------------------------------------------ N > 1
    def xgmi_block(out, m):
        # one direction of one link: measured at start-up when the ranks sit on different GPUs, nominal otherwise;
        # (N/2)^2 links cross the half/half cut, each carrying both directions
        link = cd.cudecompExtGetLinkInfo(h)
        per_dir, src = LINK_GBPS_PER_DIRECTION, "nominal (153.6 GB/s per link counting both directions)"
        if link["measured"] and link["crosses_devices"]:
            per_dir = max(link["gbps_sdma"], link["gbps_cu"])
            src = "measured at start-up (64 MiB one-direction copy to the next rank, slowest rank)"
        bis = (world // 2) ** 2 * 2 * per_dir
        ms_per_step = out["ms_per_step"]
        out["xgmi"] = {"cut_bytes_per_cycle": m["cut"], "cut_GBps": round(m["cut"] / (ms_per_step * 1e-3) / 1e9, 1),
                       "link_GBps_per_direction": round(per_dir, 1), "link_rate_source": src,
                       "bisection_GBps_both_directions": round(bis, 1),
                       "frac_of_bisection": round(m["cut"] / (ms_per_step * 1e-3) / 1e9 / bis, 4),
                       "link_probe": link}
        return out

    best, best_m, preflight = None, None, None
    everything = []  # every (phase, grid, transport) this run measured, winner or not: the N > 1 line describes itself

    def note(phase, m, cand):
        if m.get("autotuned") and m["autotuned"].get("sweep"):
            for c in m["autotuned"]["sweep"]:
                everything.append(dict(c, phase=phase + " (library autotuner, %d+%d cycles)" % (2, 3)))
        everything.append({"phase": phase + " (bench protocol)", "pdims": list(m["pdims"]), "transport": m["used"],
                           "avg_ms": cand["ms_per_step"], "status": "measured" if m["ok"] else "round trip FAILED",
                           "per_op_ms": cand["config"]["per_op_ms"], "direct_puts": m["counters"]["direct_puts"]})
    # ---- phase A: RCCL, process grid autotuned (or pinned) -- the transport the north star names; also the safety net
    if args.backend in ("auto", "nccl", "nccl_pl") or os.environ.get("BENCH_FORCE_RCCL_PHASE"):
        try:
            pin = args.backend if args.backend in ("nccl", "nccl_pl") else "nccl"
            m = measure(pin, args.pdims, None)
            if all_ok(m["ok"]):
                best = xgmi_block(record(m), m)
                best_m = m
                note("A: RCCL", m, best)
                if rank == 0:
                    fallback_line[0] = json.loads(json.dumps(best))
                    sys.stderr.write("bench.py: RCCL phase: %dx%d grid, %.3f ms per cycle\n" % (m["pdims"][0], m["pdims"][1], best["ms_per_step"]))
            else:
                if rank == 0:
                    sys.stderr.write("bench.py: RCCL phase: round-trip checksum FAILED\n")
                release(m)
        except Exception as e:  # RCCL unusable here: the one-sided transports may still work
            sys.stderr.write("bench.py: rank %d: RCCL phase failed: %s\n" % (rank, str(e)[:300]))
            all_ok(False)

    # ---- phase B: preflight of the one-sided transports on a small grid, then phase C: the library's autotuner over
    #      everything that passed
    if args.backend not in ("nccl", "nccl_pl"):
        preflight = {}
        cands = ["peer", "peer_pl", "peer_sm", "mpi"] if args.backend == "auto" else [args.backend]
        if os.environ.get("BENCH_PREFLIGHT"):  # debugging aid: the transports to try first, verbatim
            cands = os.environ["BENCH_PREFLIGHT"].split(",")
        small = 64
        for name in cands:
            ok, why = True, ""
            try:
                cfg = cd.make_config((small,) * 3, (1, world), axis_contiguous=ac, transpose_backend=backends[name])
                gd = cd.cudecompGridDescCreate(h, cfg)
                pen = Pencils(cd, torch, h, gd, 99 + rank)
                if name == cands[0]:
                    bad = cd.cudecompExtPeerProbe(h, pen.work, cd.cudecompGetTransposeWorkspaceSize(h, gd) * 8) \
                        if cd.cudecompGetTransposeWorkspaceSize(h, gd) * 8 >= 4 * 4096 else 0
                    if bad:
                        ok, why = False, "peer probe: %d wrong blocks" % bad
                r = run_cycles(cd, torch, dist, world, h, gd, pen, False, 1, 2, stream)
                if not r["ok"]:
                    ok, why = False, "round trip not exact"
                pen.free()
                cd.cudecompGridDescDestroy(h, gd)
            except Exception as e:
                ok, why = False, str(e)[:200]
            ok = all_ok(ok)
            preflight[name] = "pass" if ok else ("FAIL: " + why if why else "FAIL on another rank")
        if rank == 0:
            link = cd.cudecompExtGetLinkInfo(h)
            sys.stderr.write("bench.py: preflight of the one-sided transports (%d^3, 1x%d): %s; link probe: %s\n"
                             % (small, world, json.dumps(preflight), json.dumps(link)))
        passed = [c for c in cands if preflight[c] == "pass"]
        if passed:
            try:
                if args.backend == "auto":
                    sweep = passed + ["mpi_pl"] + (["nccl"] if best is not None else [])
                    if os.environ.get("BENCH_SWEEP"):  # debugging aid: the candidate list, verbatim
                        sweep = os.environ["BENCH_SWEEP"].split(",")
                    m = measure("auto", args.pdims, sweep)
                else:
                    m = measure(args.backend, args.pdims, None)
                if all_ok(m["ok"]):
                    cand = xgmi_block(record(m, preflight=preflight), m)
                    note("C: one-sided", m, cand)
                    if best is None or cand["ms_per_step"] < best["ms_per_step"]:
                        if best_m is not None:
                            release(best_m)
                        best, best_m = cand, m
                    else:
                        best["config"]["preflight"] = preflight
                        release(m)
                else:
                    if rank == 0:
                        sys.stderr.write("bench.py: round-trip checksum FAILED with transport %s on %dx%d; keeping the RCCL result\n"
                                         % (m["used"], m["pdims"][0], m["pdims"][1]))
                    if best is not None:
                        best["config"]["fallback"] = "round-trip checksum failed with transport %s" % m["used"]
                    release(m)
            except Exception as e:
                sys.stderr.write("bench.py: rank %d: one-sided phase failed: %s\n" % (rank, str(e)[:300]))
                all_ok(False)
                if best is not None:
                    best["config"]["fallback"] = "one-sided phase failed: %s" % str(e)[:120]
        elif best is not None:
            best["config"]["fallback"] = "no one-sided transport passed the preflight"
            best["config"]["preflight"] = preflight

    if rank == 0:
        if best is None:
            raise SystemExit("bench.py: no transport produced a valid result")
        best["config"]["also_measured"] = everything
        emit(best)
    if best_m is not None:
        release(best_m)
    cd.cudecompFinalize(h)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
