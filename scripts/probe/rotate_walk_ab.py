"""In-place rotation kernel: the orbit walk (CUDECOMP_ROTATE_WALK, tuning builds), ms per hop of in-place cycles on
cubic 1 x 1 grids; one handle per arm in one process, arms interleaved over two rounds.
    CUDECOMP_AMD_LIBRARY=cudecomp_amd/lib_tuning/libcudecomp.so python scripts/probe/rotate_walk_ab.py"""
import json
import os

import torch

import cudecomp_amd as cd

KINDS = {"fp64": (cd.DOUBLE, 8), "complex128": (cd.DOUBLE_COMPLEX, 16)}
CASES = [("%s %s^3" % tuple(c.split(":")), *KINDS[c.split(":")[0]], int(c.split(":")[1]))
         for c in os.environ.get("ROTATE_AB_CASES", "fp64:1024,complex128:768,fp64:1280").split(",")]
ROUNDS = int(os.environ.get("ROTATE_AB_ROUNDS", "2"))
CHECK = os.environ.get("ROTATE_AB_CHECK", "1") == "1"
ARMS = [int(x) for x in os.environ.get("ROTATE_AB_ARMS", "0,1,8991").split(",")]   # cl | a << 4 | b << 8 | c << 12 (kernels_rotate.hip)
REF = {}


def check(h, st, cube):
    """one forward and one inverse hop of a 256^3 fp64 / 128^3 complex128 cube: every arm must produce arm 0's bytes"""
    for dt, es, n in ((cd.DOUBLE, 8, 256), (cd.DOUBLE_COMPLEX, 16, 128), (cd.DOUBLE, 8, 208)):
        gd = cd.cudecompGridDescCreate(h, cd.make_config((n, n, n), (1, 1), axis_contiguous=(1, 1, 1)))
        work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * es)
        for op in ("XToY", "ZToY"):
            a = torch.arange(n * n * n * es // 8, dtype=torch.int64, device="cuda")
            cd.cudecompTranspose(op, h, gd, a.data_ptr(), a.data_ptr(), work, dt, stream=st)
            torch.cuda.synchronize()
            key = (es, n, op)
            if key not in REF:
                REF[key] = a.clone()
            assert torch.equal(a, REF[key]), (cube, key)
        cd.cudecompFree(h, gd, work)
        cd.cudecompGridDescDestroy(h, gd)


def arm(cube):
    os.environ["CUDECOMP_ROTATE_WALK"] = str(cube)
    h = cd.cudecompInit()
    st = torch.cuda.current_stream().cuda_stream
    if CHECK:
        check(h, st, cube)
    out = {}
    for name, dt, es, n in CASES:
        gd = cd.cudecompGridDescCreate(h, cd.make_config((n, n, n), (1, 1), axis_contiguous=(1, 1, 1)))
        a = torch.zeros(n * n * n * es, dtype=torch.uint8, device="cuda")
        work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * es)
        hops = {}
        for op in cd.OPS:
            cd.cudecompTranspose(op, h, gd, a.data_ptr(), a.data_ptr(), work, dt, stream=st)
        for op in cd.OPS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            cd.cudecompTranspose(op, h, gd, a.data_ptr(), a.data_ptr(), work, dt, stream=st)
            e1.record()
            torch.cuda.synchronize()
            hops[op] = round(e0.elapsed_time(e1), 3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            for op in cd.OPS:
                cd.cudecompTranspose(op, h, gd, a.data_ptr(), a.data_ptr(), work, dt, stream=st)
        e1.record()
        torch.cuda.synchronize()
        cyc = e0.elapsed_time(e1) / 3
        assert cd.cudecompExtGetCounters(h, gd)["rotations"] > 0
        out[name] = {"cycle_ms": round(cyc, 3), "frac": round(4 * 2 * es * n**3 / cyc / 1e6 / 8000.0, 3), "hops": hops}
        cd.cudecompFree(h, gd, work)
        cd.cudecompGridDescDestroy(h, gd)
        del a
    cd.cudecompFinalize(h)
    return out


def main():
    torch.cuda.set_device(0)
    for rnd in range(ROUNDS):
        for cube in ARMS:
            print(json.dumps({"round": rnd, "walk": cube, "cases": arm(cube)}), flush=True)


if __name__ == "__main__":
    main()
