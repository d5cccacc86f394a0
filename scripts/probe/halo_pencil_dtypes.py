"""Transposes onto halo-carrying pencils (axis-contiguous, 1 x 1 grid, halo 1) for the other element sizes at 8-GiB pencils: the
whole-line kernels (default) against the window kernel (CUDECOMP_PRESERVE_OUTPUT_HALOS=1), one process per arm.
    python scripts/probe/halo_pencil_dtypes.py ;  CUDECOMP_PRESERVE_OUTPUT_HALOS=1 python scripts/probe/halo_pencil_dtypes.py"""
import json
import os

import torch

import cudecomp_amd as cd


def main():
    torch.cuda.set_device(0)
    h = cd.cudecompInit()
    st = torch.cuda.current_stream().cuda_stream
    out = {"preserve": os.environ.get("CUDECOMP_PRESERVE_OUTPUT_HALOS", "0"), "cases": {}}
    halo = (1, 1, 1)
    for name, dt, es, gdims in (("fp32", cd.FLOAT, 4, (2048, 1024, 1024)), ("complex128", cd.DOUBLE_COMPLEX, 16, (1024, 1024, 512)),
                                ("fp64 odd extents", cd.DOUBLE, 8, (1022, 1026, 1000))):
        gd = cd.cudecompGridDescCreate(h, cd.make_config(gdims, (1, 1), axis_contiguous=(1, 1, 1)))
        nel = max(cd.cudecompGetPencilInfo(h, gd, ax, halo).size for ax in range(3))
        a = torch.zeros(nel * es, dtype=torch.uint8, device="cuda")
        b = torch.zeros(nel * es, dtype=torch.uint8, device="cuda")
        work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * es)
        ops, moved = {}, 2 * es * gdims[0] * gdims[1] * gdims[2]
        for op in cd.OPS:
            for _ in range(2):
                cd.cudecompTranspose(op, h, gd, a.data_ptr(), b.data_ptr(), work, dt, halo, halo, None, None, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                cd.cudecompTranspose(op, h, gd, a.data_ptr(), b.data_ptr(), work, dt, halo, halo, None, None, st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            ops[op] = {"ms": round(ms, 3), "frac": round(moved / ms / 1e6 / 8000.0, 3), "kernel": cd.cudecompExtLastKernelName()}
        out["cases"][name] = ops
        cd.cudecompFree(h, gd, work)
        cd.cudecompGridDescDestroy(h, gd)
        del a, b
    print(json.dumps(out))


if __name__ == "__main__":
    main()
