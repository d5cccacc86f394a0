"""Permutations onto halo-carrying pencils (window kernel): the tile walk A/B in the LIBRARY -- run with the tuning build and
CUDECOMP_TILE_WALK=0 / 1 / unset.  1 x 1 grid, axis-contiguous layout, fp64, halo 1 (1024^3) and config 5's pencil shape.
    CUDECOMP_AMD_LIBRARY=cudecomp_amd/lib_tuning/libcudecomp.so CUDECOMP_TILE_WALK=0 python scripts/probe/window_walk_ab.py"""
import json
import os

import torch

import cudecomp_amd as cd


def main():
    torch.cuda.set_device(0)
    pre = bool(os.environ.get("WALK_AB_PREALLOCATE"))
    if pre:  # as in bench.py: the caching allocator has served (and keeps) large blocks before these buffers are carved out
        big = [torch.zeros(1 << 30, dtype=torch.float64, device="cuda") for _ in range(3)]
        del big
    h = cd.cudecompInit()
    st = torch.cuda.current_stream().cuda_stream
    out = {"walk": os.environ.get("CUDECOMP_TILE_WALK", "default"), "cases": {},
           "preallocate": pre, "switches": {k: v for k, v in os.environ.items() if k.startswith("CUDECOMP_LINES") or k == "CUDECOMP_PRESERVE_OUTPUT_HALOS"}}
    park = int(os.environ.get("WALK_AB_PARK_GIB", "0"))
    if park:  # as in bench.py: the library's workspace pool holds the main cycle's pencils and workspace when the extras run
        gd0 = cd.cudecompGridDescCreate(h, cd.make_config((64, 64, 64), (1, 1)))
        ptrs = [cd.cudecompMalloc(h, gd0, (park << 30) // 4), cd.cudecompMalloc(h, gd0, (park << 30) // 4), cd.cudecompMalloc(h, gd0, (park << 30) // 2)]
        for q in ptrs:
            cd.cudecompFree(h, gd0, q)
        cd.cudecompGridDescDestroy(h, gd0)
    out["park_gib"] = park
    cases = [("1024^3 halo 1", (1024, 1024, 1024), (1, 1, 1), (1, 1, 1)), ("2048x1024x256 halo 2", (2048, 1024, 256), (2, 2, 2), (1, 1, 1))]
    if os.environ.get("WALK_AB_DEFAULT_FIRST"):  # bench.py measures the default layout on the same shapes first
        cases.insert(0, ("1024^3 halo 1 default layout", (1024, 1024, 1024), (1, 1, 1), (0, 0, 0)))
    for name, gdims, halo, ac in cases:
        gd = cd.cudecompGridDescCreate(h, cd.make_config(gdims, (1, 1), axis_contiguous=ac))
        nel = max(cd.cudecompGetPencilInfo(h, gd, ax, halo).size for ax in range(3))
        a = torch.zeros(nel, dtype=torch.float64, device="cuda")
        b = torch.zeros(nel, dtype=torch.float64, device="cuda")
        work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * 8)
        ops = {}
        for op in cd.OPS:
            for _ in range(2):
                cd.cudecompTranspose(op, h, gd, a.data_ptr(), b.data_ptr(), work, cd.DOUBLE, halo, halo, None, None, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                cd.cudecompTranspose(op, h, gd, a.data_ptr(), b.data_ptr(), work, cd.DOUBLE, halo, halo, None, None, st)
            e1.record()
            torch.cuda.synchronize()
            ops[op] = round(e0.elapsed_time(e1) / 5, 3)
        ops["kernel"] = cd.cudecompExtLastKernelName()
        out["cases"][name] = ops
        cd.cudecompFree(h, gd, work)
        cd.cudecompGridDescDestroy(h, gd)
        del a, b
    print(json.dumps(out))


if __name__ == "__main__":
    main()
