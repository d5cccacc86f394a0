"""Out-of-place and in-place cycles on cubic 1 x 1 grids of several edges (fp64, axis-contiguous): is a power-of-two edge slower?
    python scripts/probe/cube_sizes.py"""
import json
import os

import torch

import cudecomp_amd as cd


def main():
    torch.cuda.set_device(0)
    h = cd.cudecompInit()
    st = torch.cuda.current_stream().cuda_stream
    edges = [int(x) for x in os.environ.get("CUBE_EDGES", "1024,1280,1152,960,1008,1040").split(",")]
    for ac in ((1, 1, 1), (0, 0, 0)):
        for n in edges:
            gd = cd.cudecompGridDescCreate(h, cd.make_config((n, n, n), (1, 1), axis_contiguous=ac))
            a = torch.zeros(n**3, dtype=torch.float64, device="cuda")
            b = torch.zeros(n**3, dtype=torch.float64, device="cuda")
            work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * 8)
            out = {"edge": n, "axis_contiguous": ac[0]}
            for label, dst in (("out_of_place", b), ("in_place", a)):
                hops = {}
                for op in cd.OPS:
                    for _ in range(2):
                        cd.cudecompTranspose(op, h, gd, a.data_ptr(), dst.data_ptr(), work, cd.DOUBLE, stream=st)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(4):
                        cd.cudecompTranspose(op, h, gd, a.data_ptr(), dst.data_ptr(), work, cd.DOUBLE, stream=st)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / 4
                    hops[op] = [round(ms, 3), round(16 * n**3 / ms / 1e6 / 8000.0, 3), cd.cudecompExtLastKernelName().split("<")[0]]
                out[label] = hops
            print(json.dumps(out), flush=True)
            cd.cudecompFree(h, gd, work)
            cd.cudecompGridDescDestroy(h, gd)
            del a, b
    cd.cudecompFinalize(h)


if __name__ == "__main__":
    main()
