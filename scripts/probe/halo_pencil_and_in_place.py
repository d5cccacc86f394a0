"""The two round-6 kernels under a profiler: forward / inverse hops onto halo-carrying pencils (transpose_lines_kernel /
transpose_window_kernel) at 1024^3 fp64 halo 1, then in-place cycles (rotate_kernel); few launches each.
    rocprofv3 --kernel-trace --stats -- python scripts/probe/halo_pencil_and_in_place.py"""
import torch

import cudecomp_amd as cd


def main():
    torch.cuda.set_device(0)
    h = cd.cudecompInit()
    st = torch.cuda.current_stream().cuda_stream
    n, halo = 1024, (1, 1, 1)
    gd = cd.cudecompGridDescCreate(h, cd.make_config((n, n, n), (1, 1), axis_contiguous=(1, 1, 1)))
    nel = max(cd.cudecompGetPencilInfo(h, gd, ax, halo).size for ax in range(3))
    a = torch.zeros(nel, dtype=torch.float64, device="cuda")
    b = torch.zeros(nel, dtype=torch.float64, device="cuda")
    work = cd.cudecompMalloc(h, gd, cd.cudecompGetTransposeWorkspaceSize(h, gd) * 8)
    for _ in range(3):
        for op in cd.OPS:
            cd.cudecompTranspose(op, h, gd, a.data_ptr(), b.data_ptr(), work, cd.DOUBLE, halo, halo, None, None, st)
    torch.cuda.synchronize()
    for _ in range(3):   # in place, no halos: every hop is one rotate_kernel
        for op in cd.OPS:
            cd.cudecompTranspose(op, h, gd, a.data_ptr(), a.data_ptr(), work, cd.DOUBLE, stream=st)
    torch.cuda.synchronize()
    cd.cudecompFree(h, gd, work)
    cd.cudecompGridDescDestroy(h, gd)
    cd.cudecompFinalize(h)


if __name__ == "__main__":
    main()
