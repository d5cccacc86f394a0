"""Row copies onto a halo-carrying pencil: rows_shifted_kernel against rows_dense_kernel (csrc/kernels_rows.hip), through the
library's own dispatch (cudecompExtMove3D with / without the planner's "whole rows" word) and through the public API (a 1 x 1
grid in the default layout with halos on the output pencils: every transpose is ONE row copy).  Prints one JSON line.
    python scripts/probe/dense_rows_bench.py [reps]"""
import json
import sys

import torch

import cudecomp_amd as cd


def time_move(es, extent, ss, ds, dst_off, flags, reps, src, dst):
    st = torch.cuda.current_stream().cuda_stream
    args = (src.data_ptr(), dst.data_ptr() + dst_off * es, es, extent, ss, ds, flags, st)
    cd.cudecompExtMove3D(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        cd.cudecompExtMove3D(*args)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, cd.cudecompExtLastKernelName()


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    torch.cuda.set_device(0)
    out = {"what": "8-GiB row copies onto halo-carrying pencils, ms per launch and fraction of 8 TB/s (2 x bytes / time)", "moves": []}
    # (name, element bytes, row elements, rows, planes, halo on x, halo on y)
    cases = [("fp64 1024^3, halo 1 on x and y", 8, 1024, 1024, 1024, 1, 1), ("fp64 1024^3, halo 2 on x only", 8, 1024, 1024, 1024, 2, 0),
             ("fp32 2048 x 1024 x 1024, halo 1", 4, 2048, 1024, 1024, 1, 1), ("complex128 512 x 1024 x 1024, halo 1", 16, 512, 1024, 1024, 1, 1),
             ("fp64 config-5 X pencil 2048 x 1024 x 256 (+2)", 8, 2048, 1024, 256, 2, 2), ("fp64 256-wide rows, halo 1", 8, 256, 2048, 2048, 1, 1)]
    for name, es, w, h, d, hx, hy in cases:
        nbytes = w * h * d * es
        ds = [1, w + 2 * hx, (w + 2 * hx) * (h + 2 * hy)]
        dst_off = hx + hy * ds[1]
        src = torch.empty(nbytes, dtype=torch.uint8, device="cuda").random_(0, 256)
        dst = torch.zeros((ds[2] * d + dst_off + 64) * es, dtype=torch.uint8, device="cuda")
        row = {"case": name, "GiB": round(nbytes / 2**30, 2)}
        for label, flags in (("shifted", 0), ("dense", 256)):
            ms, kern = time_move(es, (w, h, d), (1, w, w * h), ds, dst_off, flags, reps, src, dst)
            row[label] = {"ms": round(ms, 3), "frac": round(2 * nbytes / ms / 8e9, 3), "kernel": kern}
        # the aligned ceiling: the same bytes into a dense destination
        ms, kern = time_move(es, (w, h, d), (1, w, w * h), (1, w, w * h), 0, 0, reps, src, dst)
        row["aligned"] = {"ms": round(ms, 3), "frac": round(2 * nbytes / ms / 8e9, 3), "kernel": kern}
        out["moves"].append(row)
        del src, dst
        torch.cuda.empty_cache()
    # public API: 1 x 1 grid, default layout, 1024^3 fp64, halo (1,1,1) on every pencil: each transpose = one row copy of 8 GiB
    h = cd.cudecompInit()
    gd = cd.cudecompGridDescCreate(h, cd.make_config((1024, 1024, 1024), (1, 1)))
    halo = (1, 1, 1)
    pin = [cd.cudecompGetPencilInfo(h, gd, ax, halo) for ax in range(3)]
    nel = max(p.size for p in pin)
    a = torch.zeros(nel * 8, dtype=torch.uint8, device="cuda")
    b = torch.zeros(nel * 8, dtype=torch.uint8, device="cuda")
    wsz = cd.cudecompGetTransposeWorkspaceSize(h, gd)
    work = cd.cudecompMalloc(h, gd, wsz * 8)
    st = torch.cuda.current_stream().cuda_stream
    api = {}
    for op in cd.OPS:
        cd.cudecompTranspose(op, h, gd, a.data_ptr(), b.data_ptr(), work, cd.DOUBLE, halo, halo, None, None, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            cd.cudecompTranspose(op, h, gd, a.data_ptr(), b.data_ptr(), work, cd.DOUBLE, halo, halo, None, None, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        api[op] = {"ms": round(ms, 3), "frac": round(2 * 8 * 2**30 / ms / 8e9, 3), "kernel": cd.cudecompExtLastKernelName()}
    out["api_1x1_default_layout_halo_1"] = api
    cd.cudecompFree(h, gd, work)
    cd.cudecompGridDescDestroy(h, gd)
    cd.cudecompFinalize(h)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
