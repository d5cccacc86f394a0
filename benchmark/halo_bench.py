#!/usr/bin/env python3
"""Halo-update timing on one rank: the per-rank X pencil of BASELINE config 5 (2048 x 2048 x 1024 fp64 on a
2x4 grid -> 2048 x 1024 x 256 per rank, halo width 2) treated as a periodic single-rank grid, so every
dim exercises the face-copy kernels (periodic self copy; reference include/internal/halo.h:165-193).
Prints per-dim time and achieved GB/s against the algorithmic bytes 2 * faces * face_bytes."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import cudecomp_amd as cd
    torch.cuda.set_device(0)
    gdims, halo = (2048, 1024, 256), (2, 2, 2)
    h = cd.cudecompInit()
    res = {}
    for name, ac in (("default", (0, 0, 0)),):
        gd = cd.cudecompGridDescCreate(h, cd.make_config(gdims, (1, 1), axis_contiguous=ac))
        p = cd.cudecompGetPencilInfo(h, gd, 0, halo)
        data = torch.zeros(p.size, dtype=torch.float64, device="cuda")
        ws = max(cd.cudecompGetHaloWorkspaceSize(h, gd, 0, halo), 1)
        work = cd.cudecompMalloc(h, gd, ws * 8)
        st = torch.cuda.current_stream().cuda_stream
        shape = list(p.shape)
        for dim in range(3):
            face = halo[dim] * (shape[(dim + 1) % 3]) * (shape[(dim + 2) % 3])
            for _ in range(3):
                cd.cudecompUpdateHalos(0, h, gd, data.data_ptr(), work, cd.DOUBLE, halo, (1, 1, 1), dim, stream=st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reps = 20
            for _ in range(reps):
                cd.cudecompUpdateHalos(0, h, gd, data.data_ptr(), work, cd.DOUBLE, halo, (1, 1, 1), dim, stream=st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            alg = 2 * 2 * face * 8  # two faces, read + write
            res["dim%d" % dim] = {"ms": round(ms, 4), "face_MiB": round(face * 8 / 2**20, 2),
                                  "GBps": round(alg / ms / 1e6, 1)}
        cd.cudecompFree(h, gd, work)
        cd.cudecompGridDescDestroy(h, gd)
    print(json.dumps({"workload": "X pencil 2048x1024x256 fp64 + halo 2, periodic self copy per dim", "result": res}))
    cd.cudecompFinalize(h)


if __name__ == "__main__":
    main()
