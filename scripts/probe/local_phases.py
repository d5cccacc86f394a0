"""Local (pack / unpack) kernel time of one rank at the per-rank shapes of the multi-GPU BASELINE configs, measured on
ONE GPU: cudecompExtRunLocalPhases launches the plan's pack and unpack phases exactly as the executor does, without the
exchange.  Next to each op: the bytes that rank sends and the time the busiest xGMI link needs for them at the nominal
per-direction rate (one link per peer, full mesh).  A model input, not a multi-GPU measurement.

    python scripts/probe/local_phases.py > gpurun_out/local_phases.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import cudecomp_amd as cd  # noqa: E402

LINK_GBPS = 76.8  # per direction, per link (csrc/internal.h kNominalLinkGBpsPerDirection)
ORDERS = {"contiguous": [(0, 1, 2), (1, 2, 0), (2, 0, 1)], "default": [(0, 1, 2)] * 3}
AXES = {"XToY": (0, 1), "YToZ": (1, 2), "ZToY": (2, 1), "YToX": (1, 0)}


def time_phase(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


def moved_bytes(moves, n, es):
    return sum(m.extent[0] * m.extent[1] * m.extent[2] for m in moves[:n]) * es


def run_config(name, gdims, pdims, es, layout, pipelined):
    grid = cd.make_grid_spec(gdims, pdims, ORDERS[layout])
    rank = 0
    sizes = [cd.cudecompExtPencilInfo(grid, rank, a).size for a in range(3)]
    ws = max(cd.cudecompExtWorkspaceSizes(grid, rank, a, (0, 0, 0))[0] for a in range(3))
    pen = [torch.empty(max(sizes) * es, dtype=torch.uint8, device="cuda") for _ in range(2)]
    work = torch.empty(ws * es, dtype=torch.uint8, device="cuda")
    for t in pen + [work]:
        t.zero_()
    stream = torch.cuda.current_stream().cuda_stream
    ops, tot_local, tot_link = {}, 0.0, 0.0
    for op in cd.OPS:
        plan = cd.cudecompExtPlanTranspose(grid, rank, op, pipelined=pipelined, symmetric_recv=True)
        rec = {}
        for phase, n, moves in ((1, plan.n_pack, plan.pack), (2, plan.n_unpack, plan.unpack)):
            key = "pack" if phase == 1 else "unpack"
            if n == 0:
                rec[key] = None
                continue
            ms = time_phase(lambda: cd.cudecompExtRunLocalPhases(grid, rank, op, phase, pen[0].data_ptr(), pen[1].data_ptr(),
                                                                work.data_ptr(), es, stream, pipelined=pipelined,
                                                                symmetric_recv=True))
            b = 2 * moved_bytes(moves, n, es)
            rec[key] = {"ms": round(ms, 4), "launches": n if pipelined else 1, "GBps": round(b / ms / 1e6, 1),
                        "kernel": cd.cudecompExtLastKernelName()}
        sent = [plan.send_cnt[i] * es for i in range(plan.nranks) if i != plan.comm_rank]
        link_ms = (max(sent) / (LINK_GBPS * 1e6)) if sent else 0.0
        local_ms = sum(r["ms"] for r in (rec["pack"], rec["unpack"]) if r)
        rec.update({"peers": len(sent), "sent_MiB": round(sum(sent) / 2**20, 1), "busiest_link_MiB": round(max(sent) / 2**20, 1) if sent else 0,
                    "link_ms_at_nominal": round(link_ms, 3), "local_ms": round(local_ms, 4)})
        ops[op] = rec
        tot_local += local_ms
        tot_link += link_ms
    del pen, work
    torch.cuda.empty_cache()
    return {"config": name, "gdims": list(gdims), "pdims": list(pdims), "element_bytes": es, "layout": layout,
            "launch_form": "per stage, all peers (one-sided pipelined, %s stages)" % os.environ.get("CUDECOMP_PIPELINE_STAGES", "4") if pipelined else "one batch per phase", "ops": ops,
            "cycle": {"local_ms": round(tot_local, 3), "link_ms_at_nominal": round(tot_link, 3),
                      "serial_ms": round(tot_local + tot_link, 3), "overlapped_floor_ms": round(max(tot_local, tot_link), 3)}}


def main():
    out = {"what": __doc__.strip().split("\n\n")[0], "nominal_link_GBps_per_direction": LINK_GBPS, "configs": []}
    cases = [("C3 1024^3 fp64, 8 ranks", (1024,) * 3, [(2, 4), (4, 2), (1, 8), (8, 1)], 8),
             ("C3 scaling points, 4 ranks", (1024,) * 3, [(2, 2), (1, 4), (4, 1)], 8),
             ("C3 scaling points, 2 ranks", (1024,) * 3, [(2, 1), (1, 2)], 8),
             ("C2 512^3 fp64, 2 ranks", (512,) * 3, [(2, 1), (1, 2)], 8),
             ("C1 256^3 fp32, 2 ranks", (256,) * 3, [(2, 1), (1, 2)], 4)]
    only = os.environ.get("LOCAL_PHASES_ONLY")  # e.g. "C3 1024^3 fp64, 8 ranks": that case, batched launches only
    for name, gdims, grids, es in cases:
        if only and name != only:
            continue
        for pd in grids:
            for layout in ("contiguous", "default"):
                for pipelined in ((False,) if only else (False, True)):
                    if pipelined and max(pd) < 4:
                        continue
                    out["configs"].append(run_config(name, gdims, pd, es, layout, pipelined))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
