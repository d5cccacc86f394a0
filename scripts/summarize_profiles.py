#!/usr/bin/env python3
"""Condense gpurun_out/prof (written by scripts/gpu_profile.sh on the GPU box) into the tracked summaries under
profiles/: per-kernel stats of the bench command for both layouts and the HBM traffic per launch from the
FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE doubled on gfx950, see MI355X_MICROARCH.md)."""
import csv
import json
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = {"round": int(rnd[1:]),
       "command": "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python "
                  "bench.py --steps 5 --warmup 3 --cpu-sample 0 --no-extras --layout <layout>",
       "note": "FETCH_SIZE is doubled: on gfx950 it counts 128-B requests of wide streaming reads as 64 B "
               "(MI355X_MICROARCH.md, HBM section)",
       "workload": "1024^3 fp64 X->Y->Z->Y->X, 1x1 grid, out-of-place", "layouts": {}}
def short(name):  # "void cudecomp::kern::transpose_kernel<8, 2, 64, 64, 2, true>(cudecomp::kern::Batch)" -> the template spelling
    name = name.split("(cudecomp")[0]
    return name.replace("void ", "").replace("cudecomp::kern::", "").replace("(anonymous namespace)::", "")


for layout in ("contiguous", "default"):
    rows = list(csv.DictReader(open("gpurun_out/prof/%s_trace/bench_kernel_stats.csv" % layout)))
    with open("profiles/%s_bench_%s_kernel_stats.csv" % (rnd, layout), "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
        for r in rows:
            n = r["Name"] if len(r["Name"]) <= 160 else r["Name"][:157] + "..."
            f.write('"%s",%s,%s,%s,%s,%s,%s,%s\n' % (n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                                                     r["MinNs"], r["MaxNs"], r["StdDev"]))
    # the dominant kernels of the timed cycles: the LDS-tiled transposes for the axis-contiguous layout (two instantiations since
    # round 5: forward and inverse hops use different tiles), the row copy for the default layout
    ks = [r for r in rows if "transpose_kernel" in r["Name"]] if layout == "contiguous" else []
    if not ks:
        ks = [[r for r in rows if "cudecomp" in r["Name"]][0]]
    calls = sum(int(r["Calls"]) for r in ks)
    avg_ns = sum(float(r["TotalDurationNs"]) for r in ks) / calls

    def mean(kind, counter):
        tot, n = 0.0, 0
        for r in csv.DictReader(open("gpurun_out/prof/%s_%s/bench_counter_summary.csv" % (layout, kind))):
            if r["counter"] == counter and any(r["kernel"] == k["Name"] for k in ks):
                tot += float(r["sum"])
                n += int(r["dispatches"])
        return tot / n if n else None

    fk, wk = mean("fetch", "FETCH_SIZE"), mean("write", "WRITE_SIZE")
    rd, wr = fk * 1024 * 2, wk * 1024
    out["layouts"][layout] = {"kernel": " + ".join(short(k["Name"]) for k in ks), "calls": calls, "avg_ns": avg_ns,
                              "per_kernel": [{"kernel": short(k["Name"]), "calls": int(k["Calls"]),
                                              "avg_ns": float(k["AverageNs"]), "min_ns": float(k["MinNs"]), "max_ns": float(k["MaxNs"])} for k in ks],
                              "FETCH_SIZE_KB_mean_per_dispatch": fk, "WRITE_SIZE_KB_mean_per_dispatch": wk,
                              "hbm_read_bytes_per_launch_corrected": rd, "hbm_write_bytes_per_launch": wr,
                              "hbm_traffic_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": 2 * 1024 ** 3 * 8}
# the other element types at the benchmark's pencil size (scripts/gpu_profile_dtypes.sh), when that pass was run
import os
out["dtypes"] = {}
for dt, es in (("fp32", 4), ("complex128", 16)):
    stats = "gpurun_out/prof/dtype_%s_trace/bench_kernel_stats.csv" % dt
    if not os.path.exists(stats):
        continue
    rows = list(csv.DictReader(open(stats)))
    ks = [r for r in rows if "transpose_kernel" in r["Name"]]
    if not ks:
        continue
    k = ks[0]

    def dmean(kind, counter):
        try:
            for r in csv.DictReader(open("gpurun_out/prof/dtype_%s_%s/bench_counter_summary.csv" % (dt, kind))):
                if r["kernel"] == k["Name"] and r["counter"] == counter:
                    return float(r["mean_per_dispatch"])
        except OSError:
            pass
        return None

    fk, wk = dmean("fetch", "FETCH_SIZE"), dmean("write", "WRITE_SIZE")
    alg = 2 * 8 * 1024 ** 3
    rec = {"kernel": k["Name"], "calls": int(k["Calls"]), "avg_ns": float(k["AverageNs"]),
           "achieved_GBps": round(alg / float(k["AverageNs"]), 1), "frac_of_8TBps": round(alg / float(k["AverageNs"]) / 8000.0, 4),
           "algorithmic_bytes_per_launch": alg, "workload": "8-GiB pencil, 1x1 grid, axis-contiguous layout, out of place "
           "(scripts/probe/dtype_table.py %s)" % dt}
    if fk is not None and wk is not None:
        rec.update({"FETCH_SIZE_KB_mean_per_dispatch": fk, "WRITE_SIZE_KB_mean_per_dispatch": wk,
                    "hbm_traffic_bytes_per_launch": fk * 1024 * 2 + wk * 1024,
                    "traffic_over_algorithmic": round((fk * 1024 * 2 + wk * 1024) / alg, 4)})
    out["dtypes"][dt] = rec
json.dump(out, open("profiles/%s_pmc_summary.json" % rnd, "w"), indent=1)
print(json.dumps({k: (v["kernel"].split("::")[-1], round(v["avg_ns"] / 1e6, 4), v["hbm_traffic_bytes_per_launch"])
                  for k, v in out["layouts"].items()}, indent=1))
