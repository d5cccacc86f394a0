#!/usr/bin/env python3
"""Where the time between the kernels of a sustained transpose cycle goes (round 5).

    python scripts/summarize_gaps.py <rocprofv3 kernel_trace.csv> > profiles/r05_cycle_gaps.json

From the begin / end timestamps of every dispatch of the library's data-movement kernels in a `bench.py` run: durations of
the kernels that run back to back (the previous kernel ended less than 1 ms before) against those that start on an idle
device, and the idle time between the end of one kernel and the begin of the next."""
import csv
import json
import statistics
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name") or r.get("Name") or ""
        if "cudecomp" not in name or "transpose_kernel" not in name:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
back, alone, gaps = [], [], []
for i, (s, e, n) in enumerate(rows):
    if i and s - rows[i - 1][1] < 1_000_000:
        back.append((e - s) / 1e6)
        gaps.append((s - rows[i - 1][1]) / 1e3)
    else:
        alone.append((e - s) / 1e6)


def st(x):
    return None if not x else {"n": len(x), "mean": round(statistics.mean(x), 4), "median": round(statistics.median(x), 4),
                               "min": round(min(x), 4), "max": round(max(x), 4)}


# per direction inside the sustained sequences: a cycle is fwd fwd bwd bwd; the sequence position is not in the trace, so split
# the back-to-back durations at their median gap of 0.15 ms between the two populations
out = {"source": "rocprofv3 --kernel-trace of python bench.py --steps 5 --warmup 3 --cpu-sample 0 --no-extras (1024^3 fp64, "
                 "axis-contiguous layout, out of place)",
       "dispatches": len(rows), "kernel": rows[0][2][:120] if rows else None,
       "kernel_ms_started_on_an_idle_device": st(alone), "kernel_ms_back_to_back": st(back),
       "idle_us_between_end_and_next_begin": st(gaps)}
if back and alone:
    out["back_to_back_minus_idle_start_ms"] = round(statistics.mean(back) - statistics.mean(alone), 4)
print(json.dumps(out, indent=1))
