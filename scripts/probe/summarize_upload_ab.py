"""Condense gpurun_out/r05_upload_ab/results.jsonl (scripts/probe/upload_ab.py) into the tracked profiles/r05_upload_ab.jsonl: one
line per (arm, program, slice) without the bulky fields, DIAG lines kept (truncated), and a totals line per arm."""
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05_upload_ab/results.jsonl"
tot = {}
for line in open(src):
    try:
        r = json.loads(line)
    except ValueError:
        continue
    if "arm" not in r:
        if "totals" not in r:
            print(json.dumps(r))
        continue
    arm = r["arm"].split("_")[0]
    keep = {k: r[k] for k in ("arm", "program", "ranks", "env", "completed", "passed", "failed", "gate_trips_input_stale",
                              "download_mismatches", "interior_overwritten", "killed_at_deadline", "wall_s", "ms_per_case",
                              "kfd_queues_max", "failing")}
    keep["env"] = {k: v for k, v in keep["env"].items() if k != "LD_LIBRARY_PATH"}
    keep["diag"] = [d[:400] for d in r["diag"][:12]]
    print(json.dumps(keep))
    t = tot.setdefault(arm, {"completed": 0, "failed": 0, "gate_trips_input_stale": 0, "download_mismatches": 0,
                             "interior_overwritten": 0, "wall_s": 0.0})
    for k in t:
        t[k] += r[k]
print(json.dumps({"totals_per_arm": tot}))
