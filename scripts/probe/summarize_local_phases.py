"""One line per (grid, layout, launch form) of a local_phases.json: pack / unpack ms and TB/s per op."""
import json
import sys

d = json.load(open(sys.argv[1]))
for c in d["configs"]:
    if c["gdims"][0] != 1024 or sum(c["pdims"]) < 6:
        continue
    row = []
    for op, r in c["ops"].items():
        for k in ("pack", "unpack"):
            x = r[k]
            row.append("-" if x is None else "%.3f ms %.2f TB/s" % (x["ms"], x["GBps"] / 1e3))
    form = "staged" if "stage" in c["launch_form"] else "batched"
    print("%s %-10s %-7s local %.3f ms | %s" % (c["pdims"], c["layout"], form, c["cycle"]["local_ms"], " | ".join(row)))
