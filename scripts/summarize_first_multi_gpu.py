#!/usr/bin/env python3
"""Condense what scripts/first_multi_gpu.sh left under gpurun_out/first_multi_gpu/ into one JSON object: link matrix,
multi-device test verdict, the bench lines at 2 / 4 / 8 ranks next to the MODEL of DESIGN.md section 7 (slab grid, 76.8
GB/s per link and direction, local passes at 6 TB/s -- a prediction made before any multi-GPU run), every candidate the
runs measured, the flags A/B and whether the timelines were captured.

    python scripts/summarize_first_multi_gpu.py gpurun_out/first_multi_gpu > profiles/r04_scale_8gpus.json"""
import json
import os
import re
import sys

root = sys.argv[1]
MODEL_MS = {1: 11.2, 2: 60.0, 4: 16.5, 8: 5.0}  # DESIGN.md section 7, cycle time of the fixed 1024^3 fp64 problem


def load(name):
    try:
        with open(os.path.join(root, name)) as f:
            text = f.read().strip()
        return json.loads(text) if text else None
    except (OSError, ValueError):
        return None


def text(name, tail=400):
    try:
        with open(os.path.join(root, name), errors="replace") as f:
            return f.read()[-tail:]
    except OSError:
        return None


plan = text("00_plan.txt", 4000) or ""
out = {"gpus": int(re.search(r"(\d+) GPU\(s\) listed", plan).group(1)) if re.search(r"(\d+) GPU\(s\) listed", plan) else None,
       "shared": "shared=1" in plan,
       "note": ("ranks SHARED one GPU: a flow check of the script, not a measurement" if "shared=1" in plan else
                "one rank per GPU"),
       "link_matrix": load("02_link_matrix.json"),
       "multi_device_tests": [l.strip() for l in (text("03_multi_device_tests.log", 3000) or "").splitlines()
                              if re.search(r"\b(passed|failed|skipped|error)\b", l)][-3:],
       "bench": {}, "flags_ab": {}, "timeline": None}
for n in (2, 4, 8):
    rec = load("04_bench_n%d.json" % n)
    if not rec:
        continue
    size = int(re.search(r"(\d+)\^3", rec["config"]["workload"]).group(1)) if re.search(r"(\d+)\^3", rec["config"]["workload"]) else None
    out["bench"][str(n)] = {
        "ms_per_step": rec["ms_per_step"], "GBps": rec["value"], "transport": rec["config"]["transport"],
        "pdims": rec["config"]["pdims"], "size": size,
        "model_ms_1024cube": MODEL_MS.get(n), "measured_over_model": (round(rec["ms_per_step"] / MODEL_MS[n], 2)
                                                                      if size == 1024 and n in MODEL_MS else None),
        "round_trip_ok": rec["config"].get("round_trip_checksum_ok"), "fallback": rec["config"].get("fallback"),
        "preflight": rec["config"].get("preflight"), "per_op_ms": rec["config"].get("per_op_ms"),
        "per_op_split": rec["config"].get("per_op_split"), "xgmi": rec.get("xgmi"),
        "candidates": [{k: c.get(k) for k in ("phase", "pdims", "transport", "avg_ms", "model_ms", "status")}
                       for c in rec["config"].get("also_measured", [])]}
out["flags_ab"]["tiny_transposes_us"] = load("05_flag_latency.json")
for flags in (0, 1):
    rec = load("05_bench_peer_pl_flags%d.json" % flags)
    if rec:
        out["flags_ab"]["nvshmem_pl_cycle_ms_flags_in_%s" % ("device_memory" if flags else "host_board")] = {
            "ms_per_step": rec["ms_per_step"], "pdims": rec["config"]["pdims"], "round_trip_ok": rec["config"].get("round_trip_checksum_ok")}
out["two_hop_relay"] = {}
for relay in (0, 1):
    rec = load("05b_bench_2xN_relay%d.json" % relay)
    if rec:
        out["two_hop_relay"]["on" if relay else "off"] = {"ms_per_step": rec["ms_per_step"], "pdims": rec["config"]["pdims"],
                                                          "per_op_ms": rec["config"].get("per_op_ms"),
                                                          "round_trip_ok": rec["config"].get("round_trip_checksum_ok")}
tl = load("06_timeline.json")
res = tl.get("result") if isinstance(tl, dict) else None
out["timeline"] = ({k: ("captured: see 06_timeline.json" if isinstance(v, dict) and "error" not in v else v) for k, v in res.items()}
                   if isinstance(res, dict) else text("06_timeline.err"))
print(json.dumps(out, indent=1))
