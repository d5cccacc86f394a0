#!/bin/bash
# L2 requests PER CHANNEL (TCC_REQ un-summed: one value per XCD x channel) of the rotation kernel under two orbit walks:
# b0 fastest (CUDECOMP_ROTATE_WALK=0) and the default per-XCD walk.  Tuning build.  Summary: gpurun_out/r06_tcc/summary.json
REPO=$PWD; O=$REPO/gpurun_out/r06_tcc; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$REPO CUDECOMP_AMD_LIBRARY=$REPO/cudecomp_amd/lib_tuning/libcudecomp.so
export ROTATE_AB_ROUNDS=1 ROTATE_AB_CHECK=0 ROTATE_AB_CASES=${ROTATE_L2_CASE:-fp64:1024}
cd /tmp && export TMPDIR=/tmp
for arm in 0 8991; do
  for ctr in TCC_REQ TCC_EA0_WRREQ TCC_EA0_RDREQ; do
    ROTATE_AB_ARMS=$arm timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/walk_${arm}_$ctr -o rot -- \
      python $REPO/scripts/probe/rotate_walk_ab.py > $O/walk_${arm}_$ctr.log 2>&1
  done
done
cd $REPO
python - <<'PY'
import csv, glob, json, collections, os
out = {}
for d in sorted(glob.glob("gpurun_out/r06_tcc/walk_*")):
    if not os.path.isdir(d):
        continue
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        out[os.path.basename(d)] = "no counter file"
        continue
    rows = list(csv.DictReader(open(files[0])))
    cols = list(rows[0].keys()) if rows else []
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # dispatch -> instance -> value
    for r in rows:
        if "rotate_kernel" not in r.get("Kernel_Name", ""):
            continue
        inst = tuple((k, r[k]) for k in cols if k.upper().startswith("DIMENSION") or k in ("Agent_Id",))
        per[r.get("Dispatch_Id")][inst] += float(r.get("Counter_Value", 0))
    stats = []
    for disp, inst in per.items():
        v = list(inst.values())
        if len(v) > 1 and sum(v) > 0:
            stats.append({"instances": len(v), "max_over_mean": round(max(v) / (sum(v) / len(v)), 3), "min_over_mean": round(min(v) / (sum(v) / len(v)), 3), "total": sum(v)})
    out[os.path.basename(d)] = {"columns": cols, "rows": len(rows), "dispatches": len(per), "first": stats[:3], "last": stats[-2:]}
    for f in files:
        if os.path.getsize(f) > (2 << 20):
            os.remove(f)
for f in glob.glob("gpurun_out/r06_tcc/**/*kernel_trace.csv", recursive=True):
    os.remove(f)
json.dump(out, open("gpurun_out/r06_tcc/summary.json", "w"), indent=1)
print(json.dumps(out)[:3000])
PY
