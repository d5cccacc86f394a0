#!/bin/bash
# round 5: memory-side request counters of the library's row copies (aligned / shifted / dense) on the probe's buffers:
# do the misaligned loads of the dense kernel reach HBM twice?
REPO=$(cd "$(dirname "$0")/.." && pwd)
O=$REPO/gpurun_out/r05_dense4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$i -o probe -- $REPO/scripts/tune/partial_probe > $O/pmc_$i.log 2>&1
done
cd $REPO
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r05_dense4/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "rows_" not in k and "dense_k" not in k and "copy_k" not in k:
            continue
        k = k.split("::")[-1][:48]
        agg[(k, row.get("Counter_Name"))][0] += 1
        agg[(k, row.get("Counter_Name"))][1] += float(row.get("Counter_Value", 0))
    for (k, c), (n, s) in sorted(agg.items()):
        print("%-50s %-26s n=%d mean=%.5g" % (k, c, n, s / max(n, 1)))
PY
find gpurun_out/r05_dense4 -name "*.csv" -size +1M -delete
