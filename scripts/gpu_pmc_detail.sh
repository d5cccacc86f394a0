#!/bin/bash
# extra PMC passes on the bench's dominant kernels (one counter group per pass; kernel-trace only)
mkdir -p gpurun_out/pmc
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVES SQ_BUSY_CYCLES" "TCC_EA_WRREQ_STALL_sum TCC_EA_RDREQ_sum" "SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  for layout in contiguous default; do
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $REPO/gpurun_out/pmc/${layout}_$i -o bench -- \
      python $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 --layout $layout > $REPO/gpurun_out/pmc/${layout}_$i.log 2>&1
  done
done
cd $REPO
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc/*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "cudecomp" not in k:
            continue
        k = k.split("::")[-1][:40]
        agg[(k, row.get("Counter_Name"))][0] += 1
        agg[(k, row.get("Counter_Value") and row.get("Counter_Name"))][1] += float(row.get("Counter_Value", 0))
    for (k, c), (n, s) in sorted(agg.items()):
        print("%-28s %-42s %-28s n=%d mean=%.4g" % (f.split("/")[2], k, c, n, s / max(n, 1)))
PY
find gpurun_out/pmc -name "*.csv" -size +1M -delete
