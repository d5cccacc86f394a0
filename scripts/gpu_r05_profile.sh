#!/bin/bash
# round 5: rocprofv3 evidence for bench.py -- kernel stats, begin/end gaps of the sustained cycle, HBM counters (separate passes)
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
export HSA_ENABLE_IPC_MODE_LEGACY=0
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for layout in contiguous default; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof/${layout}_trace -o bench -- \
     python $REPO/bench.py --steps 5 --warmup 3 --cpu-sample 0 --no-extras --layout $layout > $REPO/gpurun_out/prof/${layout}_trace.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/prof/${layout}_fetch -o bench -- \
     python $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extras --layout $layout > $REPO/gpurun_out/prof/${layout}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/prof/${layout}_write -o bench -- \
     python $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extras --layout $layout > $REPO/gpurun_out/prof/${layout}_write.log 2>&1
done
cd $REPO
python scripts/summarize_gaps.py $(find gpurun_out/prof/contiguous_trace -name "*kernel_trace.csv" | head -1) > gpurun_out/prof/cycle_gaps.json; cat gpurun_out/prof/cycle_gaps.json
for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do echo "== $f"; head -6 $f | cut -c1-200; done
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/prof/*_fetch/**/*counter_collection.csv", recursive=True) + glob.glob("gpurun_out/prof/*_write/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        k = (row.get("Kernel_Name"), row.get("Counter_Name"))
        agg[k][0] += 1
        agg[k][1] += float(row.get("Counter_Value", 0))
    out = f.replace("counter_collection.csv", "counter_summary.csv")
    with open(out, "w") as o:
        o.write("kernel,counter,dispatches,sum,mean_per_dispatch\n")
        for (k, c), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            o.write('"%s",%s,%d,%.1f,%.1f\n' % (k, c, n, s, s / n))
PY
find gpurun_out/prof -name "*kernel_trace.csv" -size +2M -delete
find gpurun_out/prof -name "*counter_collection.csv" -size +2M -delete
