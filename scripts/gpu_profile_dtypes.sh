#!/bin/bash
# rocprofv3 evidence for the other element types at the benchmark's pencil size (bench.py extra.dtypes): kernel stats, then
# HBM counters in separate passes, for fp32 and complex128 (the fp64 / complex64 kernel is the bench's own: gpu_profile.sh)
mkdir -p gpurun_out/prof
export HSA_ENABLE_IPC_MODE_LEGACY=0
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for dt in fp32 complex128; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof/dtype_${dt}_trace -o bench -- \
     python $REPO/scripts/probe/dtype_table.py $dt > $REPO/gpurun_out/prof/dtype_${dt}_trace.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/prof/dtype_${dt}_fetch -o bench -- \
     python $REPO/scripts/probe/dtype_table.py $dt > $REPO/gpurun_out/prof/dtype_${dt}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/prof/dtype_${dt}_write -o bench -- \
     python $REPO/scripts/probe/dtype_table.py $dt > $REPO/gpurun_out/prof/dtype_${dt}_write.log 2>&1
done
cd $REPO
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/prof/dtype_*_fetch/**/*counter_collection.csv", recursive=True) + glob.glob("gpurun_out/prof/dtype_*_write/**/*counter_collection.csv", recursive=True):
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
for f in $(find gpurun_out/prof -path "*dtype*" -name "*kernel_stats.csv"); do echo "== $f"; head -4 $f | cut -c1-300; done
