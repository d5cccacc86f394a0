#!/bin/bash
# queue counts per setting (evidence for DESIGN section 9 B), halo face-copy counters (read amplification of dim 0), bench N=1
mkdir -p gpurun_out/final3
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/final3
REPO=$PWD

( STRESS_ITERS=5 timeout 1200 bash scripts/probe/count_queues.sh ) > $O/count_queues.log 2>&1; cat $O/count_queues.log | cut -c1-220
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $REPO/$O/halo_$c -o halo -- python $REPO/benchmark/halo_bench.py > $REPO/$O/halo_$c.log 2>&1
done
cd $REPO
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/final3/halo_*/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        k = (row.get("Kernel_Name")[:60], row.get("Counter_Name"), row.get("Grid_Size"))
        agg[k][0] += 1
        agg[k][1] += float(row.get("Counter_Value", 0))
    with open(f.replace("counter_collection.csv", "counter_summary.csv"), "w") as o:
        o.write("kernel,counter,grid,dispatches,mean_per_dispatch\n")
        for (k, c, g), (n, s) in sorted(agg.items()):
            o.write('"%s",%s,%s,%d,%.1f\n' % (k, c, g, n, s / n))
    print(open(f.replace("counter_collection.csv", "counter_summary.csv")).read()[:1200])
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
( timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ); cut -c1-400 $O/bench_n1.json
export CUDECOMP_PEER_TIMEOUT=30
( time timeout 2400 python tests/test_gpu_runner_cases.py --full --ngpu8 ) > $O/reference_sweep_full_ngpu8.log 2>&1; tail -3 $O/reference_sweep_full_ngpu8.log | cut -c1-200
