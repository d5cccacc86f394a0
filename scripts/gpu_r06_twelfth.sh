#!/bin/bash
# Round 6, twelfth call: kernels after the load restructuring (lines fixed, window, rotate), whole suite, bench line,
# rocprofv3 kernel stats + PMC passes of the bench and of the two new kernels.
cd "$(dirname "$0")/.."
O=gpurun_out/r06_twelfth; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
REPO=$PWD
( time timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dense_rows.py tests/test_gpu_transpose.py -q -m gpu -x ) > $O/kernel_tests.log 2>&1; tail -4 $O/kernel_tests.log | cut -c1-200
( timeout 150 python scripts/probe/window_walk_ab.py 2>&1 | grep "^{" ) > $O/lines_ab.jsonl; cut -c1-500 $O/lines_ab.jsonl
( time timeout 1500 python -m pytest tests -q -m gpu --durations=30 --junitxml=$O/junit.xml ) > $O/gpu_suite.log 2>&1; tail -6 $O/gpu_suite.log | cut -c1-300
grep -E "^(FAILED|ERROR)" $O/gpu_suite.log | head -20
( time timeout 400 python bench.py ) > $O/bench.log 2>&1; grep -E '^\{' $O/bench.log | tail -1 > $O/bench_n1.json
python - <<'PY'
import json
r = json.load(open("gpurun_out/r06_twelfth/bench_n1.json"))
print(r["ms_per_step"], r["roofline"]["frac"], r["stats"].get("in_place_cycle_ms"))
h = r["extra"]["halo_pencil_transposes"]
print({k: (v["ms"], v["frac"]) for k, v in h["per_layout"]["contiguous"].items()}, {k: (v["ms"], v["frac"]) for k, v in h["config5_pencil_contiguous"]["per_op"].items()})
PY
bash scripts/gpu_profile.sh > $O/gpu_profile.log 2>&1; tail -5 $O/gpu_profile.log | cut -c1-200
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof/new_kernels_trace -o k -- python $REPO/scripts/probe/halo_pencil_and_in_place.py > $REPO/$O/new_kernels_trace.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/prof/new_kernels_fetch -o k -- python $REPO/scripts/probe/halo_pencil_and_in_place.py > $REPO/$O/new_kernels_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/prof/new_kernels_write -o k -- python $REPO/scripts/probe/halo_pencil_and_in_place.py > $REPO/$O/new_kernels_write.log 2>&1 )
for f in $(find gpurun_out/prof/new_kernels_trace -name "*kernel_stats.csv"); do cp $f $O/new_kernels_kernel_stats.csv; head -8 $f | cut -c1-220; done
python - <<'PY'
import csv, glob, collections
for what in ("fetch", "write"):
    for f in glob.glob("gpurun_out/prof/new_kernels_%s/**/*counter_collection.csv" % what, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            k = (row.get("Kernel_Name", "")[:90], row.get("Counter_Name"))
            agg[k][0] += 1
            agg[k][1] += float(row.get("Counter_Value", 0))
        with open("gpurun_out/r06_twelfth/new_kernels_%s_summary.csv" % what, "w") as o:
            o.write("kernel,counter,dispatches,mean_per_dispatch\n")
            for (k, c), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                o.write('"%s",%s,%d,%.1f\n' % (k, c, n, s / n))
        print(open("gpurun_out/r06_twelfth/new_kernels_%s_summary.csv" % what).read()[:1200])
PY
find gpurun_out/prof -name "*kernel_trace.csv" -size +1M -delete; find gpurun_out/prof -name "*counter_collection.csv" -size +1M -delete
