#!/bin/bash
# First GPU call of the next round (prepared at the end of round 5): the state at HEAD on a fresh box -- the whole -m gpu suite in
# ONE run (round 5 closed with it split over three calls), the bench line with rocprofv3 kernel stats, and the two harness
# measurements the round's notes end with (profiles/r05_tuning.md section 3): the window kernel with the run walk over batch planes,
# the library's window kernel with its walk forced.
cd "$(dirname "$0")/.."
O=gpurun_out/next_first; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD
REPO=$PWD
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 ) > $O/gpu_suite.log 2>&1; tail -14 $O/gpu_suite.log | cut -c1-200
( time timeout 400 python bench.py ) > $O/bench.log 2>&1; grep -E '^\{' $O/bench.log | tail -1 > $O/bench_n1.json
python -c "import json; r = json.load(open('$O/bench_n1.json')); print(r['ms_per_step'], r['roofline']['frac'], json.dumps(r['extra'].get('halo_pencil_transposes'))[:800])"
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace -o bench -- \
     python $REPO/bench.py --steps 5 --warmup 3 --cpu-sample 0 --no-extras > $REPO/$O/trace.log 2>&1 )
for f in $(find $O/trace -name "*kernel_stats.csv"); do head -5 $f | cut -c1-200; done
find $O/trace -name "*kernel_trace.csv" -size +1M -delete
[ -x scripts/tune/tune_misaligned ] && ( TUNE_FROM=1 timeout 120 scripts/tune/tune_misaligned ) > $O/tune_misaligned_runs.log 2>&1
for w in "" 0 1; do
  ( [ -n "$w" ] && export CUDECOMP_TILE_WALK=$w; CUDECOMP_AMD_LIBRARY=$PWD/cudecomp_amd/lib_tuning/libcudecomp.so timeout 60 python scripts/probe/window_walk_ab.py 2>&1 | grep "^{" ) >> $O/window_walk_ab.jsonl
done
cat $O/window_walk_ab.jsonl | cut -c1-400
