#!/bin/bash
# round 3 measurements: flags comparison, bench N=1, rocprofv3 stats + PMC of the bench, local phases (incl. the staged
# launch form), halo bench with neighbours, shared-GPU flow of the N>1 bench line, overlap timeline
mkdir -p gpurun_out/r03
export HSA_ENABLE_IPC_MODE_LEGACY=0
export CUDECOMP_PEER_TIMEOUT=60
O=gpurun_out/r03
( time timeout 300 python scripts/probe/stress_eight_ranks.py mix 12 ) > $O/stress_devflags.log 2>&1; grep "iterations failed" $O/stress_devflags.log | cut -c1-200
( time timeout 300 python scripts/probe/stress_eight_ranks.py mix 12 CUDECOMP_FLAGS_IN_HOST_MEMORY=1 ) > $O/stress_hostflags.log 2>&1; grep "iterations failed" $O/stress_hostflags.log | cut -c1-200
( timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ); cut -c1-600 $O/bench_n1.json
bash scripts/gpu_profile.sh > $O/profile.log 2>&1; python scripts/summarize_profiles.py r03 > $O/profile_summary.log 2>&1; cat $O/profile_summary.log | head -20
( timeout 900 python scripts/probe/local_phases.py > $O/local_phases.json 2> $O/local_phases.err ); python scripts/probe/summarize_local_phases.py $O/local_phases.json 2>/dev/null | head -40
( timeout 900 python scripts/probe/halo_bench_ranks.py > $O/halo_bench_ranks.json 2> $O/halo_bench_ranks.err ); head -c 1500 $O/halo_bench_ranks.json
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29524 bench.py --gpus 4 --steps 3 --warmup 1 ) > $O/bench_n4_shared.log 2>&1; grep -E "^\{" $O/bench_n4_shared.log | cut -c1-1500
bash scripts/gpu_profile_overlap.sh > $O/overlap.log 2>&1; tail -5 $O/overlap.log
