#!/bin/bash
# does a NINTH process with a live GPU context (the pytest process of the suite) bring the rare wrong case back?
mkdir -p gpurun_out/parent
export HSA_ENABLE_IPC_MODE_LEGACY=0 CUDECOMP_PEER_TIMEOUT=30
O=gpurun_out/parent
( time STRESS_PARENT_CONTEXT=1 timeout 400 python scripts/probe/stress_eight_ranks.py mix 60 ) > $O/with_parent_context.log 2>&1; grep -E "iterations failed|FAILED" $O/with_parent_context.log | cut -c1-200 | tail -4
( time timeout 300 python scripts/probe/stress_eight_ranks.py mix 60 ) > $O/without.log 2>&1; grep -E "iterations failed|FAILED" $O/without.log | cut -c1-200 | tail -4
