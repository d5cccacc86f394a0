#!/bin/bash
# round 5: the dense walk with aligned loads + lane shift (probe only)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_dense5; mkdir -p $O
( timeout 300 scripts/tune/partial_probe ) > $O/partial_probe_dense2.log 2>&1; grep -v "^library" $O/partial_probe_dense2.log | grep -A8 "^==" | grep "^==\|dense\|pieces" | cut -c1-160
