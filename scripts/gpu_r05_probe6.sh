#!/bin/bash
# round 5: partial_probe with the dense linear walk (whole lines across row boundaries, gap bytes rewritten unchanged)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_probe6; mkdir -p $O
( time timeout 300 scripts/tune/partial_probe ) > $O/partial_probe_dense.log 2>&1
cat $O/partial_probe_dense.log | cut -c1-200
