#!/bin/bash
# the driver's multi-GPU bench command lines, with all ranks sharing the one GPU of the test box
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
for n in 8 4 2; do
echo "== N=$n"
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2952$n bench.py --gpus $n --steps 3 --warmup 1 ) 2>&1
done
} > gpurun_out/bench_full_multi.log 2>&1
grep -E "^\{|SELECTED|real|WARN|Error|error|Traceback|failed" gpurun_out/bench_full_multi.log | cut -c1-1800
