#!/bin/bash
# final validation of the tree: the full -m gpu suite, then bench.py --gpus 8 / 2 on the one-GPU box (flow check)
mkdir -p gpurun_out/final4
export HSA_ENABLE_IPC_MODE_LEGACY=0
export CUDECOMP_PEER_TIMEOUT=30
O=gpurun_out/final4
( time timeout 2400 python -m pytest tests -q -m gpu --durations=8 -p no:cacheprovider ) > $O/gpu_tests.log 2>&1
tail -6 $O/gpu_tests.log | cut -c1-200
for n in 8 2; do
( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2953$n bench.py --gpus $n --steps 3 --warmup 1 ) > $O/bench_n${n}_shared.log 2>&1; grep -E "^\{" $O/bench_n${n}_shared.log | cut -c1-400
done
