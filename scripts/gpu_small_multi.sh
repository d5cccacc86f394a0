#!/bin/bash
# host-side cost of one multi-rank transpose per transport: tiny grid, 4 ranks sharing the GPU
export HSA_ENABLE_IPC_MODE_LEGACY=0
p=29600
for b in peer peer_pl peer_sm; do
  p=$((p+1))
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 4 --size 64 --steps 200 --warmup 10 --backend $b --pdims 2 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$b', 'ms/cycle', d['ms_per_step'], d['config']['per_op_split'])"
done
