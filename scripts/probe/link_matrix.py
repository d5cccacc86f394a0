"""Link probe of every ordered GPU pair of the node, both engines: GB/s of a one-direction 256-MiB copy i -> j with the
copy engines (hipMemcpyPeerAsync, what CUDECOMP_PEER_COPY_ENGINE=sdma uses) and with compute-unit stores (the library's
row-copy kernel running on GPU i, destination in GPU j's memory: what the fused put and kernel copies do).  One process;
prints one JSON object {"gpus": N, "sdma": [[...]], "cu": [[...]]} (diagonal = a copy inside one GPU).

    python scripts/probe/link_matrix.py [MiB] > gpurun_out/first_multi_gpu/link_matrix.json

On a one-GPU box the matrix is 1 x 1 (the local copy rate) -- enough to show that the tool runs."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cudecomp_amd as cd  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = torch.cuda.device_count()
nel = mib * (1 << 20) // 8
bufs = []
for d in range(n):
    torch.cuda.set_device(d)
    bufs.append((torch.ones(nel, dtype=torch.float64, device="cuda:%d" % d), torch.zeros(nel, dtype=torch.float64, device="cuda:%d" % d)))


def timed(fn, dev, reps=5):
    torch.cuda.set_device(dev)
    fn()
    fn()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    return round(nel * 8 / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9, 1)


sdma = [[None] * n for _ in range(n)]
cu = [[None] * n for _ in range(n)]
for i in range(n):
    for j in range(n):
        src, dst = bufs[i][0], bufs[j][1]
        try:
            sdma[i][j] = timed(lambda: dst.copy_(src, non_blocking=True), i)  # (also turns peer access on)
        except Exception as e:  # noqa: BLE001
            sdma[i][j] = "error: %s" % str(e)[:80]
        try:
            torch.cuda.set_device(i)
            st = torch.cuda.current_stream().cuda_stream
            cu[i][j] = timed(lambda: cd.cudecompExtMove3D(src.data_ptr(), dst.data_ptr(), 8, (nel, 1, 1), (1, nel, nel), (1, nel, nel),
                                                          0, st), i)
            torch.cuda.synchronize(i)
            if not bool((dst[:1024] == 1).all()):
                cu[i][j] = "wrong data"
            dst.zero_()
        except Exception as e:  # noqa: BLE001
            cu[i][j] = "error: %s" % str(e)[:80]
print(json.dumps({"gpus": n, "MiB": mib, "unit": "GB/s, one direction", "sdma": sdma, "cu": cu}))
