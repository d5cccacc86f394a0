import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.mp import run_ranks
import cudecomp_amd as cd


def body(rank, nranks, args):
    import torch
    from tests import gpu_bodies as B
    t0 = time.time()
    h = B._handle(rank)
    gd = cd.cudecompGridDescCreate(h, cd.make_config((64, 64, 64), (1, nranks)))
    out = []
    for mib in args["sizes_mib"]:
        t = time.time()
        p = cd.cudecompMalloc(h, gd, mib << 20)
        t1 = time.time() - t
        bad = cd.cudecompExtPeerProbe(h, p, mib << 20)
        print('[r%d] %d MiB probe mismatches %d' % (rank, mib, bad), flush=True)
        cd.cudecompFree(h, gd, p)
        out.append((mib, round(t1, 3), 'mismatches=%d' % bad))
        print("[r%d] %d MiB malloc %.3fs free %.3fs" % (rank, mib, t1, time.time() - t - t1), flush=True)
    return out


if __name__ == "__main__":
    n = int(sys.argv[1])
    sizes = [int(x) for x in sys.argv[2].split(",")]
    try:
        res = run_ranks(n, "scripts.ipc_probe", "body", {"sizes_mib": sizes}, timeout=int(sys.argv[3]))
        print(n, res[0])
    except AssertionError as e:
        print(str(e)[-3000:])
