"""World-N (ranks sharing one GPU) soak of the peer all-reduce with a per-size mismatch report (debug aid for csrc/p2p.hip; it found
the VMEM store-data hazard of round 6):   P2P_DEBUG_REPS=100 python tools/p2p_debug.py [world]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def worker(rank, world, port):
    import torch.distributed as dist
    from dcase2019_task4_amd import dist as sdist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ar = sdist.PeerAllReduce(214356, dev, dist.group.WORLD, timeout_s=3.0)
    g = torch.Generator().manual_seed(5 + rank)
    reps = int(os.environ.get("P2P_DEBUG_REPS", "30"))
    for n in (1000, 65536, 71451, 214356):
        tot = 0
        for rep in range(reps):
            x = torch.randn(n, generator=g).to(dev)
            parts = [torch.zeros_like(x) for _ in range(world)]
            dist.all_gather(parts, x)
            want = sum(parts[1:], parts[0].clone())
            y = x.clone()
            ar.all_reduce(y)
            torch.cuda.synchronize()
            ys = [torch.zeros_like(y) for _ in range(world)]
            dist.all_gather(ys, y)
            bad = (y != want) | torch.isnan(y)
            nb = int(bad.sum())
            tot += nb
            if nb:
                idx = bad.nonzero().flatten()[:8].tolist()
                other = ys[(rank + 1) % world]
                print(f"rank {rank} n {n} rep {rep}: {nb} wrong at {idx}; got {[float(y[i]) for i in idx[:4]]} want {[float(want[i]) for i in idx[:4]]} "
                      f"mine {[float(x[i]) for i in idx[:4]]} peer-input {[float(parts[(rank + 1) % world][i]) for i in idx[:4]]} peer-result-ok "
                      f"{[bool(other[i] == want[i]) for i in idx[:4]]} errors {ar.errors()}", flush=True)
            dist.barrier()
        print(f"rank {rank} n {n}: {tot} wrong elements over {reps} calls", flush=True)
    ar.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    import torch.multiprocessing as mp
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mp.spawn(worker, args=(world, port), nprocs=world, join=True)
