"""The library's peer all-reduce (csrc/p2p.hip) alone, on ONE GPU: us per captured call for the step's two message sizes at
world 1 (launch structure only: a rank's own slice never goes through the communication buffers) and at world 2 with both
ranks sharing the GPU (every byte does go through uncached fine-grained memory - the closest this box gets to a peer).

    python tools/p2p_bench.py [--world 2] [--wgs 0,32,64,128]   ->  one line per (world, size, workgroups)
"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
SIZES = (("base_857KB", 214356), ("base_tail_349KB", 87344), ("base_head_508KB", 127012), ("wide_8.5MB", 2132628))


def worker(rank, world, port, wgs, out):
    import torch.distributed as dist
    from dcase2019_task4_amd import dist as sdist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rows = []
    for wg in wgs:
        ar = sdist.PeerAllReduce.create(SIZES[-1][1], dev, dist.group.WORLD, workgroups=wg)
        assert ar is not None, sdist.PeerAllReduce.last_error
        for label, n in SIZES:
            us = ar.time_us(n, iters=200)
            rows.append((world, label, wg or ar.workgroups, round(us, 2), round(4 * n / us * 1e-3, 1)))
        assert ar.errors(reduce=True) == 0
        ar.close()
    if rank == 0:
        out.put(rows)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", default="1,2")
    ap.add_argument("--wgs", default="0,32,64,128")
    a = ap.parse_args()
    import socket
    import torch.multiprocessing as mp
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ctx = mp.get_context("spawn")
    for world in [int(v) for v in a.world.split(",")]:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        out = ctx.SimpleQueue()
        ps = [ctx.Process(target=worker, args=(r, world, port, [int(v) for v in a.wgs.split(",")], out)) for r in range(world)]
        for p in ps:
            p.start()
        for p in ps:
            p.join(600)
        assert all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
        for w, label, wg, us, gbs in out.get():
            print(f"world {w} {label:16s} workgroups {wg:3d}: {us:8.2f} us per call, {gbs:7.1f} GB/s (message bytes / time)")


if __name__ == "__main__":
    main()
