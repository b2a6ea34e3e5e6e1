"""World-size-2 gloo tests (CPU) of the data-parallel plumbing in dcase2019_task4_amd/dist.py:
stream-wise batch sharding, positional masks, bucketed flat-gradient all-reduce.  The per-rank
gradient function here is the oracle (test infrastructure standing in for the HIP backward, which
needs a GPU); the code under test is the sharding / collective logic, which is backend-agnostic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dcase2019_task4_amd import _lib
from dcase2019_task4_amd import dist as sdist
from oracle import ref_cpu, synth

B_GLOBAL, T = 8, 64
BATCH_SIZES = [2, 4, 2]          # [weak | unlabeled | strong] = [B/4, B/2, B/4]  (main.py:240)


def test_stream_wise_sharding_keeps_the_batch_contract():
    idx = list(range(100, 108))
    s0 = sdist.shard_indices(idx, BATCH_SIZES, 0, 2)
    s1 = sdist.shard_indices(idx, BATCH_SIZES, 1, 2)
    assert s0 == [100, 102, 103, 106] and s1 == [101, 104, 105, 107]      # one weak, two unlabeled, one strong each
    assert sorted(s0 + s1) == idx
    wm, sm = sdist.local_masks(BATCH_SIZES, 2)
    assert wm == slice(1) and sm == slice(3, 4)
    tgt, gwm, gsm = synth.make_target(0, B_GLOBAL, T // 8)
    for r in range(2):
        (lt,) = sdist.shard_batch([tgt], BATCH_SIZES, r, 2)
        assert (lt[wm] >= 0).all() and (lt[sm] >= 0).all() and (lt[1:3] == -1).all()
    with pytest.raises(ValueError):
        sdist.local_batch_sizes([3, 4, 2], 2)
    assert sdist.local_masks([6, 18], 2) == (slice(3), None)                   # --no_synthetic layout (main.py:243-245)


def _layout():
    offs = _lib.param_layout(_lib.make_dims(1, 16, p_drop=0.0))
    shapes = list(ref_cpu.param_shapes().values())
    return [(offs[i], offs[i + 1], tuple(s)) for i, s in enumerate(shapes)]


def _local_grads(rank, world):
    params = synth.make_params(0)
    x = synth.make_input(1, B_GLOBAL, T)
    xe = synth.make_input(2, B_GLOBAL, T)
    tgt, _, _ = synth.make_target(0, B_GLOBAL, T // 8)
    lx, lxe, lt = sdist.shard_batch([x, xe, tgt], BATCH_SIZES, rank, world)
    wm, sm = sdist.local_masks(BATCH_SIZES, world)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    pe = synth.make_params(1)
    with torch.no_grad():
        se, we = ref_cpu.crnn_forward(pe, lxe, True, ref_cpu.new_bn_state())
    s, w = ref_cpu.crnn_forward(p, lx, True, ref_cpu.new_bn_state())
    loss, _ = ref_cpu.mean_teacher_loss(s, w, se, we, lt, wm, sm, 0.5)
    g = torch.autograd.grad(loss, list(p.values()))
    flat = torch.cat([t.reshape(-1) for t in g]).contiguous()
    return flat, float(loss)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        flat, loss = _local_grads(rank, world)
        layout = _layout()
        assert layout[-1][1] == flat.numel()
        tail, head = sdist.grad_buckets(layout)
        assert tail[0] == head[1] and head[0] == 0 and tail[1] == flat.numel()
        w1 = sdist.allreduce_bucket(flat, *tail, group=None, async_op=True)       # GRU + heads bucket first
        w2 = sdist.allreduce_bucket(flat, *head, group=None, async_op=True)
        w1.wait(); w2.wait()
        flat /= world
        # identical replicas afterwards: Adam on the averaged gradient gives the same parameters everywhere
        p = torch.cat([v.reshape(-1) for v in synth.make_params(0).values()])
        pp, gg = {"p": p.clone()}, {"p": flat}
        ref_cpu.adam_step(pp, gg, {"p": torch.zeros_like(p)}, {"p": torch.zeros_like(p)}, 1)
        gathered = [torch.zeros_like(p) for _ in range(world)]
        dist.all_gather(gathered, pp["p"])
        same = all(torch.equal(gathered[0], t) for t in gathered)
        t0 = torch.full((4,), float(rank))
        sdist.broadcast_parameters([t0], None, src=0)
        q.put((rank, flat.numpy(), loss, same, float(t0.sum())))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
def test_two_rank_gloo_allreduce_matches_single_process_average():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single process: the two shards one after the other, then the plain average
    torch.set_num_threads(4)
    g0, l0 = _local_grads(0, world)
    g1, l1 = _local_grads(1, world)
    want = ((g0 + g1) / 2).numpy()
    for rank, flat, loss, same, bsum in res:
        np.testing.assert_allclose(flat, want, rtol=1e-4, atol=2e-6)   # thread-count-dependent fp32 summation order on the CPU side
        assert same and bsum == 0.0
        assert loss == pytest.approx([l0, l1][rank], rel=1e-6)
    np.testing.assert_array_equal(res[0][1], res[1][1])
