"""-m gpu: the data-parallel mean-teacher step at WORLD SIZE 2 - two processes, MeanTeacherStep itself on each.

On the one-GPU box both ranks share cuda:0 and talk over gloo (which moves CUDA tensors through host memory); on a box
with >= 2 GPUs they take one GPU each and talk over nccl (= RCCL).  Checks, for both schedules, eager and hipGraph:
  * the all-reduced gradient equals the SUM of the two ranks' single-process gradients (each rank also runs a plain
    single-process step on its own shard with the same rank-folded dropout seed), i.e. the update uses their mean;
  * the replicas (student, teacher, Adam moments) stay bit-identical over 5 steps although every rank sees different
    clips and different dropout masks;
  * the batch is sharded stream-wise (dist.shard_batch: main.py:238-247's [weak|unlabeled|strong] contract).
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, schedule, graph, out, Bg=16, T=128, C=64, H=64, dtype="f32", collective="pg"):
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    import torch.distributed as dist
    from dcase2019_task4_amd import dist as sdist
    from dcase2019_task4_amd.train import MeanTeacherStep
    from oracle import synth
    from tests import gpu_util as gu
    n_dev = torch.cuda.device_count()
    backend = "nccl" if n_dev >= world else "gloo"
    dev = torch.device("cuda", rank if n_dev >= world else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        sizes = [Bg // 4, Bg // 2, Bg // 4]
        tgt_g, _, _ = synth.make_target(1, Bg, T // 8)
        wm, sm = sdist.local_masks(sizes, world)
        B = Bg // world
        steps = 5
        batches = []
        for i in range(steps):
            xg, xeg = synth.make_input(60 + i, Bg, T), synth.make_input(70 + i, Bg, T)
            batches.append([t.to(dev) for t in sdist.shard_batch([xg, xeg, tgt_g], sizes, rank, world)])

        def models(s0, s1):
            s, _ = gu.make_model(s0, dropout=0.5, device=dev, C=C, H=H, mfma_dtype=dtype)
            t, _ = gu.make_model(s1, dropout=0.5, device=dev, C=C, H=H, mfma_dtype=dtype)
            s.train(); t.train()
            return s, t
        # single-process step on this rank's shard, same folded seed as the data-parallel rank will use
        s1, t1 = models(0, 1)
        fold = sdist.fold_rank_seed(1234, rank)
        ref = MeanTeacherStep(s1, t1, B, T, 40, wm, sm, seed=fold, use_graph=False)
        ref.step(*batches[0])
        torch.cuda.synchronize()
        g_local = ref.grads.clone()
        gl = [torch.zeros_like(g_local) for _ in range(world)]
        dist.all_gather(gl, g_local)
        g_sum = sum(gl[1:], gl[0].clone())
        # data-parallel: rank 1 starts from DIFFERENT weights - the constructor must broadcast rank 0's
        s2, t2 = models(0, 1) if rank == 0 else models(7, 8)
        dp = MeanTeacherStep(s2, t2, B, T, 40, wm, sm, seed=1234, use_graph=graph, process_group=dist.group.WORLD,
                             dp_schedule=schedule, collective=None if collective == "auto" else collective)
        if collective == "auto":
            # the default: both collectives TIMED on the step's two buckets, the faster kept (dist.choose_collective); the record
            # of that decision is what bench.py prints.  gloo cannot be captured, RCCL at world 2 is a real candidate
            rec = dp.collective_record
            assert rec is not None and rec["choice"] == dp.collective and len(rec["buckets_bytes"]) == 2, rec
            assert rec["p2p_us"] is not None and all(u > 0 for u in rec["p2p_us"]), rec
            if backend == "gloo":
                assert dp.collective == "p2p" and rec["pg_us"] is None, rec
            collective = dp.collective
        assert dp.dp and dp.world == world and dp.rank == rank and dp.collective == collective
        if collective == "p2p":
            # the all-reduce is the library's own kernel over peer-mapped memory: capturable on any backend, so the DEFAULT
            # schedule - collectives inside the step's hipGraph - runs at world 2 on this one-GPU box too
            assert dp._p2p is not None and dp._p2p.world == world
            if graph and schedule == "overlap":
                assert dp.dp_capture is True and dp.dp_schedule == "overlap" and dp.single_graph
        if backend == "nccl":
            # first contact with real RCCL at world > 1: the default must be the captured-collective overlap schedule
            assert dist.get_backend() == "nccl"
            if schedule == "overlap" and graph:
                assert dp.dp_capture is True and dp.dp_schedule == "overlap", (dp.dp_capture, dp.dp_schedule, dp._capture_error)
        seeds = []
        if graph:
            dp._warm = 2
        st0 = dp.read_state()
        seeds += [st0.seed_student, st0.seed_teacher]
        dp.step(*batches[0])
        torch.cuda.synchronize()
        err = float((dp.grads - g_sum).abs().max())
        scale = float(g_sum.abs().max())
        # (reduced-precision modes: the BatchNorm sums of the bf16 conv kernels are fp64 atomics - a last-bit difference in a
        # statistic can flip a bf16 rounding downstream, so two runs of the same step agree to bf16 noise, not to the bit)
        tol = 2e-6 if dtype == "f32" else 5e-3
        assert err <= tol * scale + 1e-9, ("all-reduced gradient != sum of the single-rank gradients", err, scale)
        if H == 256:
            dp.check_health()                      # the cluster recurrence's sticky timeout counter, under two processes
        # after step 1 the student is what the single-process update would be with the MEAN gradient
        for i in range(1, steps):
            sti = dp.read_state()
            seeds += [sti.seed_student, sti.seed_teacher]
            dp.step(*batches[i])
        torch.cuda.synchronize()
        # every (rank, step, model) draws from its OWN Philox key: no key of one rank appears in another rank's
        # sequence (the old fold, seed + rank * PHI with per-step stride PHI, made rank 1's student stream rank 0's
        # teacher stream of the same step)
        all_seeds = [None] * world
        dist.all_gather_object(all_seeds, seeds)
        flat = [k for r in all_seeds for k in r]
        assert len(set(flat)) == len(flat) == 2 * steps * world, "dropout keys collide across ranks / steps"
        for name, t in (("student", s2._flat), ("teacher", t2._flat), ("exp_avg", dp.exp_avg), ("exp_avg_sq", dp.exp_avg_sq)):
            tl = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(tl, t.contiguous())
            for r in range(1, world):
                assert torch.equal(tl[0], tl[r]), f"replicas diverged: {name}"
        st = dp.read_state()
        assert st.global_step == steps
        m_loc, m_glob = dp.meters(), dp.meters(reduce=True)
        ml = [None] * world
        dist.all_gather_object(ml, m_loc["loss"])
        assert abs(m_glob["loss"] - sum(ml) / world) < 1e-6
        # different ranks really drew different masks / saw different clips
        sl = [torch.zeros_like(dp.strong) for _ in range(world)]
        dist.all_gather(sl, dp.strong)
        assert not torch.equal(sl[0], sl[1])
        dp.check_health()                          # (p2p: no cross-rank wait timed out)
        if rank == 0:
            out.put(("ok", backend + "+" + collective, err / (scale + 1e-30)))
    except Exception as e:      # surface the failure in the parent
        import traceback
        out.put(("fail", rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_world2(target, args, world=2):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (out,)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    alive = [p for p in procs if p.is_alive()]
    for p in alive:
        p.kill()
    assert not alive, "data-parallel worker hung"
    msgs = []
    while not out.empty():
        msgs.append(out.get())
    fails = [m for m in msgs if m[0] == "fail"]
    assert not fails, fails[0][2]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    ok = [m for m in msgs if m[0] == "ok"]
    assert ok, "rank 0 reported nothing"
    return ok[0]


def _worker_entry(rank, world, port, schedule, graph, Bg, T, C, H, dtype, collective, out):
    _worker(rank, world, port, schedule, graph, out, Bg, T, C, H, dtype, collective)


# 4th case: BASELINE.json configs[3]'s per-rank composition - 64 clips per rank as [16 | 32 | 16] of a global
# [32 | 64 | 32] at T = 628 (main.py:238-247, DataLoad.py:562-571); last two: configs[4]'s model (wide CRNN, bf16 arithmetic and
# storage, cluster recurrence) and its bf16x3 mode under the data-parallel step
@pytest.mark.parametrize("schedule,graph,Bg,T,C,H,dtype,collective",
                         [("overlap", False, 16, 128, 64, 64, "f32", "pg"), ("overlap", True, 16, 128, 64, 64, "f32", "pg"),
                          ("single", True, 16, 128, 64, 64, "f32", "pg"), ("overlap", True, 128, 628, 64, 64, "f32", "pg"),
                          ("overlap", True, 16, 216, 128, 256, "bf16", "pg"), ("overlap", True, 16, 216, 128, 256, "bf16x3", "pg"),
                          ("overlap", True, 16, 216, 64, 64, "bf16", "pg"),
                          # round 5: the same checks with the library's own all-reduce (csrc/p2p.hip) - collectives CAPTURED in the
                          # step's hipGraph at world 2 (dp_capture True), which RCCL cannot do on one GPU and gloo cannot do at all
                          ("overlap", True, 16, 128, 64, 64, "f32", "p2p"), ("overlap", False, 16, 128, 64, 64, "f32", "p2p"),
                          ("single", True, 16, 128, 64, 64, "f32", "p2p"), ("overlap", True, 128, 628, 64, 64, "f32", "p2p"),
                          ("overlap", True, 16, 216, 128, 256, "bf16", "p2p"), ("overlap", True, 16, 216, 64, 64, "bf16", "p2p"),
                          ("overlap", True, 16, 216, 128, 256, "f16", "p2p"), ("overlap", True, 16, 216, 64, 64, "f16", "pg"),
                          # round 6: collective="auto" decides by TIME, not by "p2p works"
                          ("overlap", True, 16, 128, 64, 64, "f32", "auto")])
def test_mean_teacher_step_world2(schedule, graph, Bg, T, C, H, dtype, collective):
    ok = _run_world2(_worker_entry, (schedule, graph, Bg, T, C, H, dtype, collective))
    print(f"[dp world 2] backend {ok[1]} schedule {schedule} graph {graph} global batch {Bg} T {T} C {C} H {H} {dtype}: "
          f"|allreduce - sum| / max = {ok[2]:.2e}")


def _worker_peer_allreduce(rank, world, port, mode, out):
    """dist.PeerAllReduce alone: sizes from 1 float to the buffer size (not multiples of 4, of the workgroup count or of the
    world size), eager and replayed from a hipGraph 20 times with fresh data, two messages back to back (the two gradient
    buckets of a step) - every rank must end with bit-for-bit the RANK-ORDER sum.  mode "timeout": rank 1 skips one call; rank
    0's waits must time out (bounded, 3 s), raise the sticky error counter and return - never hang the GPU."""
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    import torch.distributed as dist
    from dcase2019_task4_amd import dist as sdist
    dev = torch.device("cuda", rank if torch.cuda.device_count() >= world else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        n_max = 214356                              # the gradient buffer of cfg.crnn_kwargs
        # (the wait budget is 600 s by default - skew is not failure; the timeout leg sets 2 s)
        ar = sdist.PeerAllReduce.create(n_max, dev, dist.group.WORLD, timeout_s=2.0 if mode == "timeout" else None)
        assert ar is not None, sdist.PeerAllReduce.last_error
        g = torch.Generator().manual_seed(99 + rank)

        def expect(x):
            parts = [torch.zeros_like(x) for _ in range(world)]
            dist.all_gather(parts, x)
            want = parts[0].clone()
            for p in parts[1:]:
                want += p
            return want
        if mode == "timeout":
            x = torch.randn(1000, generator=g).to(dev)
            assert ar.poll() == 0
            if rank == 0:
                ar.all_reduce(x)                    # rank 1 never issues this one
            torch.cuda.synchronize()
            dist.barrier()
            n_err = ar.errors()
            assert (n_err > 0) == (rank == 0), (rank, n_err)
            assert ar.errors(reduce=True) > 0
            # the host-visible word the step driver polls without synchronising, and the poisoned output: a timed-out
            # all-reduce must not look like a result
            assert (ar.poll() > 0) == (rank == 0)
            if rank == 0:
                assert bool(torch.isnan(x).all()), "timed-out all-reduce left numbers in its output"
            # rank 1 joins LATE (its first call = epoch 1, which rank 0 has already given up on): it must not hang, and what
            # it gets is poisoned too (rank 0 broadcast NaN for its slice) - nobody trains on half an all-reduce
            if rank == 1:
                y = torch.randn(1000, generator=g).to(dev)
                ar.all_reduce(y)
                torch.cuda.synchronize()
                assert bool(torch.isnan(y).any())
            if rank == 0:
                out.put(("ok", "gloo", float(n_err)))
            return
        for n in (1, 3, 4, 5, 127, 1000, 65537, 127012, 87344, n_max):
            x = torch.randn(n, generator=g).to(dev)
            want = expect(x)
            y = x.clone()
            ar.all_reduce(y)
            torch.cuda.synchronize()
            assert torch.equal(y, want), (n, float((y - want).abs().max()))
        # a slice of a larger buffer (the bucket form: flat[lo:hi]); lo a multiple of 4 floats as the parameter layout gives
        flat = torch.randn(n_max, generator=g).to(dev)
        lo, hi = 127012, n_max
        want = flat.clone()
        want[lo:hi] = expect(flat[lo:hi].clone())
        ar.all_reduce(flat, lo, hi)
        torch.cuda.synchronize()
        assert torch.equal(flat, want)
        # captured, two buckets back to back, replayed with fresh data
        a, b = torch.zeros(127012, device=dev), torch.zeros(87344, device=dev)
        st = torch.cuda.Stream(device=dev)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st, capture_error_mode="thread_local"):
            ar.all_reduce(b)
            ar.all_reduce(a)
        for it in range(20):
            xa, xb = torch.randn(a.numel(), generator=g).to(dev), torch.randn(b.numel(), generator=g).to(dev)
            wa, wb = expect(xa), expect(xb)
            a.copy_(xa); b.copy_(xb)
            gr.replay()
            torch.cuda.synchronize()
            assert torch.equal(a, wa) and torch.equal(b, wb), it
        assert ar.errors(reduce=True) == 0
        ar.close()
        if rank == 0:
            out.put(("ok", "gloo", 1.0 if ar.fine_grained else 0.0))
    except Exception:
        import traceback
        out.put(("fail", rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["sums", "timeout"])
def test_peer_allreduce_world2_on_one_gpu(mode):
    ok = _run_world2(_worker_peer_allreduce, (mode,))
    print(f"[p2p all-reduce world 2] mode {mode}: {'fine-grained buffers' if ok[2] == 1.0 and mode == 'sums' else ok[2]}")


@pytest.mark.parametrize("world", [3, 4])
def test_peer_allreduce_other_world_sizes_on_one_gpu(world):
    """World 3 takes the any-W instantiation of the kernel (k_p2p_allreduce<0>: run-time rank loop, 32-bit chunk arithmetic),
    world 4 the compile-time one with two chunks x four ranks per load batch: the same size sweep, eager and replayed, every
    rank bit-for-bit the rank-order sum (three / four processes sharing this box's GPU over gloo)."""
    ok = _run_world2(_worker_peer_allreduce, ("sums",), world=world)
    print(f"[p2p all-reduce world {world}] sums: {'fine-grained buffers' if ok[2] == 1.0 else ok[2]}")


def _worker_skew(rank, world, port, collective, out):
    """Rank 1 falls 5 s behind between two steps (a checkpoint, a validation pass, a data-loader stall): the captured
    data-parallel step must simply wait - no timeout, no NaN, replicas bit-identical afterwards.  Round 5's kernel gave up
    after 3 s and trained on stale staging data."""
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    import time
    import torch.distributed as dist
    from dcase2019_task4_amd import dist as sdist
    from dcase2019_task4_amd.train import MeanTeacherStep
    from oracle import synth
    from tests import gpu_util as gu
    n_dev = torch.cuda.device_count()
    dev = torch.device("cuda", rank if n_dev >= world else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl" if n_dev >= world else "gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        Bg, T = 16, 128
        sizes = [Bg // 4, Bg // 2, Bg // 4]
        tgt_g, _, _ = synth.make_target(1, Bg, T // 8)
        wm, sm = sdist.local_masks(sizes, world)
        s, _ = gu.make_model(0, dropout=0.5, device=dev)
        t, _ = gu.make_model(1, dropout=0.5, device=dev)
        s.train(); t.train()
        dp = MeanTeacherStep(s, t, Bg // world, T, 40, wm, sm, seed=1234, use_graph=True, process_group=dist.group.WORLD,
                             collective=collective)
        assert dp.collective == collective
        for i in range(6):
            xg, xeg = synth.make_input(60 + i, Bg, T), synth.make_input(70 + i, Bg, T)
            b = [v.to(dev) for v in sdist.shard_batch([xg, xeg, tgt_g], sizes, rank, world)]
            if i == 4 and rank == 1:
                torch.cuda.synchronize()
                time.sleep(5.0)
            dp.step(*b)
        torch.cuda.synchronize()
        dp.check_health()
        assert dp._p2p is None or (dp._p2p.poll() == 0 and dp._p2p.errors(reduce=True) == 0)
        assert bool(torch.isfinite(dp.grads).all()) and bool(torch.isfinite(s._flat).all())
        for name, tns in (("student", s._flat), ("teacher", t._flat), ("exp_avg", dp.exp_avg)):
            tl = [torch.zeros_like(tns) for _ in range(world)]
            dist.all_gather(tl, tns.contiguous())
            assert torch.equal(tl[0], tl[1]), f"replicas diverged: {name}"
        dp.rendezvous()
        dp.close()
        if rank == 0:
            out.put(("ok", collective, 5.0))
    except Exception:
        import traceback
        out.put(("fail", rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def test_a_rank_that_falls_five_seconds_behind_is_waited_for():
    ok = _run_world2(_worker_skew, ("p2p",))
    print(f"[dp world 2] rank 1 slept {ok[2]} s before a step under collective {ok[1]}: no timeout, replicas bit-identical")


def _worker_step_stops_after_timeout(rank, world, port, out):
    """MeanTeacherStep.run() must RAISE on the step after a peer all-reduce ran out of its budget (rank 1 stops stepping):
    the pinned host word is polled on every run(), the gradients of the timed-out step are NaN."""
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    import torch.distributed as dist
    from dcase2019_task4_amd import _lib, dist as sdist
    from dcase2019_task4_amd.train import MeanTeacherStep
    from oracle import synth
    from tests import gpu_util as gu
    n_dev = torch.cuda.device_count()
    dev = torch.device("cuda", rank if n_dev >= world else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        Bg, T = 16, 128
        sizes = [Bg // 4, Bg // 2, Bg // 4]
        tgt_g, _, _ = synth.make_target(1, Bg, T // 8)
        wm, sm = sdist.local_masks(sizes, world)
        s, _ = gu.make_model(0, dropout=0.5, device=dev)
        t, _ = gu.make_model(1, dropout=0.5, device=dev)
        s.train(); t.train()
        dp = MeanTeacherStep(s, t, Bg // world, T, 40, wm, sm, seed=1234, use_graph=False, process_group=dist.group.WORLD,
                             collective="p2p")
        dp._p2p.set_timeout(1.0)
        xg, xeg = synth.make_input(60, Bg, T), synth.make_input(70, Bg, T)
        b = [v.to(dev) for v in sdist.shard_batch([xg, xeg, tgt_g], sizes, rank, world)]
        dp.step(*b)
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 0:
            dp.step(*b)                            # rank 1 does not take this step: every wait of rank 0 runs out (1 s each)
            torch.cuda.synchronize()
            assert bool(torch.isnan(dp.grads).any()), "the timed-out step left finite gradients"
            with pytest.raises(_lib.SedError, match="cross-rank waits ran out"):
                dp.run()
            with pytest.raises(_lib.SedError):
                dp.save_checkpoint("/tmp/never_written.pt")
            out.put(("ok", "p2p", 1.0))
        dist.barrier()
    except Exception:
        import traceback
        out.put(("fail", rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def test_step_raises_after_a_collective_timed_out():
    _run_world2(_worker_step_stops_after_timeout, ())


def _worker_frontend(rank, world, port, dtype, out):
    """WaveformFrontEnd under the data-parallel step: every rank streams its OWN waveforms and targets through feed() / flush();
    with RCCL (captured collectives) the next batch's extraction runs inside the step's hipGraph, with gloo the same protocol
    runs serially - either way the replicas must stay bit-identical and every batch is trained on exactly once."""
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    import numpy as np
    import torch.distributed as dist
    from dcase2019_task4_amd import dist as sdist
    from dcase2019_task4_amd.features import FeatureConfig, WaveformFrontEnd
    from dcase2019_task4_amd.train import MeanTeacherStep
    from oracle import synth
    from tests import gpu_util as gu
    n_dev = torch.cuda.device_count()
    backend = "nccl" if n_dev >= world else "gloo"
    dev = torch.device("cuda", rank if n_dev >= world else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        B, T, n_steps = 8, 628, 5
        sizes = [B * world // 4, B * world // 2, B * world // 4]
        wm, sm = sdist.local_masks(sizes, world)
        res = []
        for overlap in (False, True):
            s, _ = gu.make_model(0, dropout=0.5, device=dev, mfma_dtype=dtype)
            t, _ = gu.make_model(1, dropout=0.5, device=dev, mfma_dtype=dtype)
            s.train(); t.train()
            st = MeanTeacherStep(s, t, B, T, 40, wm, sm, seed=77, use_graph=True, process_group=dist.group.WORLD)
            fe = WaveformFrontEnd(st, np.zeros((B, 160000), np.float32), FeatureConfig.baseline_16k(), overlap=overlap, seed=5 + rank,
                                  fft_dtype="f32")
            assert fe.overlap == (overlap and st.single_graph)
            for k in range(n_steps):
                waves = np.stack([synth.make_wave(1000 * rank + 10 * k + i, 160000) for i in range(B)]).astype(np.float32)
                tg, _, _ = synth.make_target(50 + 7 * rank + k, B * world, T // 8)
                fe.feed(waves, sdist.shard_batch([tg], sizes, rank, world)[0])
            fe.flush()
            torch.cuda.synchronize()
            assert st.steps_done == n_steps and st.read_state().global_step == n_steps
            for name, tns in (("student", s._flat), ("teacher", t._flat), ("exp_avg", st.exp_avg)):
                tl = [torch.zeros_like(tns) for _ in range(world)]
                dist.all_gather(tl, tns.contiguous())
                assert torch.equal(tl[0], tl[1]), f"replicas diverged: {name}"
            assert np.isfinite(st.meters()["loss"])
            res.append((s._flat.clone(), fe.overlap))
        assert torch.equal(res[0][0], res[1][0]), "one-batch-ahead front-end under DP differs from the serial protocol"
        if rank == 0:
            out.put(("ok", backend, float(res[1][1])))
    except Exception:
        import traceback
        out.put(("fail", rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def test_waveform_front_end_world2():
    ok = _run_world2(_worker_frontend, ("bf16",))
    print(f"[dp world 2] WaveformFrontEnd under DP: backend {ok[1]}, extraction inside the step's graph: {bool(ok[2])}")


def test_bench_gpus_2_code_path_on_a_shared_gpu():
    """`bench.py --gpus 2` end to end - self-launch through torch.distributed.run, process group, data-parallel step with the
    default collective choice, the schedule A/B legs, configs[3] / configs[4] legs, the replica fingerprint - with the two ranks
    sharing this box's GPU over gloo (SED_BENCH_SHARE_GPU=1).  The number means nothing; what is checked is that the command the
    driver runs on an 8-GPU node produces ONE JSON line with the contract's fields, that the library's own all-reduce was chosen
    and captured, and that the replicas are still bit-identical after the timed steps."""
    import json
    import subprocess
    env = dict(os.environ, SED_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SED_POISON", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().split("\n")[-1])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["unit"] == "clips/s" and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 48 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["dp_schedule"] == "overlap" and d["config"]["dp_collectives"] == "captured"
    assert d["config"]["dp_collective"].startswith("library kernel over peer-mapped memory")
    dd = d["distributed"]
    assert dd["shared_gpu"] is True and dd["replicas_bit_identical_after_timed_steps"] is True
    assert set(dd["schedule_ab"]) == {"overlap_captured_rccl", "single_eager_rccl", "overlap_eager_rccl", "single_eager_p2p"}
    assert all("value" in v for v in dd["schedule_ab"].values()), dd["schedule_ab"]
    assert "value" in d["config3_ddp"] and d["config3_ddp"]["global_batch"] == 128
    assert "value" in d["config4_ddp"] and d["config4_ddp"]["dp_collective"] == "p2p"
    assert d["roofline"]["frac"] > 0 and "cpu_baseline" not in d

