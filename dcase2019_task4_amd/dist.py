"""Data-parallel sharding of the mean-teacher step: one process per GPU, RCCL over xGMI.

The reference is single-process, so this is new capability; what it has to preserve
is the reference's batch contract.  main.py builds every batch as [weak | unlabeled | strong] with
sizes [B/4, B/2, B/4] (main.py:238-247) and addresses the labelled parts with positional slices
``weak_mask = slice(B/4)``, ``strong_mask = slice(3B/4, B)``.  Each rank therefore takes its share
OF EACH STREAM (not a contiguous chunk of the concatenated batch), so that

* the same positional masks stay valid on the local batch, and
* every mean-reduced loss term is a mean over equally many elements on every rank, hence the
  average of the per-rank gradients equals the gradient of the global-batch loss
  (BatchNorm statistics are per rank, as in standard DDP).

The only collective is ONE sum all-reduce of the flat fp32 gradient buffer per step (857 KB),
issued as two buckets - the GRU/heads tail (ready first) and the conv head - so the first overlaps
the conv-block backward.  ``torch.distributed`` backend "nccl" is RCCL on ROCm; everything here is
backend-agnostic and is covered on CPU with world_size-2 gloo tests.
"""
import os

import torch
import torch.distributed as dist


def local_batch_sizes(batch_sizes, world):
    """Per-rank stream sizes for global stream sizes (main.py:240 ``batch_sizes``)."""
    for b in batch_sizes:
        if b % world != 0:
            raise ValueError(f"every stream size must be divisible by world size {world}: {list(batch_sizes)}")
    return [b // world for b in batch_sizes]


def local_masks(batch_sizes, world):
    """(weak_mask, strong_mask) valid on a rank's local batch (main.py:241,247)."""
    lb = local_batch_sizes(batch_sizes, world)
    total = sum(lb)
    weak = slice(lb[0])
    strong = slice(total - lb[-1], total) if len(lb) == 3 else None
    return weak, strong


def shard_indices(batch_indices, batch_sizes, rank, world):
    """Rank's share of ONE global batch of dataset indices laid out stream after stream
    (what MultiStreamBatchSampler yields, DataLoad.py:562-571)."""
    lb = local_batch_sizes(batch_sizes, world)
    out, off = [], 0
    for b, l in zip(batch_sizes, lb):
        out.extend(batch_indices[off + rank * l: off + (rank + 1) * l])
        off += b
    return out


def shard_batch(tensors, batch_sizes, rank, world):
    """Rank's share of already-collated global tensors (each [B_global, ...])."""
    idx = shard_indices(list(range(sum(batch_sizes))), batch_sizes, rank, world)
    idx_t = torch.as_tensor(idx)
    return [t[idx_t.to(t.device)] for t in tensors]


def grad_buckets(layout):
    """Two contiguous buckets of the flat gradient buffer in the order backward produces them:
    (tail = GRU + heads, head = conv blocks).  ``layout``: [(start, end, shape)] per parameter in
    named_parameters() order; the first 18 tensors are the cnn (models/CRNN.py:12-31)."""
    cnn_end = layout[17][1]
    total = layout[-1][1]
    return (cnn_end, total), (0, cnn_end)


def allreduce_bucket(flat, lo, hi, group=None, async_op=False, force=False):
    """Sum all-reduce of flat[lo:hi] in place; returns the work handle when async_op.  A one-rank group is a no-op
    unless ``force`` (path test of the collective on a single-GPU box)."""
    if group is None and not dist.is_initialized():
        return None
    if dist.get_world_size(group) == 1 and not force:
        return None
    return dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def broadcast_parameters(flat_tensors, group=None, src=0):
    """Make every rank start from rank ``src``'s parameters / optimiser state."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in flat_tensors:
        dist.broadcast(t, src=src, group=group)


_M64 = 2 ** 64 - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def fold_rank_seed(seed, rank):
    """Philox base seed of data-parallel rank ``rank``.  Rank 0 keeps the user's seed (a one-process run and rank 0 of a
    data-parallel run draw the same stream); every other rank gets a HASH of (seed, rank).  The per-step keys are
    base + k * 0x9E37...15 + 1 (student k = 2 * step, teacher k = 2 * step + 1, csrc/optim.hip), so the rank must not
    enter as a multiple of that same constant: ``seed + rank * 0x9E37...15`` made rank 1's student masks equal rank 0's
    teacher masks of the same step and rank 2's stream rank 0's shifted by one step."""
    seed &= _M64
    if rank == 0:
        return seed
    return _splitmix64(seed ^ _splitmix64((rank + 0xD1B54A32D192ED03) & _M64))


class PeerAllReduce:
    """The gradient all-reduce as ONE kernel over peer-mapped device memory (csrc/p2p.hip, sed_p2p_*): every rank of the node
    maps every other rank's communication buffer (hipIpcMemHandle) and a single launch does reduce-scatter + all-gather with
    direct remote loads / stores, the sums formed in rank order on every rank.  No library collective inside the step: the
    launch is captured into the step's hipGraph like any other kernel, and - unlike RCCL - two ranks may share one GPU, so the
    captured data-parallel schedule is testable at world 2 on a one-GPU box.

    ``group`` is only used at construction (handle exchange, the self-check) and by ``errors(reduce=True)``.
    ``PeerAllReduce.create`` returns None instead of raising when the ranks are not on one host, a peer cannot be mapped, or
    the self-check against ``dist.all_reduce`` fails on ANY rank - the caller then keeps the process group's collective.

    Skew and failure.  Every cross-rank wait of the kernel has a wall-clock budget (``timeout_s``; default SED_P2P_TIMEOUT_S
    or 600 s - a rank that validates, writes a checkpoint or captures its graph while the others enter the next step is late,
    not dead, and RCCL would wait for ever).  A wait that does run out raises a sticky counter in the buffer and a word in
    pinned host memory (``poll()``: free, called on every step by MeanTeacherStep.run) and fills the launch's output with NaN,
    on this rank and in the slice it broadcasts: a timed-out all-reduce can neither hang the GPU nor pass for a result."""

    def __init__(self, n_floats_max, device, group=None, workgroups=0, fine_grained=True, timeout_s=None):
        import weakref
        from . import _lib
        self.l = _lib.lib()
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(device)
        self.n_max = int(n_floats_max)
        self.workgroups = int(workgroups) or int(os.environ.get("SED_P2P_WGS", "0"))
        self._own = None
        self._peers = {}
        self._host_err = None
        # what close() needs lives in a dict the finalizer can hold without keeping `self` alive: a step that is dropped without
        # close() (bench.py's legs) must not leak 16 x n bytes of device memory plus its IPC mappings
        self._res = {"own": None, "peers": {}, "host_err": None}
        self._finalizer = weakref.finalize(self, PeerAllReduce._release, self.l, self._res, self.device)
        if self.world > 16:
            raise _lib.SedError("PeerAllReduce: at most 16 ranks (one node)")
        try:
            self._build(fine_grained, timeout_s)
        except BaseException:
            self.close()                # the buffer and whatever peers were already mapped go back (sed_p2p_open may raise mid-loop)
            raise

    def _build(self, fine_grained, timeout_s):
        import ctypes as C
        import socket
        from . import _lib
        group = self.group
        with torch.cuda.device(self.device):
            nbytes = self.l.sed_p2p_buffer_bytes(self.n_max)
            own, fg = C.c_void_p(), C.c_int(0)
            handle = (C.c_ubyte * 64)()
            # (no rank may skip a collective of this constructor: an allocation that fails HERE still reaches the exchange)
            mine, err = None, None
            try:
                _lib.check(self.l.sed_p2p_alloc(nbytes, 1 if fine_grained else 0, C.byref(own), handle, C.byref(fg)), "sed_p2p_alloc")
                self._own = self._res["own"] = own.value
                # a word in pinned host memory that a timed-out wait raises besides the sticky counter in the buffer: poll()
                # reads it every step without touching the device
                self._host_err = self._res["host_err"] = torch.zeros(1, dtype=torch.int32).pin_memory()
                _lib.check(self.l.sed_p2p_configure(C.c_void_p(self._own), float(timeout_s or 0.0),
                                                    C.c_void_p(self._host_err.data_ptr()), 0), "sed_p2p_configure")
                self.fine_grained = bool(fg.value)
                mine = (socket.gethostname(), self.device.index or 0, os.getpid(), bytes(handle), self.fine_grained)
            except Exception as e:                     # noqa: BLE001
                err = e
            every = [None] * self.world
            dist.all_gather_object(every, mine, group=group)
            if any(v is None for v in every):
                raise _lib.SedError(f"PeerAllReduce: a rank could not create its communication buffer ({err!r})")
            if len({h for h, *_ in every}) != 1:
                raise _lib.SedError("PeerAllReduce: the ranks are not on one host")
            ptrs = (C.c_void_p * self.world)()
            for p, (_h, pdev, _pid, ph, _fg) in enumerate(every):
                if p == self.rank:
                    ptrs[p] = self._own
                    continue
                if not self.l.sed_p2p_can_access(int(pdev)):
                    raise _lib.SedError(f"PeerAllReduce: device {self.device.index} cannot map device {pdev}")
                q = C.c_void_p()
                hb = (C.c_ubyte * 64).from_buffer_copy(ph)
                _lib.check(self.l.sed_p2p_open(hb, C.byref(q)), "sed_p2p_open")
                self._peers[p] = self._res["peers"][p] = q.value
                ptrs[p] = q.value
            self._ptrs = ptrs
            self.layout = [(h, d, pid, fg_) for h, d, pid, _, fg_ in every]
            # Workgroups per launch: the library's default grows with the buffers' capacity (one per 2 K floats, 32 .. 128: the
            # phases are round trips to uncached memory, more lanes share them).  Same-index workgroups of different ranks spin
            # on each other, so ranks that SHARE a GPU (the one-GPU test set-up; never a production layout) stay at 32: more
            # spinning workgroups per process than that did not all become resident beside both ranks' persistent kernels.
            if not self.workgroups and len({d for _h, d, *_ in every}) < self.world:
                self.workgroups = 32
        # (no barrier needed: a rank's kernels only touch buffers IT has mapped, and every buffer was allocated and zeroed
        # before its handle was exchanged)

    @classmethod
    def create(cls, n_floats_max, device, group=None, verify=True, **kw):
        """Build + self-check on every rank; None (on every rank alike) if anything fails anywhere."""
        obj, ok = None, 1.0
        try:
            obj = cls(n_floats_max, device, group, **kw)
        except Exception as e:                         # noqa: BLE001 - any failure means "keep the process group's collective"
            ok = 0.0
            cls.last_error = repr(e)
        flag = torch.tensor([ok], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if flag.item() != 1.0:
            if obj is not None:
                obj.close()
            return None
        if verify:
            # the self-check runs under a SHORT wait budget: if flags or payload do not become visible between these devices
            # the construction must fall back to the process group's collective within seconds, not after the production
            # budget (600 s per wait); the configured budget is restored once the check has passed
            budget = float(kw.get("timeout_s") or os.environ.get("SED_P2P_TIMEOUT_S") or 600.0)
            obj.set_timeout(float(os.environ.get("SED_P2P_CHECK_TIMEOUT_S", "5")))
            if not obj.self_check():
                obj.close()
                return None
            obj.set_timeout(budget)
        return obj

    def all_reduce(self, flat, lo=0, hi=None, stream=None):
        """In-place sum of flat[lo:hi] (fp32, contiguous, 16-byte aligned start) over the ranks, on ``stream`` (default: the
        current stream).  Every rank must issue the same sequence of calls."""
        from . import _lib
        hi = flat.numel() if hi is None else hi
        if flat.dtype != torch.float32 or not flat.is_contiguous() or flat.device != self.device:
            raise _lib.SedError("PeerAllReduce.all_reduce: needs a contiguous fp32 tensor on the communicator's device")
        import ctypes as C
        ptr = flat.data_ptr() + 4 * lo
        if ptr % 16:
            raise _lib.SedError("PeerAllReduce.all_reduce: the message must start on a 16-byte boundary (callers choose "
                                "bucket bounds with PeerAllReduce.aligned(), or keep the process group's collective)")
        st = C.c_void_p(stream.cuda_stream) if stream is not None else _lib.stream_ptr()
        _lib.check(self.l.sed_p2p_allreduce(C.c_void_p(ptr), hi - lo, self.rank, self.world, self._ptrs, self.n_max,
                                            self.workgroups, st), "sed_p2p_allreduce")

    @staticmethod
    def aligned(flat, bounds):
        """True if every (lo, hi) of ``bounds`` starts a 16-byte-aligned message inside ``flat`` (what all_reduce needs)."""
        return all((flat.data_ptr() + 4 * lo) % 16 == 0 for lo, _hi in bounds)

    def poll(self):
        """Timed-out waits so far, read from the pinned host word the kernel raises: no device synchronisation, so the step
        driver calls it on EVERY step (a timed-out all-reduce has already filled the gradients with NaN; this turns it into an
        exception at the next call instead of a training run that carries on)."""
        return int(self._host_err[0]) if self._host_err is not None else 0

    def set_timeout(self, seconds):
        """New wait budget for every later launch on this rank's buffer (blocking; the default is SED_P2P_TIMEOUT_S or 600 s)."""
        import ctypes as C
        from . import _lib
        with torch.cuda.device(self.device):
            _lib.check(self.l.sed_p2p_configure(C.c_void_p(self._own), float(seconds), None, 0), "sed_p2p_configure")

    def errors(self, reduce=False):
        """Sticky count of cross-rank waits that ran out of their budget: non-zero = results since then are NaN-poisoned."""
        import ctypes as C
        from . import _lib
        n = C.c_uint(0)
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            _lib.check(self.l.sed_p2p_errors(C.c_void_p(self._own), C.byref(n)), "sed_p2p_errors")
        v = int(n.value)
        if reduce and self.world > 1:
            t = torch.tensor([float(v)], device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            v = int(t.item())
        return v

    def self_check(self):
        """A few all-reduces of different sizes (full buffer, an odd size, a tiny one; twice each so that both halves of the
        double-buffered staging are used), eager and replayed from a hipGraph, against dist.all_reduce; True only if every
        rank got bit-for-bit the rank-ordered sum and nothing timed out."""
        ok = 1.0
        try:
            sizes = sorted({self.n_max, max(1, self.n_max // 3) | 1, 5})
            g = torch.Generator(device="cpu").manual_seed(1234 + self.rank)
            for n in sizes:
                for rep in range(2):
                    x = torch.randn(n, generator=g).to(self.device)
                    parts = [torch.zeros_like(x) for _ in range(self.world)]
                    dist.all_gather(parts, x, group=self.group)
                    want = parts[0].clone()
                    for p in parts[1:]:
                        want += p                      # rank order, as the kernel sums
                    y = x.clone()
                    self.all_reduce(y)
                    torch.cuda.synchronize(self.device)
                    if not torch.equal(y, want):
                        ok = 0.0
                    # a wait ran out somewhere: no point in paying for the remaining calls' budgets - but every rank must leave
                    # the loop TOGETHER (each turn holds an all_gather), so the decision is itself a collective
                    bad = torch.tensor([1.0 if self.poll() else 0.0], device=self.device)
                    dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.group)
                    if bad.item() != 0.0:
                        raise RuntimeError("a cross-rank wait timed out during the self-check")
            # captured: the launch must replay correctly (device-side epochs)
            x = torch.randn(sizes[-1], generator=g).to(self.device)
            parts = [torch.zeros_like(x) for _ in range(self.world)]
            dist.all_gather(parts, x, group=self.group)
            want = parts[0].clone()
            for p in parts[1:]:
                want += p
            y = x.clone()
            st = torch.cuda.Stream(device=self.device)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st, capture_error_mode="thread_local"):
                self.all_reduce(y)
            for _ in range(2):
                y.copy_(x)
                gr.replay()
                torch.cuda.synchronize(self.device)
                if not torch.equal(y, want):
                    ok = 0.0
            if self.errors() != 0:
                ok = 0.0
        except Exception as e:                         # noqa: BLE001
            ok = 0.0
            PeerAllReduce.last_error = repr(e)
        flag = torch.tensor([ok], device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(flag.item() == 1.0)

    def time_us(self, n, iters=50, stream=None):
        """Mean microseconds of one captured all-reduce of n floats over `iters` hipGraph replays (every rank calls this with
        the same arguments; the MAX over the ranks is what a step would see)."""
        return graph_time_us(lambda t: self.all_reduce(t), n, self.device, self.group, iters, stream)

    @staticmethod
    def _release(l, res, device):
        import ctypes as C
        try:
            if res["own"] is None and not res["peers"]:
                return
            with torch.cuda.device(device):
                torch.cuda.synchronize(device)
                for q in res["peers"].values():
                    l.sed_p2p_close(C.c_void_p(q))
                res["peers"].clear()
                if res["own"] is not None:
                    l.sed_p2p_free(C.c_void_p(res["own"]))
                    res["own"] = None
        except Exception:                              # noqa: BLE001 - interpreter shutdown: the driver reclaims the memory
            pass

    def close(self):
        PeerAllReduce._release(self.l, self._res, self.device)
        self._peers = {}
        self._own = None

    last_error = None


def graph_time_us(fn, n, device, group=None, iters=50, stream=None):
    """Capture ``fn(t)`` - one in-place all-reduce of an n-float tensor - into a hipGraph, replay it `iters` times behind
    two warm-up replays and return the mean microseconds per call, MAX-reduced over the ranks (a collective is as slow
    as its slowest participant).  Raises whatever the capture raises (the caller treats that as "not available")."""
    t = torch.zeros(int(n), device=device, dtype=torch.float32)
    st = stream or torch.cuda.Stream(device=device)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st, capture_error_mode="thread_local"):
        fn(t)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        for _ in range(2):
            gr.replay()
        st.synchronize()
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.barrier(group=group)
        e0.record(st)
        for _ in range(iters):
            gr.replay()
        e1.record(st)
        st.synchronize()
    us = torch.tensor([e0.elapsed_time(e1) * 1e3 / iters], device=device, dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(us, op=dist.ReduceOp.MAX, group=group)
    return float(us.item())


def choose_collective(p2p, flat, buckets, group, device, iters=50):
    """collective="auto": time the library's peer all-reduce against the process group's (captured) all-reduce on the two
    gradient buckets the step will send and keep the FASTER (summed over the buckets; every rank sees the same MAX-reduced
    times, hence takes the same decision).  Returns (choice, record): record holds both sets of times for config.dp_collective.
    The process group is only a candidate where its collective can be captured and does something (backend nccl = RCCL,
    world > 1); with gloo, or at one rank, the peer kernel is the only capturable one and is kept if it passed its self-check."""
    rec = {"p2p_us": None, "pg_us": None, "buckets_bytes": [4 * (hi - lo) for lo, hi in buckets], "iters": iters}
    if p2p is None:
        rec["why"] = "peer all-reduce not available: " + str(PeerAllReduce.last_error)
        return "pg", rec
    world = dist.get_world_size(group)
    try:
        rec["p2p_us"] = [round(p2p.time_us(hi - lo, iters), 2) for lo, hi in buckets]
    except Exception as e:                             # noqa: BLE001
        rec["why"] = "timing the peer all-reduce failed: " + repr(e)[:200]
    pg_ok = dist.get_backend(group) == "nccl" and world > 1
    if pg_ok:
        try:
            rec["pg_us"] = [round(graph_time_us(lambda t: dist.all_reduce(t, group=group), hi - lo, device, group, iters), 2)
                            for lo, hi in buckets]
        except Exception as e:                         # noqa: BLE001
            rec["pg_capture_error"] = repr(e)[:200]
    # every rank must agree: the times are MAX-reduced already, the exceptions are not - settle them with one MIN
    have = torch.tensor([1.0 if rec["p2p_us"] else 0.0, 1.0 if rec["pg_us"] else 0.0], device=device)
    if world > 1:
        dist.all_reduce(have, op=dist.ReduceOp.MIN, group=group)
    have_p2p, have_pg = bool(have[0].item()), bool(have[1].item())
    if have_p2p and have_pg:
        choice = "p2p" if sum(rec["p2p_us"]) <= sum(rec["pg_us"]) else "pg"
        rec["why"] = "faster on the step's two buckets"
    elif have_p2p:
        choice = "p2p"
        rec.setdefault("why", "the process group's collective cannot be captured here (backend %s, world %d)" % (dist.get_backend(group), world))
    else:
        choice = "pg"
    rec["choice"] = choice
    return choice, rec

