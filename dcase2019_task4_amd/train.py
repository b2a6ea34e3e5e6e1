"""Mean-teacher train step on MI355X - host-side mirror of baseline/main.py:45-165.

Two ways in:

* drop-in: keep baseline/main.py's own ``train`` loop and ``torch.optim.Adam`` and just build the
  models with ``dcase2019_task4_amd.crnn.CRNN`` - forward/backward then run in the HIP kernels
  through one autograd node.  ``update_ema_variables`` below is the same call as main.py:45-49
  fused into one kernel over the flat parameter buffers.
* fused: ``MeanTeacherStep`` runs the whole body of the loop (main.py:84-157: teacher forward,
  student forward, losses, backward, Adam, EMA, step counter) as a fixed sequence of C-ABI calls
  on persistent buffers, optionally captured into one hipGraph, with every per-step scalar
  (consistency weight, EMA alpha, Adam bias correction, dropout seeds) advanced ON DEVICE - none
  of the ~11 host syncs per step of the reference loop (main.py:106-149 ``.item()``).  ``train``
  wraps it in the reference's epoch-loop signature.
"""
import ctypes as C
import os
import time

import numpy as np
import torch

from . import _lib
from . import dist as sdist
from .crnn import CRNN

LOSS_NAMES = ("loss", "weak_class_loss", "strong_loss", "cons_strong", "cons_weak", "weak_ema_loss",
              "strong_ema_loss", "cons_weight")


def update_ema_variables(model, ema_model, alpha, global_step):
    """main.py:45-49 - ``alpha = min(1 - 1/(global_step+1), alpha)``; parameters only (buffers are
    not averaged).  One fused kernel over the two flat buffers."""
    alpha = min(1 - 1 / (global_step + 1), alpha)
    model.flatten_parameters_()
    ema_model.flatten_parameters_()
    n = model._flat.numel()
    _lib.check(_lib.lib().sed_ema_update(n, _lib.ptr(model._flat), _lib.ptr(ema_model._flat), float(alpha),
                                         _lib.stream_ptr()), "sed_ema_update")


_STREAM_POOL = {}


def shared_stream(device, role, priority=0):
    """One torch stream per (device, role, priority) for the whole process.  Every stream a step hands to the library also gets
    two helper streams inside it (sed_stream_prepare); until round 5 nothing ever destroyed those, and a process that built
    several steps one after the other (bench.py's legs, a test session) ended up with dozens of live streams on four hardware
    queues - the graph branches of the later steps shared queues with them (measured: 0.90 instead of 0.66 ms for the fifth
    step built in a process).  The library now has sed_stream_release, and a step built with ``pool_streams=False`` owns its
    streams and releases them in ``close()``; the pool stays the default because steps of one process run one at a time and
    torch itself hands out streams from a fixed pool of 32 per device and priority."""
    key = (torch.device(device).index or 0, role, int(priority))
    st = _STREAM_POOL.get(key)
    if st is None:
        st = _STREAM_POOL[key] = torch.cuda.Stream(device=device, priority=int(priority))
    return st


def _slice_range(s, B):
    if s is None:
        return 0, 0
    lo, hi, st = s.indices(B)
    if st != 1:
        raise NotImplementedError("masks must be contiguous slices (main.py:241,247)")
    return lo, max(lo, hi)


class MeanTeacherStep:
    """The body of main.train's loop (main.py:84-157) as one fused, graph-capturable device step.

    ``teacher=None`` gives the supervised loop body of main_simple_CRNN.train (main_simple_CRNN.py:39-75):
    student forward, weak + strong BCE (either mask may be None), backward, Adam - no teacher forward, no
    consistency terms, no EMA."""

    def __init__(self, student, teacher, batch_size, n_frames, rampup_length, weak_mask, strong_mask, lr=1e-3,
                 betas=(0.9, 0.999), eps=1e-8, ema_decay=0.999, max_consistency_cost=2.0, seed=0, use_graph=True,
                 process_group=None, overlap_streams=True, dp_schedule=None, pool_streams=True, collective=None):
        assert isinstance(student, CRNN) and (teacher is None or isinstance(teacher, CRNN))
        if not student.hot_path or (teacher is not None and not teacher.hot_path):
            raise _lib.SedError("MeanTeacherStep runs the HIP hot path only: this CRNN configuration is served by stock torch "
                                "operators (CRNN.hot_path is False) - use the reference's own loop for it")
        self.l = _lib.lib()
        self.student, self.teacher = student, teacher
        self.supervised = teacher is None
        dev = next(student.parameters()).device
        if dev.type != "cuda":
            raise _lib.SedError("MeanTeacherStep needs the models on the GPU (no CPU fallback)")
        self.device = dev
        student.flatten_parameters_(dev)
        if teacher is not None:
            teacher.flatten_parameters_(dev)
        self.B, self.T = int(batch_size), int(n_frames)
        if teacher is not None and (teacher._C, teacher._H, teacher._dtype) != (student._C, student._H, student._dtype):
            raise ValueError("student and teacher must have the same geometry and mfma_dtype")
        self.dims = student.make_dims(self.B, self.T)
        self._pool_streams = bool(pool_streams)
        self._owned_streams = []
        self.T3, self.NC = self.T // 8, student._nclass
        # the student's output heads are left to sed_mt_step_backward where the library runs them inside the backward recurrence
        # (csrc/kernels.h heads_fusable; SED_DEFER_HEADS=0/1 overrides for A/B timing)
        self._defer_heads = (student._H == 64 and self.T3 <= 128)
        if os.environ.get("SED_DEFER_HEADS") in ("0", "1"):
            self._defer_heads = os.environ["SED_DEFER_HEADS"] == "1"
        self.wlo, self.whi = _slice_range(weak_mask, self.B) if weak_mask is not None else (0, 0)
        self.slo, self.shi = _slice_range(strong_mask, self.B) if strong_mask is not None else (0, 0)
        n = student._flat.numel()
        self.n = n
        f32 = dict(device=dev, dtype=torch.float32)
        self.grads = torch.zeros(n, **f32)
        self.exp_avg = torch.zeros(n, **f32)
        self.exp_avg_sq = torch.zeros(n, **f32)
        self.state = torch.zeros(C.sizeof(_lib.SedStepState), device=dev, dtype=torch.uint8)
        self.steps_done = 0
        self._gs_offset = 0                # global_step_host = steps_done + _gs_offset: host mirror of state.global_step
        self.pg = process_group
        self.world, self.rank = 1, 0
        if process_group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
            self.rank = dist.get_rank(process_group)
        # every replica draws its OWN dropout masks / teacher noise: the rank is folded into the Philox base seed here
        # (rank 0 keeps the user's seed, so a one-process run and rank 0 of a data-parallel run see the same stream)
        self.seed_user = int(seed) & (2 ** 64 - 1)
        _lib.check(self.l.sed_step_state_init(_lib.ptr(self.state), self._folded_seed(), int(rampup_length),
                                              float(lr), float(betas[0]), float(betas[1]), float(eps),
                                              float(ema_decay), float(max_consistency_cost), _lib.stream_ptr()),
                   "sed_step_state_init")
        self.lr = float(lr)
        base = self.state.data_ptr()
        self._seed_s = C.c_void_p(base + _lib.SedStepState.seed_student.offset)
        self._seed_t = C.c_void_p(base + _lib.SedStepState.seed_teacher.offset)
        self.ctx_bytes = self.l.sed_crnn_ctx_bytes(C.byref(self.dims))
        self.ws_bytes = self.l.sed_crnn_bwd_ws_bytes(C.byref(self.dims))
        if self.ctx_bytes == 0 or self.ws_bytes == 0:
            raise _lib.SedError(self.l.sed_last_error().decode())
        # the library initialises what it needs initialised in fresh buffers (the wide model's cluster recurrence keeps
        # launch epochs, tagged exchange granules and its sticky timeout counter in there): sed_crnn_buffers_init
        self.ctx_s = _lib.scratch(self.ctx_bytes, dev)
        self.ctx_t = _lib.scratch(self.ctx_bytes if teacher is not None else 0, dev)
        self.ws = _lib.scratch(self.ws_bytes, dev)
        _lib.check(self.l.sed_crnn_buffers_init(C.byref(self.dims), _lib.ptr(self.ctx_s), self.ctx_bytes, _lib.ptr(self.ws),
                                                self.ws_bytes, _lib.stream_ptr()), "sed_crnn_buffers_init")
        if teacher is not None:
            _lib.check(self.l.sed_crnn_buffers_init(C.byref(self.dims), _lib.ptr(self.ctx_t), self.ctx_bytes, None, 0,
                                                    _lib.stream_ptr()), "sed_crnn_buffers_init")
        self._err_view = None
        if student._H == 256:
            off, nb = C.c_size_t(), C.c_size_t()
            _lib.check(self.l.sed_crnn_ctx_view(C.byref(self.dims), b"gru_err", C.byref(off), C.byref(nb)), "sed_crnn_ctx_view")
            self._err_view = (off.value, nb.value)
        self.x = torch.zeros(self.B, 1, self.T, 64, **f32)
        self.x_ema = torch.zeros(self.B, 1, self.T, 64, **f32)
        self.target = torch.zeros(self.B, self.T3, self.NC, **f32)
        self.strong = torch.empty(self.B, self.T3, self.NC, **f32)
        self.weak = torch.empty(self.B, self.NC, **f32)
        # supervised: the "teacher" outputs ARE the student's, so both consistency terms and their gradients are
        # exactly zero and the loss kernel needs no second variant
        self.strong_ema = torch.empty(self.B, self.T3, self.NC, **f32) if teacher is not None else self.strong
        self.weak_ema = torch.empty(self.B, self.NC, **f32) if teacher is not None else self.weak
        self.d_strong = torch.empty(self.B, self.T3, self.NC, **f32)
        self.d_weak = torch.empty(self.B, self.NC, **f32)
        self.losses = torch.zeros(8 + 8 * self.B + 8, **f32)          # SED_LOSS_FLOATS(B): meters | scratch
        # Data-parallel schedule (SED_FORCE_DP=1 takes it with a one-rank group too, so that the whole RCCL path can be
        # exercised on a single-GPU box).
        #   "overlap" (default): forward + loss + backward of heads/GRU (parts 5) | on a second stream: the tail bucket's
        #       weight-gradient GEMMs (parts 8) then its all-reduce, OVERLAPPING | the conv-block backward (parts 2) on the
        #       main stream | all-reduce of the conv bucket | Adam + EMA.  The GRU weight-gradient GEMMs stay off the
        #       critical path (they used to be on it in round 1's "split" schedule: +70 us), and 60 % of the gradient
        #       bytes cross xGMI while the conv backward (the longest part of the backward) is still computing.
        #   "single": whole backward (parts 3) | ONE all-reduce of the flat gradient buffer | update.  Nothing overlaps.
        #   SED_DP_CAPTURE=1 (RCCL only): the collectives are captured INTO the hipGraph, the whole step is one replay.
        self.dp = process_group is not None and (self.world > 1 or os.environ.get("SED_FORCE_DP") == "1")
        # Default: the overlap schedule with the collectives CAPTURED into the step's hipGraph when the backend is RCCL and a
        # trial capture of an all-reduce replays correctly on every rank (one replay per step; measured with a one-rank
        # RCCL group: 0.836 ms against 0.816 ms without data parallelism, 0.839 ms for "single", and 0.969 ms for the overlap
        # schedule with EAGER collectives between four graph segments - that one is host-bound: 4 graph launches + 2 async
        # collectives + stream waits per step).  Otherwise (gloo, or capture not available): "single" with an eager collective.
        self._cap_stream = self._stream("capture")
        # The gradient all-reduce itself: "p2p" = ONE kernel over peer-mapped device memory (dist.PeerAllReduce, csrc/p2p.hip:
        # reduce-scatter + all-gather with direct loads / stores between the ranks of a node, sums in rank order), "pg" = the
        # process group's all_reduce (RCCL, gloo), "auto" (default) = p2p when every rank can map every other one AND a
        # self-check against the process group's result passes on every rank, else pg.  The p2p launch is an ordinary kernel:
        # it is always capturable, and two ranks may share one GPU (RCCL refuses that), which is how the captured schedule is
        # tested at world 2 on a one-GPU box.
        # "auto" no longer stops at "p2p works": it TIMES both on the step's two buckets (50 captured replays each, MAX over the
        # ranks) and keeps the faster - dist.choose_collective; both sets of times go into self.collective_record (bench.py
        # prints them as config.dp_collective_record).
        want_coll = collective or os.environ.get("SED_DP_COLLECTIVE") or "auto"
        if want_coll not in ("auto", "p2p", "pg"):
            raise ValueError(f"unknown collective {want_coll!r}")
        self._buckets = sdist.grad_buckets(student._layout)
        self._p2p = None
        self.collective_record = None
        if self.dp and want_coll != "pg":
            if sdist.PeerAllReduce.aligned(self.grads, self._buckets + ((0, n),)):
                self._p2p = sdist.PeerAllReduce.create(n, dev, process_group)
            else:      # (never with the reference's parameter shapes: every tensor is a multiple of 4 floats)
                sdist.PeerAllReduce.last_error = "a gradient bucket does not start on a 16-byte boundary"
            if self._p2p is None and want_coll == "p2p":
                raise _lib.SedError(f"collective='p2p' is not available here: {sdist.PeerAllReduce.last_error}")
            if want_coll == "auto":
                choice, self.collective_record = sdist.choose_collective(self._p2p, self.grads, self._buckets, process_group, dev)
                if choice == "pg" and self._p2p is not None:
                    self._p2p.close()
                    self._p2p = None
        self.collective = "p2p" if self._p2p is not None else ("pg" if self.dp else None)
        env_cap = os.environ.get("SED_DP_CAPTURE")
        want = dp_schedule or os.environ.get("SED_DP_SCHEDULE")
        if want == "split":
            want = "overlap"
        if want not in (None, "overlap", "single"):
            raise ValueError(f"unknown data-parallel schedule {want!r}")
        self.dp_capture = False
        if self.dp and use_graph and env_cap != "0" and (want != "single" or env_cap == "1"):
            self.dp_capture = (env_cap == "1") or self._p2p is not None or self._collective_capture_works()
        self.dp_schedule = want or ("overlap" if (self.dp_capture or not self.dp) else "single")
        self._dp_stream = self._stream("collective") if self.dp else None
        # train_cnn=False (CRNN.py:18-20, main.py:289-290 filters the optimiser's parameters on requires_grad): the conv
        # blocks' backward is skipped and their gradient stays zero, which makes Adam's update of those entries exactly
        # zero (zero moments, no weight decay) while the EMA still covers every parameter (main.py:45-49 zips ALL of them)
        frozen = [not p.requires_grad for p in student.parameters()]
        self.cnn_frozen = all(frozen[:18]) and not any(frozen[18:])
        if any(frozen) and not self.cnn_frozen:
            raise NotImplementedError("MeanTeacherStep supports all parameters trainable, or the whole CNN frozen "
                                      "(train_cnn=False); other requires_grad patterns are not implemented")
        if process_group is not None:
            self.sync_replicas()
        self.use_graph = bool(use_graph)
        self.overlap = bool(overlap_streams)
        self._side = (self._stream("teacher", int(os.environ.get("SED_SIDE_PRIO", "0")))
                      if (self.overlap and teacher is not None) else None)
        self._capture_error = None
        self._graph_a = None
        self._graph_w = None
        self._graph_c = None
        self._graph_b = None
        # graphs are captured on a stream of our own whose library helper stream exists BEFORE the capture starts
        for st in (torch.cuda.current_stream(dev), self._cap_stream, self._side, self._dp_stream):
            if st is not None:
                _lib.check(self.l.sed_stream_prepare(C.c_void_p(st.cuda_stream)), "sed_stream_prepare")
        self._warm = 0
        # Patch moments one step ahead (sed_crnn_moments, csrc/blk0.hip): block 0's train-mode BatchNorm statistics come from the
        # 9 + 45 moments of the 3x3 input patch, a function of the BATCH only.  When the batch in self.x / self.x_ema is still the
        # one the previous run() trained on (a resident batch: bench.py; anything that calls run() without load_batch() in
        # between), the previous step has already computed them - on a third stream beside its heads / BiGRU backward, which
        # leave most of the chip idle - and this step's forwards start at k_blk0_prep: 16 us (B = 24) off the head of the
        # critical chain, every step still computing one pair of moments.  step() / load_batch() + run() never speculates:
        # self.x may ONLY change through load_batch() (or be followed by invalidate_batch()).
        self.moments_ahead = os.environ.get("SED_MOMENTS_AHEAD", "0") == "1"
        self._mom_external = False         # features.WaveformFrontEnd: every batch arrives with its moments already in ctx
        self._mom_valid = False            # ctx_s / ctx_t hold the moments of what is in self.x / self.x_ema right now
        self._resident = False            # a run() has consumed the batch in self.x and no load_batch() came since
        self._mom_ready = self._mom_next = False      # the variant being run / captured
        self._mom_stream = self._stream("moments", int(os.environ.get("SED_MOM_PRIO", "0")))
        self._mom_fork = os.environ.get("SED_MOM_FORK", "backward")
        self._graph_sets = {}              # (mom_ready, mom_next) -> (graph_a, graph_w, graph_c, graph_b)

    @property
    def global_step_host(self):
        return self.steps_done + self._gs_offset

    def _stream(self, role, priority=0):
        if self._pool_streams:
            return shared_stream(self.device, role, priority)
        st = torch.cuda.Stream(device=self.device, priority=int(priority))
        self._owned_streams.append(st)
        return st

    def close(self):
        """Drops the graphs and releases what the library created for the streams this step OWNS (``pool_streams=False``):
        sed_stream_release destroys the two helper streams + events per stream.  Pooled streams are shared with every other
        step of the process and stay prepared.  The step must not be used afterwards."""
        torch.cuda.synchronize(self.device)
        self._graph_a = self._graph_w = self._graph_c = self._graph_b = None
        self._graph_sets = {}
        if self._p2p is not None:
            self._p2p.close()
            self._p2p = None
        for st in self._owned_streams:
            _lib.check(self.l.sed_stream_release(C.c_void_p(st.cuda_stream)), "sed_stream_release")
        self._owned_streams = []

    # ---- pieces ------------------------------------------------------------------------------------
    def _forward(self, model, x, ctx, seed, strong, weak):
        """strong = weak = None: output heads deferred to sed_mt_step_backward (the student's forward).  The teacher's forward is
        train-mode but never differentiated (main.py:87-89): train = 3 tells the library so."""
        train = 3 if (model is self.teacher and model is not None) else 1
        if self._mom_ready:
            train |= 4               # this batch's patch moments are already in ctx (the previous step's _moments_ahead)
        _lib.check(self.l.sed_crnn_forward(C.byref(self.dims), _lib.ptr(model._flat), _lib.ptr(model._bn_flat),
                                           _lib.ptr(model._bn_tracked), _lib.ptr(x), train, 1, seed, _lib.ptr(ctx),
                                           self.ctx_bytes, _lib.ptr(strong), _lib.ptr(weak), _lib.stream_ptr()),
                   "sed_crnn_forward")

    def _fwd_bwd(self, after_forward=None, at_recurrence=None):
        self._fork_exc = None
        if self._mom_next and self._mom_fork == "gru" and at_recurrence is None:
            at_recurrence = self._moments_ahead
        try:
            self._fwd_bwd_impl(after_forward, at_recurrence)
        finally:
            if at_recurrence is not None:      # one-shot hook: never leave it registered (e.g. a forward that failed before it)
                self.l.sed_crnn_fork_callback(_lib.stream_ptr(), None, None)
        if self._fork_exc is not None:
            # ctypes swallows (prints) an exception raised inside a C callback: a failing hook - the next batch's feature
            # extraction - would otherwise leave a captured graph without those kernels and the step training on stale slots
            exc, self._fork_exc = self._fork_exc, None
            raise exc

    def _student_forward(self, at_recurrence):
        """The student forward (main.py:91).  The one-shot `at_recurrence` hook is registered right in front of THIS call: the
        library hands it to the next sed_crnn_forward on the stream, which without a side stream would be the teacher's."""
        if at_recurrence is not None:
            def hook(_user):
                try:
                    at_recurrence()
                except BaseException as e:      # noqa: BLE001 - re-raised by _fwd_bwd once the C call has returned
                    self._fork_exc = e
            self._fork_cb = _lib.FORK_CALLBACK(hook)      # (kept alive until the forward has run)
            _lib.check(self.l.sed_crnn_fork_callback(_lib.stream_ptr(), self._fork_cb, None), "sed_crnn_fork_callback")
        # the student's output heads are deferred where the library fuses them (n_RNN_cell = 64, T / 8 <= 128):
        # sed_mt_step_backward runs them with the loss and their backward inside the top layer's backward recurrence
        # (csrc/hfuse.h).  Other geometries (the 256-cell BiGRU) have no fused form: deferred, the student's k_heads_fwd would
        # queue up behind the teacher's AND the join (21 us of the wide step's critical chain); run by the forward it overlaps
        # the teacher's on the other stream.  Same kernels, same results.
        if self._defer_heads:
            self._forward(self.student, self.x, self.ctx_s, self._seed_s, None, None)
        else:
            self._forward(self.student, self.x, self.ctx_s, self._seed_s, self.strong, self.weak)

    def _fwd_bwd_impl(self, after_forward=None, at_recurrence=None):
        """teacher forward (main.py:87-89), student forward (:91), losses (:93-145), backward (:152-153).
        `after_forward`: called between the forwards and the backward; `at_recurrence`: called from INSIDE the student
        forward, between its conv stack and its recurrence (sed_crnn_fork_callback) - where the waveform front-end forks
        the next batch's feature kernels (features.WaveformFrontEnd)."""
        if self.supervised:
            self._student_forward(at_recurrence)
        elif self._side is not None:
            # the teacher forward is independent of the student forward: run it on a second stream
            cur = torch.cuda.current_stream()
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                self._forward(self.teacher, self.x_ema, self.ctx_t, self._seed_t, self.strong_ema, self.weak_ema)
            self._student_forward(at_recurrence)
            cur.wait_stream(self._side)
        else:
            self._forward(self.teacher, self.x_ema, self.ctx_t, self._seed_t, self.strong_ema, self.weak_ema)
            self._student_forward(at_recurrence)
        if after_forward is not None:
            after_forward()
        if self._mom_next and self._mom_fork != "gru":
            self._moments_ahead()
        # losses (main.py:93-145) + backward in one call: the heads-backward kernel forms the loss gradient per clip itself
        # (sed_mt_loss as a kernel of its own was 12 us on the critical path).  One process: the whole backward
        # (parts = 3), which lets the library overlap the GRU weight gradients with the conv-block backward;
        # data-parallel "split": part 1 here, part 2 after its bucket's all-reduce has been started (run())
        if self.cnn_frozen:
            parts = 1
        elif self.dp and self.dp_schedule == "overlap":
            parts = 5
        else:
            parts = 3
        if self._defer_heads:
            _lib.check(self.l.sed_mt_step_backward(C.byref(self.dims), _lib.ptr(self.student._flat), _lib.ptr(self.x),
                                                   self._seed_s, _lib.ptr(self.ctx_s), self.ctx_bytes,
                                                   _lib.ptr(self.strong), _lib.ptr(self.weak),
                                                   _lib.ptr(self.strong_ema), _lib.ptr(self.weak_ema), _lib.ptr(self.target),
                                                   self.wlo, self.whi, self.slo, self.shi, _lib.ptr(self.state), 1,
                                                   _lib.ptr(self.losses), None, None, _lib.ptr(self.grads),
                                                   _lib.ptr(self.ws), self.ws_bytes, parts, _lib.stream_ptr()),
                       "sed_mt_step_backward")
        else:
            _lib.check(self.l.sed_mt_loss_backward(C.byref(self.dims), _lib.ptr(self.student._flat), _lib.ptr(self.x),
                                                   self._seed_s, _lib.ptr(self.ctx_s), self.ctx_bytes,
                                                   _lib.ptr(self.strong_ema), _lib.ptr(self.weak_ema), _lib.ptr(self.target),
                                                   self.wlo, self.whi, self.slo, self.shi, _lib.ptr(self.state), 1,
                                                   _lib.ptr(self.losses), None, None, _lib.ptr(self.grads),
                                                   _lib.ptr(self.ws), self.ws_bytes, parts, _lib.stream_ptr()),
                       "sed_mt_loss_backward")

        if self._mom_next:
            torch.cuda.current_stream().wait_stream(self._mom_stream)

    def _moments_ahead(self, x=None, x_ema=None):
        """The patch moments of the batch the NEXT step trains on (default: the resident one in self.x / self.x_ema) into the
        models' ctx, on a stream of their own beside this step's backward."""
        cur = torch.cuda.current_stream()
        self._mom_stream.wait_stream(cur)
        with torch.cuda.stream(self._mom_stream):
            self._moments(self.x if x is None else x, self.ctx_s)
            if self.teacher is not None:
                self._moments(self.x_ema if x_ema is None else x_ema, self.ctx_t)

    def _moments(self, x, ctx):
        _lib.check(self.l.sed_crnn_moments(C.byref(self.dims), _lib.ptr(x), _lib.ptr(ctx), self.ctx_bytes, _lib.stream_ptr()),
                   "sed_crnn_moments")

    def invalidate_batch(self):
        """Tell the step that self.x / self.x_ema / self.target were written behind its back (anything but load_batch())."""
        self._mom_valid = False
        self._resident = False

    def _backward(self, parts):
        _lib.check(self.l.sed_crnn_backward(C.byref(self.dims), _lib.ptr(self.student._flat), _lib.ptr(self.x),
                                            self._seed_s, _lib.ptr(self.ctx_s), self.ctx_bytes,
                                            _lib.ptr(self.d_strong), _lib.ptr(self.d_weak), _lib.ptr(self.grads),
                                            _lib.ptr(self.ws), self.ws_bytes, parts, _lib.stream_ptr()), "sed_crnn_backward")

    def _update(self):
        """Adam (main.py:154) + EMA teacher (:155-157), one kernel."""
        _lib.check(self.l.sed_adam_ema(self.n, _lib.ptr(self.student._flat), _lib.ptr(self.grads),
                                       _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                                       _lib.ptr(self.teacher._flat) if self.teacher is not None else None,
                                       _lib.ptr(self.state), 1.0 / self.world,
                                       _lib.stream_ptr()), "sed_adam_ema")
        # (no sed_step_state_advance: the loss / heads-backward kernel of this step already moved the counters on and
        # left the update's own fields derived for this step - sed_mt_loss_backward(advance_state = 1))

    def _collective_capture_works(self):
        """Trial: capture one all-reduce into a hipGraph, replay it, compare with the eager result; every rank must succeed."""
        import torch.distributed as dist
        try:
            if dist.get_backend(self.pg) != "nccl":
                return False
            if self.world == 1:
                return True          # a one-rank all-reduce is a no-op: nothing to try
            t = torch.arange(1024, device=self.device, dtype=torch.float32) + self.rank
            ref = t.clone()
            dist.all_reduce(ref, group=self.pg)
            torch.cuda.synchronize(self.device)
            ok = 1.0
            try:
                g = torch.cuda.CUDAGraph()
                # thread_local: the process group's watchdog thread queries events of earlier collectives, which a
                # capture in the default "global" mode treats as an illegal call and aborts on
                with torch.cuda.graph(g, stream=self._cap_stream, capture_error_mode="thread_local"):
                    dist.all_reduce(t, group=self.pg)
                g.replay()
                torch.cuda.synchronize(self.device)
                ok = 1.0 if torch.equal(t, ref) else 0.0
            except Exception:
                ok = 0.0
            flag = torch.tensor([ok], device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
            return bool(flag.item() == 1.0)
        except Exception:
            return False

    def _folded_seed(self):
        return sdist.fold_rank_seed(self.seed_user, self.rank)

    def sync_replicas(self, src=0):
        """Every replica starts from rank ``src``'s parameters, BatchNorm buffers, Adam moments and step counters (the
        reference has one copy of each; DDP convention: rank 0 wins) - e.g. after rank 0 loaded a checkpoint - and
        then re-folds its own rank into the dropout seed."""
        ts = [self.student._flat, self.student._bn_flat, self.student._bn_tracked, self.exp_avg, self.exp_avg_sq, self.state]
        if self.teacher is not None:
            ts += [self.teacher._flat, self.teacher._bn_flat, self.teacher._bn_tracked]
        sdist.broadcast_parameters(ts, self.pg, src=src)
        _lib.check(self.l.sed_step_state_update(_lib.ptr(self.state), self._folded_seed(), 0.0, 1, _lib.stream_ptr()),
                   "sed_step_state_update")
        self._gs_offset = int(self.read_state().global_step) - self.steps_done

    def set_global_step(self, global_step):
        """main.py:74 - ``global_step = epoch * len(train_loader) + i``: the reference recomputes the counter from the epoch
        argument on every call of ``train``; this sets the device counter the ramp-up, the EMA alpha and the dropout keys
        derive from (Adam's own step count is the optimiser's and stays)."""
        _lib.check(self.l.sed_step_state_set_global_step(_lib.ptr(self.state), int(global_step), _lib.stream_ptr()),
                   "sed_step_state_set_global_step")
        self._gs_offset = int(global_step) - self.steps_done

    def set_lr(self, lr):
        """Honour an optimiser whose lr changed between epochs (main.py never does; utils.adjust_learning_rate exists)."""
        if float(lr) != self.lr:
            self.lr = float(lr)
            _lib.check(self.l.sed_step_state_update(_lib.ptr(self.state), 0, self.lr, 2, _lib.stream_ptr()),
                       "sed_step_state_update")

    def _allreduce(self, lo, hi, async_op):
        if self._p2p is not None:                      # one kernel on the current stream; nothing to wait for on the host
            self._p2p.all_reduce(self.grads, lo, hi)
            return None
        return sdist.allreduce_bucket(self.grads, lo, hi, self.pg, async_op=async_op, force=True)

    def _dp_tail(self):
        """Second stream: weight gradients of the GRU + heads bucket (parts 8), then that bucket's all-reduce - both
        while the main stream runs the conv-block backward."""
        _lib.check(self.l.sed_crnn_backward(C.byref(self.dims), _lib.ptr(self.student._flat), _lib.ptr(self.x),
                                            self._seed_s, _lib.ptr(self.ctx_s), self.ctx_bytes,
                                            _lib.ptr(self.d_strong), _lib.ptr(self.d_weak), _lib.ptr(self.grads),
                                            _lib.ptr(self.ws), self.ws_bytes, 8, _lib.stream_ptr()), "sed_crnn_backward")

    def _dp_step_eager_collectives(self, graph):
        """One data-parallel step with the collectives issued between graph segments (any backend)."""
        (tlo, thi), (hlo, hhi) = self._buckets
        if self.dp_schedule == "single" or self.cnn_frozen:
            self._graph_a.replay() if graph else self._fwd_bwd()
            # synchronous form: this torch runs it on the CURRENT stream (no hop to the process group's own stream and back)
            if self.cnn_frozen:
                self._allreduce(tlo, thi, False)
            else:
                self._allreduce(0, self.n, False)
            self._graph_b.replay() if graph else self._update()
            return
        cur = torch.cuda.current_stream()
        self._graph_a.replay() if graph else self._fwd_bwd()                  # ... backward of heads + GRU (parts 5)
        self._dp_stream.wait_stream(cur)
        with torch.cuda.stream(self._dp_stream):
            self._graph_w.replay() if graph else self._dp_tail()              # tail weight gradients (parts 8)
            w1 = self._allreduce(tlo, thi, True)                              # tail bucket over xGMI ...
        self._graph_c.replay() if graph else self._backward(2)                # ... while the conv blocks run backward
        # ONE collective in flight per communicator: the head bucket's all-reduce is issued only after the tail bucket's
        # has completed in stream order (it finished long before: the conv backward is ~350 us) - two collectives enqueued
        # from two streams on one communicator are otherwise ordered only by what the backend serialises internally
        with torch.cuda.stream(self._dp_stream):
            if w1 is not None:
                w1.wait()
        cur.wait_stream(self._dp_stream)
        w2 = self._allreduce(hlo, hhi, True)
        if w2 is not None:
            w2.wait()
        self._graph_b.replay() if graph else self._update()

    def _dp_step_body(self, after_forward=None, at_recurrence=None):
        """The same schedule as straight-line stream code (what SED_DP_CAPTURE=1 captures into ONE graph)."""
        (tlo, thi), (hlo, hhi) = self._buckets
        self._fwd_bwd(after_forward, at_recurrence)
        if self.dp_schedule == "single" or self.cnn_frozen:
            self._allreduce(tlo if self.cnn_frozen else 0, thi if self.cnn_frozen else self.n, False)
        else:
            cur = torch.cuda.current_stream()
            self._dp_stream.wait_stream(cur)
            with torch.cuda.stream(self._dp_stream):
                self._dp_tail()
                self._allreduce(tlo, thi, False)
            self._backward(2)
            cur.wait_stream(self._dp_stream)          # explicit order: tail all-reduce complete before the head one is issued
            self._allreduce(hlo, hhi, False)
        self._update()

    def _step_body(self, after_forward=None, at_recurrence=None):
        """One whole step as straight-line stream code - the unit a single hipGraph holds (one process, or data-parallel
        with captured collectives)."""
        if self.dp:
            self._dp_step_body(after_forward, at_recurrence)
        else:
            self._fwd_bwd(after_forward, at_recurrence)
            self._update()

    @property
    def single_graph(self):
        """True when run() replays ONE graph per step (what features.WaveformFrontEnd can fold the next batch's features into)."""
        return self.use_graph and (not self.dp or self.dp_capture)

    # ---- public ------------------------------------------------------------------------------------
    def load_batch(self, x, x_ema, target):
        self.invalidate_batch()
        self.x.copy_(x.reshape(self.x.shape), non_blocking=True)
        if x_ema is not None:
            self.x_ema.copy_(x_ema.reshape(self.x.shape), non_blocking=True)
        self.target.copy_(target, non_blocking=True)

    def run(self):
        """One step on the batch currently in self.x / self.x_ema / self.target."""
        if self._p2p is not None and self._p2p.poll():
            # (a pinned host word the kernel raises: no synchronisation.  The launch that timed out has already NaN-filled its
            # gradients; this stops the run at the next step instead of letting it train on)
            raise _lib.SedError(f"peer all-reduce: {self._p2p.poll()} cross-rank waits ran out of their budget (a rank did not "
                                "launch the same sequence of collectives, died, or fell further behind than SED_P2P_TIMEOUT_S); "
                                "the gradients of that step were filled with NaN")
        # which form of the step: forwards that find their patch moments in ctx (valid: the previous run left them there) and / or
        # a tail that computes them for the next run (the batch is resident: nothing was loaded since the previous run)
        key = (self._mom_valid, self._resident) if self.moments_ahead else (self._mom_external, False)
        self._mom_ready, self._mom_next = key
        graph = self.use_graph and self._warm >= 2
        if graph:
            gs = self._graph_sets.get(key)
            if gs is None:
                mode = (self.dp_capture, self.dp_schedule)
                self._graph_a = self._graph_w = self._graph_c = self._graph_b = None
                self._capture()
                if (self.dp_capture, self.dp_schedule) != mode:       # a capture fell back to another schedule: for every form
                    self._graph_sets = {}
                gs = self._graph_sets[key] = (self._graph_a, self._graph_w, self._graph_c, self._graph_b)
            self._graph_a, self._graph_w, self._graph_c, self._graph_b = gs
        if not self.dp:
            if graph:
                self._graph_a.replay()
            else:
                self._fwd_bwd()
                self._update()
        elif self.dp_capture:
            self._graph_a.replay() if graph else self._dp_step_body()
        else:
            self._dp_step_eager_collectives(graph)
        self._warm += 1
        self.steps_done += 1
        self._mom_valid = self._mom_next       # (the tail computed them for the batch that is in self.x now)
        self._resident = True
        self._mom_ready = self._mom_next = False

    def step(self, x, x_ema, target):
        """``x_ema`` is ignored (may be None) in supervised mode."""
        self.load_batch(x, x_ema, target)
        self.run()

    def _capture(self):
        torch.cuda.synchronize(self.device)
        cap = dict(stream=self._cap_stream)
        if self.dp:
            cap["capture_error_mode"] = "thread_local"      # see _collective_capture_works
        if self.dp and not self.dp_capture:
            overlap = self.dp_schedule == "overlap" and not self.cnn_frozen
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, **cap):
                self._fwd_bwd()
            if overlap:
                gw, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(gw, **cap):
                    self._dp_tail()
                with torch.cuda.graph(gc, **cap):
                    self._backward(2)
                self._graph_w, self._graph_c = gw, gc
            with torch.cuda.graph(gb, **cap):
                self._update()
            self._graph_a, self._graph_b = ga, gb
        elif self.dp:
            # collectives captured into the step's graph.  Every rank must end up with the same schedule: if the capture
            # fails anywhere, all ranks fall back to "single" with an eager collective between two graph segments.
            import torch.distributed as dist
            ok = 1.0
            try:
                ga = torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga, **cap):
                    self._dp_step_body()
            except Exception as e:                     # noqa: BLE001 - anything the capture raises means "not capturable here"
                ok, ga = 0.0, None
                self._capture_error = repr(e)
            flag = torch.tensor([ok], device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
            if flag.item() == 1.0:
                self._graph_a = ga
            else:
                self.dp_capture, self.dp_schedule = False, "single"
                torch.cuda.synchronize(self.device)
                self._capture()
        else:
            ga = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, **cap):
                self._fwd_bwd()
                self._update()
            self._graph_a = ga
        # capture executes nothing: the captured work runs on replay

    def meters(self, reduce=False):
        """The meters main.train logs (main.py:106-149); ONE device->host copy.  They are rank-local (each rank's share
        of the batch); ``reduce=True`` averages them over the process group - equal shares, so this is the meter of the
        global batch."""
        m = self.losses[:8].clone()
        if reduce and self.pg is not None and self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(m, op=dist.ReduceOp.SUM, group=self.pg)
            m /= self.world
        return dict(zip(LOSS_NAMES, m.tolist()))

    def rendezvous(self):
        """Host-side barrier of the data-parallel ranks + a device synchronise.  The in-kernel waits of the peer all-reduce tolerate
        SED_P2P_TIMEOUT_S (default 600 s) of skew between ranks, as a blocking collective would; call this after rank-asymmetric
        work that may take longer (a long validation on rank 0 only), so that no rank's GPU sits in a spinning kernel meanwhile.
        It is NOT triggered from a local clock inside run(): a rank that was delayed and one that was not would disagree on
        whether to enter the barrier, and the late one would wait for ever."""
        torch.cuda.synchronize(self.device)
        if self.pg is not None and self.world > 1:
            import torch.distributed as dist
            dist.barrier(group=self.pg)

    def check_health(self):
        """Raise if a cross-workgroup wait of the wide model's cluster recurrence (csrc/ggru.hip) ever timed out: the
        kernels then carried on with a stale hidden state and every result since is suspect.  The counter is sticky (only
        sed_crnn_buffers_init clears it); one 4-byte device->host copy per model.  train() calls this whenever it reads
        the meters."""
        if self._p2p is not None:
            n = self._p2p.errors()
            if n:
                raise _lib.SedError(f"peer all-reduce: {n} cross-rank waits timed out (a rank did not launch the same "
                                    "sequence of collectives, or died); gradients since then are invalid")
        if self._err_view is None:
            return
        off, nb = self._err_view
        for name, ctx in (("student", self.ctx_s), ("teacher", self.ctx_t)):
            if ctx.numel():
                n = int(ctx[off:off + 4].view(torch.int32).item())
                if n:
                    raise _lib.SedError(f"cluster GRU recurrence: {n} cross-workgroup waits timed out in the {name} model "
                                        "(the four workgroups of a chain were not co-resident); results are invalid")

    def read_state(self):
        raw = bytes(self.state.cpu().numpy().tobytes())
        return _lib.SedStepState.from_buffer_copy(raw)

    # ---- checkpoint / true resume (the reference can save but not resume, main.py:293-354) -------
    def optimizer_state_dict(self):
        """The Adam state in torch.optim.Adam's own state_dict layout (what main.py:302-305 stores under
        state['optimizer']['state_dict']), so the checkpoint stays loadable by stock torch."""
        st = self.read_state()
        params = list(self.student.parameters())
        state = {}
        for i, ((o0, o1, shp), _) in enumerate(zip(self.student._layout, params)):
            state[i] = {"step": torch.tensor(float(st.opt_step - 1)),
                        "exp_avg": self.exp_avg[o0:o1].reshape(shp).detach().cpu().clone(),
                        "exp_avg_sq": self.exp_avg_sq[o0:o1].reshape(shp).detach().cpu().clone()}
        group = {"lr": st.lr, "betas": (st.beta1, st.beta2), "eps": st.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(params)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        for i, (o0, o1, _) in enumerate(self.student._layout):
            if i in sd["state"]:
                self.exp_avg[o0:o1].copy_(sd["state"][i]["exp_avg"].reshape(-1))
                self.exp_avg_sq[o0:o1].copy_(sd["state"][i]["exp_avg_sq"].reshape(-1))

    def state_dict(self):
        """Everything a bit-exact resume needs: both models in the reference's nested checkpoint layout (plus the
        attention layer it drops, CRNN.py:49-53), the Adam moments, and the device step state (step counters - hence
        the ramp-up, EMA alpha, bias corrections - and the dropout / noise seed chain)."""
        return {"format": 1,
                "model": {k: {n: t.detach().cpu().clone() for n, t in v.items()} for k, v in self.student.state_dict().items()},
                "model_ema": ({k: {n: t.detach().cpu().clone() for n, t in v.items()}
                               for k, v in self.teacher.state_dict().items()} if self.teacher is not None else None),
                "optimizer": self.optimizer_state_dict(),
                "step_state": bytes(self.state.cpu().numpy().tobytes()),
                "seed_user": self.seed_user,
                "steps_done": self.steps_done}

    def load_state_dict(self, sd):
        assert sd.get("format") == 1, "unknown checkpoint format"
        self.student.load(parameters=sd["model"])
        if self.teacher is not None and sd.get("model_ema") is not None:
            self.teacher.load(parameters=sd["model_ema"])
        self.student.flatten_parameters_(self.device)
        if self.teacher is not None:
            self.teacher.flatten_parameters_(self.device)
        self.load_optimizer_state_dict(sd["optimizer"])
        raw = np.frombuffer(sd["step_state"], dtype=np.uint8).copy()
        assert raw.size == self.state.numel(), "step state size mismatch"
        self.state.copy_(torch.from_numpy(raw))
        self.seed_user = int(sd.get("seed_user", self.seed_user))
        self.steps_done = int(sd.get("steps_done", 0))
        self._gs_offset = int(_lib.SedStepState.from_buffer_copy(bytes(sd["step_state"])).global_step) - self.steps_done
        if self.rank != 0:       # the file holds rank 0's stream; every other rank re-folds its own rank into the seed
            _lib.check(self.l.sed_step_state_update(_lib.ptr(self.state), self._folded_seed(), 0.0, 1, _lib.stream_ptr()),
                       "sed_step_state_update")
        torch.cuda.synchronize(self.device)

    def save_checkpoint(self, path, extra=None):
        self.check_health()            # never write parameters that a timed-out collective or recurrence has touched
        torch.save(dict(self.state_dict(), **(extra or {})), path)

    def load_checkpoint(self, path):
        sd = torch.load(path, map_location="cpu", weights_only=False)
        self.load_state_dict(sd)
        return sd


def train(train_loader, model, optimizer, epoch, ema_model=None, weak_mask=None, strong_mask=None, n_epoch=100,
          log=print, check_every=50):
    """main.train (main.py:52-165) with the loop body replaced by MeanTeacherStep.  With ``ema_model=None`` and
    two-element batches ``(batch_input, target)`` it is main_simple_CRNN.train (main_simple_CRNN.py:31-82).

    ``optimizer`` supplies lr / betas / eps (its per-tensor state is not used: Adam moments live in the
    step object's flat buffers, kept on ``model._mt_step`` across epochs)."""
    start = time.time()
    step_obj = getattr(model, "_mt_step", None)
    it = iter(train_loader)
    n_batches = len(train_loader)
    pg0 = optimizer.param_groups[0]

    def check(m):
        loss = m["loss"]
        assert not (loss != loss or loss > 1e5), 'Loss explosion: {}'.format(loss)       # main.py:147
        assert not loss < 0, 'Loss problem, cannot be negative'                            # main.py:148

    for i in range(n_batches):
        batch = next(it)
        if ema_model is None:
            (batch_input, target), ema_batch_input = batch, None
        else:
            batch_input, ema_batch_input, target = batch
        if step_obj is None:
            B, T = batch_input.shape[0], batch_input.shape[-2]
            dev = torch.device("cuda", torch.cuda.current_device())
            model.to(dev)
            if ema_model is not None:
                ema_model.to(dev)
            step_obj = MeanTeacherStep(model, ema_model, B, T, n_batches * n_epoch // 2, weak_mask, strong_mask,
                                       lr=pg0["lr"], betas=pg0["betas"], eps=pg0["eps"])
            model._mt_step = step_obj
        if i == 0:
            step_obj.set_lr(pg0["lr"])        # an lr the caller changed between epochs is honoured (utils.py:227-241)
            # main.py:74: global_step = epoch * len(train_loader) + i - recomputed from `epoch` at every call.  A call sequence
            # epoch = 0, 1, 2 ... over one loader finds the device counter already there; a fresh step object entered with
            # epoch = k (a resumed run that did not load_checkpoint), a repeated or a skipped epoch gets the reference's value
            if step_obj.global_step_host != int(epoch) * n_batches:
                step_obj.set_global_step(int(epoch) * n_batches)
        step_obj.step(batch_input.to(step_obj.device, non_blocking=True),
                      ema_batch_input.to(step_obj.device, non_blocking=True) if ema_batch_input is not None else None,
                      target.to(step_obj.device, non_blocking=True))
        # the reference asserts on the loss after EVERY batch (main.py:147-148), which costs it a host sync per step;
        # here the meters are read back every `check_every` steps and after the last one
        if (i + 1) % check_every == 0:
            check(step_obj.meters())
            step_obj.check_health()
    m = step_obj.meters()
    check(m)
    step_obj.check_health()
    log('Epoch: {}\tTime {:.2f}\t{}'.format(epoch, time.time() - start,
                                            "\t".join(f"{k} {v:.4g}" for k, v in m.items())))
    return m
