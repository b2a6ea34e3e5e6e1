"""Per-kernel HIP-event timing on the buffers of a real step (tuning aid).
Usage (GPU box): python tools/kbench.py [name ...]     default: every replayable kernel"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dcase2019_task4_amd import _lib  # noqa: E402
from dcase2019_task4_amd.train import MeanTeacherStep  # noqa: E402

ALL = ["x_moments", "blk0_fwd", "conv1_fwd", "glu1_fwd", "conv2_fwd", "glu2_fwd", "gru0_fwd", "gru1_fwd", "heads_fwd",
       "gru1_bwd", "gru0_bwd", "glu2_bwd", "conv2_wgrad", "conv2_dgrad", "glu1_bwd", "conv1_wgrad", "conv1_dgrad",
       "blk0_bwd"]


def main():
    names = sys.argv[1:] or ALL
    dev = torch.device("cuda", 0)
    student, teacher = bench.build_models(dev, 0)
    x, xe, tgt, wm, sm = bench.synthetic_batch(bench.B_PER_GPU, bench.T_FRAMES, 1000, dev)
    step = MeanTeacherStep(student, teacher, bench.B_PER_GPU, bench.T_FRAMES, 10500, wm, sm, use_graph=False)
    step.load_batch(x, xe, tgt)
    for _ in range(3):
        step.run()
    torch.cuda.synchronize()
    l = _lib.lib()
    st = _lib.stream_ptr()
    for name in names:
        def call():
            _lib.check(l.sed_kernel_replay(name.encode(), C.byref(step.dims), _lib.ptr(step.student._flat), _lib.ptr(step.x),
                                           step._seed_s, _lib.ptr(step.ctx_s), step.ctx_bytes, _lib.ptr(step.grads),
                                           _lib.ptr(step.ws), step.ws_bytes, st), name)
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            call()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:14s} {e0.elapsed_time(e1) * 1e3 / 30:9.2f} us", flush=True)


if __name__ == "__main__":
    main()
