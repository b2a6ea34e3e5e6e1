"""Fixed (prologue / epilogue) vs per-step cost of the GRU kernels: times gru*_fwd / gru*_bwd at two clip lengths and solves
time = a + b * steps.  Usage (GPU box): python tools/gru_prologue.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dcase2019_task4_amd import _lib  # noqa: E402
from dcase2019_task4_amd.train import MeanTeacherStep  # noqa: E402


def times(T):
    dev = torch.device("cuda", 0)
    student, teacher = bench.build_models(dev, 0)
    x, xe, tgt, wm, sm = bench.synthetic_batch(bench.B_PER_GPU, T, 1000, dev)
    step = MeanTeacherStep(student, teacher, bench.B_PER_GPU, T, 10500, wm, sm, use_graph=False)
    step.load_batch(x, xe, tgt)
    for _ in range(3):
        step.run()
    torch.cuda.synchronize()
    l = _lib.lib()
    st = _lib.stream_ptr()
    out = {}
    for name in ("gru0_fwd", "gru1_fwd", "gru1_bwd", "gru0_bwd"):
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
        out[name] = e0.elapsed_time(e1) * 1e3 / 30
    return out


def main():
    Ta, Tb = 320, 1280
    a, b = times(Ta), times(Tb)
    for k in a:
        per = (b[k] - a[k]) / (Tb // 8 - Ta // 8)
        print(f"{k:10s} T={Ta}: {a[k]:6.2f} us  T={Tb}: {b[k]:6.2f} us  -> {per * 1e3:6.1f} ns per step, fixed {a[k] - per * (Ta // 8):5.2f} us")


if __name__ == "__main__":
    main()
