"""Phase timestamps of an instrumented kernel (library built with `make EXTRA=-DSED_TS`): replays one kernel and
prints, per stamp index, the mean / max offset from the earliest stamp 0 over all workgroups.
Usage (GPU box): python tools/ts_kernel.py glu1_bwd [bnglu|conv|...]   (second argument: the translation unit)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dcase2019_task4_amd import _lib  # noqa: E402
from dcase2019_task4_amd.train import MeanTeacherStep  # noqa: E402


def main():
    name = sys.argv[1]
    tag = sys.argv[2] if len(sys.argv) > 2 else "bnglu"
    dev = torch.device("cuda", 0)
    student, teacher = bench.build_models(dev, 0)
    x, xe, tgt, wm, sm = bench.synthetic_batch(bench.B_PER_GPU, bench.T_FRAMES, 1000, dev)
    step = MeanTeacherStep(student, teacher, bench.B_PER_GPU, bench.T_FRAMES, 10500, wm, sm, use_graph=False)
    step.load_batch(x, xe, tgt)
    for _ in range(2):
        step.run()
    torch.cuda.synchronize()
    l = _lib.lib()
    st = _lib.stream_ptr()
    for _ in range(3):
        _lib.check(l.sed_kernel_replay(name.encode(), C.byref(step.dims), _lib.ptr(step.student._flat), _lib.ptr(step.x),
                                       step._seed_s, _lib.ptr(step.ctx_s), step.ctx_bytes, _lib.ptr(step.grads),
                                       _lib.ptr(step.ws), step.ws_bytes, st), name)
        torch.cuda.synchronize()
    n = 1024 * 16
    buf = (C.c_ulonglong * n)()
    fn = getattr(l, "sed_debug_ts_" + tag)
    fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    fn.restype = C.c_int
    assert fn(buf, n) == 0
    ts = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.int64)
    live = ts[:, 0] > 0
    ts = ts[live]
    t0 = ts[:, 0].min()
    print(f"{live.sum()} workgroups; offsets in us from the first workgroup's start (100 MHz clock)")
    if (ts[:, 15] > 0).all():
        cyc = (ts[:, 15] - ts[:, 14]).astype(float)
        last = max(k for k in range(14) if (ts[:, k] > 0).all())
        us = (ts[:, last] - ts[:, 0]) / 100.0
        if os.environ.get("TS_PAIR"):            # the TSC pair brackets these two wall stamps instead of the whole kernel
            k0, k1 = (int(v) for v in os.environ["TS_PAIR"].split(","))
            us = (ts[:, k1] - ts[:, k0]) / 100.0
        print(f"  shader clock (clock64 delta / wall delta): {np.mean(cyc / us):.1f} counts/us; mean delta {cyc.mean():.0f} cycles")
    for k in range(14):
        col = ts[:, k]
        ok = col > 0
        if not ok.any():
            continue
        d = (col[ok] - t0) / 100.0
        print(f"  stamp {k:2d}: n={ok.sum():4d}  mean {d.mean():8.2f}  min {d.min():8.2f}  max {d.max():8.2f}")


if __name__ == "__main__":
    main()
