"""Raw stamp deltas of an instrumented kernel (see tools/ts_kernel.py): prints, per stamp, the mean delta to the previous
non-empty stamp over all workgroups.  Usage: python tools/ts_raw.py <replay name> <translation unit tag> [first] [last]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dcase2019_task4_amd import _lib
from dcase2019_task4_amd.train import MeanTeacherStep
name, tag = sys.argv[1], sys.argv[2]
k0, k1 = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 15)
dev = torch.device("cuda", 0)
student, teacher = bench.build_models(dev, 0)
x, xe, tgt, wm, sm = bench.synthetic_batch(bench.B_PER_GPU, bench.T_FRAMES, 1000, dev)
step = MeanTeacherStep(student, teacher, bench.B_PER_GPU, bench.T_FRAMES, 10500, wm, sm, use_graph=False)
step.load_batch(x, xe, tgt)
for _ in range(2):
    step.run()
torch.cuda.synchronize()
l = _lib.lib(); st = _lib.stream_ptr()
for _ in range(3):
    _lib.check(l.sed_kernel_replay(name.encode(), C.byref(step.dims), _lib.ptr(step.student._flat), _lib.ptr(step.x), step._seed_s,
                                   _lib.ptr(step.ctx_s), step.ctx_bytes, _lib.ptr(step.grads), _lib.ptr(step.ws), step.ws_bytes, st), name)
    torch.cuda.synchronize()
n = 1024 * 16
buf = (C.c_ulonglong * n)()
fn = getattr(l, "sed_debug_ts_" + tag); fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]; fn.restype = C.c_int
assert fn(buf, n) == 0
ts = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.int64)
ts = ts[ts[:, k0] > 0]
prev = k0
for k in range(k0 + 1, k1 + 1):
    if (ts[:, k] > 0).all():
        d = ts[:, k] - ts[:, prev]
        print(f"  stamp {prev:2d} -> {k:2d}: mean {d.mean():9.1f}  min {d.min():7d}  max {d.max():7d}   (counts)")
        prev = k
