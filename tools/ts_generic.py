"""Stamp deltas of an instrumented GENERIC-path kernel (library built with -DSED_TS): runs supervised steps of the given
geometry, then prints per stamp the mean offset from stamp 0 over all workgroups (100 MHz wall clock -> us).
Usage: python tools/ts_generic.py <tag> [--C 128 --H 256 --dtype bf16]"""
import argparse, ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dcase2019_task4_amd import _lib
from dcase2019_task4_amd.train import MeanTeacherStep
ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--C", type=int, default=128); ap.add_argument("--H", type=int, default=256); ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
dev = torch.device("cuda", 0)
s, t = bench.build_models(dev, 0, nb_filters=[a.C] * 3, n_RNN_cell=a.H, mfma_dtype=a.dtype)
x, xe, tgt, wm, sm = bench.synthetic_batch(24, 628, 1, dev)
st = MeanTeacherStep(s, None, 24, 628, 100, wm, sm, use_graph=False, overlap_streams=False)
st.load_batch(x, xe, tgt)
for _ in range(4):
    st.run()
torch.cuda.synchronize()
l = _lib.lib()
n = 1024 * 16
buf = (C.c_ulonglong * n)()
fn = getattr(l, "sed_debug_ts_" + a.tag); fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]; fn.restype = C.c_int
assert fn(buf, n) == 0
ts = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.int64)
ts = ts[ts[:, 0] > 0]
print(len(ts), "workgroups")
raw = os.environ.get("TS_RAW")
prev = 0
for k in range(1, 16):
    if (ts[:, k] > 0).all():
        if raw:
            d = ts[:, k] - ts[:, prev]
            print(f"  stamp {prev:2d} -> {k:2d}: mean {d.mean():9.1f} counts  min {d.min():7d}  max {d.max():7d}")
            prev = k
        else:
            d = (ts[:, k] - ts[:, 0]) / 100.0
            print(f"  stamp {k:2d}: mean {d.mean():8.2f} us  min {d.min():8.2f}  max {d.max():8.2f}")
