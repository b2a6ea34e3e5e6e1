"""Profiling aid: N supervised steps (student only: no second stream competing for the GPU, so the rocprofv3 kernel
durations are close to solo times) of a given geometry / dtype.  Usage:
    rocprofv3 --kernel-trace --stats -d out -o p -- python tools/prof_generic.py --C 128 --H 256 --dtype bf16"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dcase2019_task4_amd.train import MeanTeacherStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--C", type=int, default=128)
ap.add_argument("--H", type=int, default=256)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--batch", type=int, default=24)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--teacher", action="store_true")
ap.add_argument("--no-side", action="store_true", help="SED_NO_SIDE: weight-gradient kernels on the caller's stream too")
a = ap.parse_args()
dev = torch.device("cuda", 0)
s, t = bench.build_models(dev, 0, nb_filters=[a.C] * 3, n_RNN_cell=a.H, mfma_dtype=a.dtype)
x, xe, tgt, wm, sm = bench.synthetic_batch(a.batch, 628, 1, dev)
st = MeanTeacherStep(s, t if a.teacher else None, a.batch, 628, 100, wm, sm, use_graph=False, overlap_streams=False)
st.load_batch(x, xe, tgt)
for _ in range(a.steps):
    st.run()
torch.cuda.synchronize()
print("ok", st.meters()["loss"])
