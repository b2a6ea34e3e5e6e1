"""Throughput of the batched inference + post-processing path next to the reference's way of doing
it (one clip per forward + host numpy/scipy post-processing, evaluation_measures.py:203-231) restated by the oracle.
Usage (GPU box): python tools/bench_inference.py [--clips 1168] [--frames 864] [--batch 64]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dcase2019_task4_amd.inference import get_predictions, postprocess  # noqa: E402


class DS:
    def __init__(self, x):
        self.x = x
        self.filenames = pd.Series([f"clip_{i}.wav" for i in range(len(x))])

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        return self.x[i], 0


class Enc:
    labels = [f"class_{i}" for i in range(10)]

    def decode_strong(self, m):
        raise RuntimeError("host decoder must not be used")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=1168)          # size of the reference's validation set
    ap.add_argument("--frames", type=int, default=864)          # config.py:17-22
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--cpu-clips", type=int, default=16)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    student, _ = bench.build_models(dev, 0)
    student.eval()
    x = torch.randn(a.clips, 1, a.frames, 64)
    ds = DS(x.pin_memory())
    get_predictions(student, DS(x[:a.batch]), Enc().decode_strong, 8, batch_size=a.batch)       # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    df = get_predictions(student, ds, Enc().decode_strong, 8, batch_size=a.batch)
    torch.cuda.synchronize()
    t_e2e = time.perf_counter() - t0
    xb = x[:a.batch].to(dev)
    with torch.no_grad():
        for _ in range(3):
            postprocess(student(xb)[0], 0.5, 5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            postprocess(student(xb)[0], 0.5, 5)
        torch.cuda.synchronize()
        t_dev = (time.perf_counter() - t0) / reps
    # the reference's way, restated: eval forward one clip at a time on the host cores + numpy/scipy post-processing
    from oracle import postprocess_np as pp, ref_cpu
    params = {k: v.detach().cpu() for k, v in student.named_parameters()}
    bn = {k: v.detach().cpu() for k, v in student.named_buffers()}
    torch.set_num_threads(bench.usable_cores())
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(a.cpu_clips):
            s, _ = ref_cpu.crnn_forward(params, x[i:i + 1], False, bn, None, 2)
            pp.decode_strong(pp.filter_decisions(s[0].numpy()), Enc.labels)
    t_cpu = (time.perf_counter() - t0) / a.cpu_clips
    print(json.dumps({
        "what": "eval-mode CRNN + threshold / median filter / run-length decode, events as the reference's DataFrame",
        "clips": a.clips, "frames": a.frames, "batch": a.batch, "events": int(len(df)),
        "end_to_end_clips_per_s": round(a.clips / t_e2e, 1),
        "device_only_clips_per_s": round(a.batch / t_dev, 1), "device_ms_per_batch": round(t_dev * 1e3, 3),
        "cpu_oracle_clip_by_clip_clips_per_s": round(1.0 / t_cpu, 2), "cpu_cores": bench.usable_cores(),
        "note": "end_to_end includes the pageable->device copies and the pandas assembly on the host"}))


if __name__ == "__main__":
    main()
