"""TEST INFRASTRUCTURE ONLY - procedural weights / inputs shared by gen_golden.py and the tests.

Golden fixtures hold reference OUTPUTS only; the inputs and weights are regenerated from fixed
numpy ``RandomState`` seeds (legacy MT19937 stream: stable across numpy versions and platforms).
Weight statistics follow weights_init (baseline/utils/utils.py:205-224) closely enough to keep
activations in a realistic range (they need not match it exactly - they only need to be identical
on both sides of a comparison).
"""
from collections import OrderedDict

import numpy as np
import torch

from .ref_cpu import param_shapes


def make_params(seed=0, dtype=torch.float32, **model_kw):
    rs = np.random.RandomState(1000 + seed)
    out = OrderedDict()
    for name, shp in param_shapes(**model_kw).items():
        if ".conv" in name and name.endswith("weight"):
            fan_in = shp[1] * 9
            fan_out = shp[0] * 9
            bound = np.sqrt(2.0) * np.sqrt(6.0 / (fan_in + fan_out))
            a = rs.uniform(-bound, bound, shp)
        elif ".conv" in name:
            a = rs.normal(0, 0.05, shp)
        elif "batchnorm" in name and name.endswith("weight"):
            a = rs.normal(1.0, 0.1, shp)
        elif "batchnorm" in name:
            a = rs.normal(0.0, 0.1, shp)
        elif "rnn.rnn.weight" in name:
            a = rs.normal(0, 1.0 / np.sqrt(shp[1]), shp)
        elif "rnn.rnn.bias" in name:
            a = rs.uniform(-0.125, 0.125, shp)
        elif name.endswith("weight"):          # glu linear, dense, dense_softmax
            a = rs.normal(0, 0.08, shp)
        else:
            a = rs.normal(0, 0.05, shp)
        out[name] = torch.tensor(a, dtype=dtype)
    return out


def make_input(seed, B, T, F=64, dtype=torch.float32):
    rs = np.random.RandomState(2000 + seed)
    return torch.tensor(rs.standard_normal((B, 1, T, F)), dtype=dtype)


def make_target(seed, B, T_out, nclass=10, dtype=torch.float32):
    """[weak | unlabeled | strong] batch layout of main.py:238-247; -1 rows for unlabeled
    (utils.py:82-85); weak rows carry the clip label on every frame (utils.py:108-111)."""
    rs = np.random.RandomState(3000 + seed)
    t = np.zeros((B, T_out, nclass))
    nw = B // 4
    ns = B // 4
    t[:nw] = (rs.uniform(size=(nw, 1, nclass)) < 0.2).astype(np.float64)
    t[nw:B - ns] = -1.0
    t[B - ns:] = (rs.uniform(size=(ns, T_out, nclass)) < 0.2).astype(np.float64)
    return torch.tensor(t, dtype=dtype), slice(nw), slice(B - ns, B)


def make_wave(seed, n_samples):
    rs = np.random.RandomState(4000 + seed)
    t = np.arange(n_samples) / 16000.0
    y = 0.3 * np.sin(2 * np.pi * (220.0 + 30 * seed) * t) + 0.1 * np.sin(2 * np.pi * 3100.0 * t * (1 + 0.1 * t))
    y = y + 0.05 * rs.standard_normal(n_samples)
    env = 0.5 + 0.5 * np.sin(2 * np.pi * 0.7 * t + seed)
    return (y * env).astype(np.float64)


def make_posteriors(seed, N, T_out, nclass=10):
    """Strong posteriors with enough temporal structure to exercise the post-processing (a randomly initialised CRNN
    emits nearly time-constant posteriors): smoothed noise pushed through a sigmoid, plus hand-placed edge cases -
    1/2/3-frame blips and gaps around the median window, activity at both clip ends, all-on / all-off / alternating
    columns and values exactly at the threshold."""
    rs = np.random.RandomState(6000 + seed)
    z = rs.standard_normal((N, T_out + 8, nclass))
    k = np.array([1, 2, 3, 2, 1], dtype=np.float64) / 9.0
    sm = sum(k[i] * z[:, i:i + T_out, :] for i in range(5))
    p = 1.0 / (1.0 + np.exp(-4.0 * sm))
    p = np.where(np.abs(p - 0.5) < 1e-3, 0.6, p)          # keep a margin that fp32 cannot flip ...
    if nclass < 8 or T_out < 40:
        return torch.tensor(p, dtype=torch.float32)
    p[0, :, 0] = 0.9                                       # all on
    p[0, :, 1] = 0.1                                       # all off
    p[0, :, 2] = np.where(np.arange(T_out) % 2 == 0, 0.8, 0.2)     # alternating: the median filter decides everything
    p[0, :, 3] = 0.1; p[0, 10, 3] = 0.9; p[0, 20:22, 3] = 0.9; p[0, 30:33, 3] = 0.9     # blips of 1, 2, 3 frames
    p[0, :, 4] = 0.9; p[0, 10, 4] = 0.1; p[0, 20:22, 4] = 0.1; p[0, 30:33, 4] = 0.1     # gaps of 1, 2, 3 frames
    p[0, :, 5] = 0.1; p[0, :2, 5] = 0.9; p[0, -2:, 5] = 0.9                               # reflect boundary, both ends
    p[0, :, 6] = 0.1; p[0, :3, 6] = 0.9; p[0, -1:, 6] = 0.9
    p[0, :, 7] = 0.5                                        # ... except exactly AT the threshold: p > 0.5 is False
    return torch.tensor(p, dtype=torch.float32)


def make_params_for(named_shapes, seed=0, dtype=torch.float32):
    """Procedural weights for ANY module layout (the constructor variants outside the hot path, G11): one draw per tensor in the
    order given, statistics by tensor kind as in make_params."""
    rs = np.random.RandomState(7000 + seed)
    out = OrderedDict()
    for name, shp in named_shapes:
        shp = tuple(shp)
        if "batchnorm" in name and name.endswith("weight"):
            a = rs.normal(1.0, 0.1, shp)
        elif name.endswith("bias") or len(shp) == 1:
            a = rs.normal(0.0, 0.08, shp)
        else:
            fan_in = int(np.prod(shp[1:]))
            a = rs.normal(0, 1.0 / np.sqrt(fan_in), shp)
        out[name] = torch.tensor(a, dtype=dtype)
    return out
