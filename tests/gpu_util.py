"""Helpers shared by the -m gpu parity tests: build the HIP module with procedural weights, build the
oracle's inputs (incl. the Philox dropout masks for a given seed) and compare."""
import numpy as np
import torch

from oracle import philox, ref_cpu, synth

CRNN_KW = dict(n_in_channel=1, nclass=10, attention=True, n_RNN_cell=64, n_layers_RNN=2, activation="glu", dropout=0.5,
               kernel_size=3 * [3], padding=3 * [1], stride=3 * [1], nb_filters=[64, 64, 64],
               pooling=list(3 * ((2, 4),)))


def make_model(seed=0, dropout=0.5, device="cuda", n_layers=2, nclass=10, C=64, H=64, mfma_dtype="f32"):
    from dcase2019_task4_amd.crnn import CRNN
    kw = dict(CRNN_KW, dropout=dropout, n_layers_RNN=n_layers, nclass=nclass, nb_filters=[C] * 3, n_RNN_cell=H,
              mfma_dtype=mfma_dtype)
    m = CRNN(**kw)
    params = synth.make_params(seed, n_layers_RNN=n_layers, nclass=nclass, nb_filters=(C,) * 3, n_RNN_cell=H)
    with torch.no_grad():
        for (n, p) in m.named_parameters():
            p.copy_(params[n])
    return m.to(device), params


def set_bn(model, bn_state):
    bufs = dict(model.named_buffers())
    with torch.no_grad():
        for k, v in bn_state.items():
            bufs[k].copy_(v.to(bufs[k].device))


def oracle_masks(seed, B, T, p, C=64, H=64):
    """The four dropout masks the HIP kernels draw for Philox key ``seed`` (oracle/philox.py)."""
    if p <= 0:
        return None
    H1, H2, T3 = T // 2, T // 4, T // 8
    return {
        "drop0": torch.tensor(philox.dropout_mask_pooled(seed, 0, B, T, 64, C, p)),
        "drop1": torch.tensor(philox.dropout_mask_pooled(seed, 1, B, H1, 16, C, p)),
        "drop2": torch.tensor(philox.dropout_mask_pooled(seed, 2, B, H2, 4, C, p)),
        "drop_rnn": torch.tensor(philox.dropout_mask_flat(seed, 8, (B, T3, 2 * H), p)),
    }


def seed_tensor(seed, device="cuda"):
    return torch.tensor([seed], dtype=torch.int64, device=device)


def nchw(t_nhwc):
    return t_nhwc.permute(0, 3, 1, 2)


def report(name, got, want):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    scale = np.abs(want).max() + 1e-30
    print(f"[parity] {name:34s} max|err| {err.max():.3e}  rel-to-max {err.max() / scale:.3e}  |want|max {scale:.3e}")
    return err.max(), err.max() / scale


def grads_dict(model):
    return {n: p.grad.detach().float().cpu() for n, p in model.named_parameters()}


def bn_state_from_model(model):
    return {k: v.detach().cpu().clone() for k, v in model.named_buffers()}
