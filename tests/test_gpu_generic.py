"""-m gpu parity tests of the GENERIC kernel set (csrc/gen.h): the wide CRNN of BASELINE.json configs[4]
(nb_filters 3 x 128, n_RNN_cell 256; baseline/models/CNN.py:35-67, CRNN.py:12-31 are shape-generic) and the bf16-operand
variants (configs[2], [4]) - through the same C-ABI entry points, against the CPU oracle on identical inputs and Philox masks.

Tolerances.  fp32 (sed_dims.dtype = f32): the same bounds as the specialised kernel set - posteriors 2e-5, gradients 1e-3
of their typical magnitude.  bf16 operands: the north-star bound is 1e-3 on the posteriors "fp32"; what bf16 operands can
hold is MEASURED here and asserted with head-room (see BF16_POST_TOL / BF16_GRAD_TOL and DESIGN.md section 4b).
"""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu, synth
from tests import gpu_util as gu

pytestmark = pytest.mark.gpu

POST_TOL = 2e-5
# SED_DTYPE_BF16 (bf16 operands AND bf16 storage of the conv-block activations / gradients, bf16 W_hh and GRU projections at
# H = 256): the asserted bounds are the MEASURED maxima over the shapes below plus ~35 % head-room - posteriors 9.5e-4 (base
# geometry) / 2.3e-3 (wide), worst gradient element 0.10 of the gradient's typical magnitude.  This mode does NOT hold the north
# star's 1e-3 on the wide model; SED_DTYPE_BF16X3 (below) does, and is asserted at 1e-3.
BF16_POST_TOL_BASE = 1.3e-3
BF16_POST_TOL = 3e-3
BF16_GRAD_TOL = 1.4e-1


def _stage_report(model, inter, B, T, C, H):
    """Per-stage max errors against the oracle's intermediates (localises a failure; printed, not asserted)."""
    out = {}
    H1, H2, T3 = T // 2, T // 4, T // 8
    views = {"pool0": ("p0", (B, H1, 16, C)), "conv1": ("y1", (B, H1, 16, C)), "pool1": ("p1", (B, H2, 4, C)),
             "conv2": ("y2", (B, H2, 4, C)), "pool2": ("p2", (B, T3, 1, C))}
    for k, (name, shp) in views.items():
        got = gu.nchw(model.ctx_view(name).view(*shp)).cpu()
        want = inter[k].detach()
        out[k] = gu.report(k, got, want)[1]
    for l in range(2):
        if f"gru{l}" in inter:
            got = model.ctx_view(f"gru{l}").view(B, T3, 2 * H).cpu()
            out[f"gru{l}"] = gu.report(f"gru{l}", got, inter[f"gru{l}"].detach())[1]
    return out


def _fwd_bwd(B, T, p, C, H, dtype, seed=987654321, n_layers=2, nclass=10):
    model, params = gu.make_model(0, dropout=p, n_layers=n_layers, nclass=nclass, C=C, H=H, mfma_dtype=dtype)
    model.train()
    x = synth.make_input(40, B, T)
    tgt, wm, sm = synth.make_target(5, B, T // 8, nclass=nclass)
    rs = np.random.RandomState(99)
    s_ema = torch.tensor(rs.uniform(0.05, 0.95, (B, T // 8, nclass)), dtype=torch.float32)
    w_ema = torch.tensor(rs.uniform(0.05, 0.95, (B, nclass)), dtype=torch.float32)

    def loss_fn(s, w, dev):
        return ref_cpu.mean_teacher_loss(s, w, s_ema.to(dev), w_ema.to(dev), tgt.to(dev), wm, sm, 0.7)[0]

    s, w = model(x.cuda(), seed=gu.seed_tensor(seed) if p > 0 else None)
    loss = loss_fn(s, w, "cuda")
    loss.backward()
    torch.cuda.synchronize()
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    nb = [C, C, C]
    bn = ref_cpu.new_bn_state(nb)
    so, wo, inter = ref_cpu.crnn_forward(po, x, True, bn, gu.oracle_masks(seed, B, T, p, C, H), n_layers_RNN=n_layers,
                                         return_intermediates=True)
    lo = loss_fn(so, wo, "cpu")
    go = dict(zip(po.keys(), torch.autograd.grad(lo, list(po.values()))))
    stages = _stage_report(model, inter, B, T, C, H)
    if H == 256:      # the cluster recurrence's bounded spins never timed out
        assert int(model.ctx_view("gru_err").view(torch.int32)[0]) == 0
    return dict(s=s.detach().cpu(), w=w.detach().cpu(), loss=float(loss.detach()), g=gu.grads_dict(model),
                bn=gu.bn_state_from_model(model), so=so.detach(), wo=wo.detach(), lo=float(lo.detach()), go=go, bno=bn,
                stages=stages)


def _grad_class(name):
    return "cnn" if name.startswith("cnn") else ("rnn" if name.startswith("rnn") else "heads")


def _grad_errors(g_hip, go, per_class=None):
    """Worst gradient element per parameter, relative to the gradient's typical magnitude (rms); prints the per-layer table.
    ``per_class``: a dict that receives the maximum per layer class (cnn / rnn / heads)."""
    worst, worst_name = 0.0, None
    for n, g in go.items():
        if ".conv" in n and n.endswith("bias"):
            assert float(g_hip[n].abs().max()) < 2e-5, n          # exactly-zero gradient in front of a train-mode BN
            continue
        scale = float(g.double().norm()) / np.sqrt(g.numel()) + 1e-30
        err = float((g_hip[n] - g).abs().max())
        rel = (err - 1e-7) / scale
        print(f"[grad] {n:40s} |g|/sqrt(n) {scale:.3e}  max|err| {err:.3e}  err/typ {err / scale:.3e}")
        if per_class is not None:
            per_class[_grad_class(n)] = max(per_class.get(_grad_class(n), 0.0), rel)
        if rel > worst:
            worst, worst_name = rel, n
    return worst, worst_name


@pytest.mark.parametrize("B,T,p,C,H", [(4, 128, 0.0, 128, 256), (4, 128, 0.5, 128, 256), (4, 628, 0.5, 128, 256),
                                       (5, 150, 0.25, 128, 256), (4, 216, 0.5, 128, 64), (4, 216, 0.5, 64, 256),
                                       (4, 22, 0.5, 128, 256), (4, 2112, 0.5, 128, 256)])
def test_wide_fp32_forward_backward_vs_oracle(B, T, p, C, H, n_layers=2):
    """Wide / mixed geometries in exact fp32: the bounds of the specialised kernel set.  T = 2112: 264 GRU frames = 9 chunks of
    the H = 256 heads kernels on the capped 8 workgroups per clip (one of them takes two chunks)."""
    r = _fwd_bwd(B, T, p, C, H, "f32", n_layers=n_layers)
    es, _ = gu.report("strong", r["s"], r["so"])
    ew, _ = gu.report("weak", r["w"], r["wo"])
    assert es < POST_TOL and ew < POST_TOL
    assert r["loss"] == pytest.approx(r["lo"], rel=1e-5)
    worst, name = _grad_errors(r["g"], r["go"])
    assert worst < 1e-3, (name, worst)
    for k, v in r["bno"].items():
        np.testing.assert_allclose(r["bn"][k].numpy(), v.numpy(), rtol=3e-5, atol=3e-6, err_msg=k)


@pytest.mark.parametrize("C", [64, 128])
def test_single_layer_h256_gru_weight_gradients(C):
    """n_layers_RNN = 1 (the CRNN constructor's default, CRNN.py:13) with n_RNN_cell = 256: the W_hh weight-gradient
    problems (N = H = 256) are then WIDER than the W_ih ones (N = C), which is what sizes the split-K partial buffer -
    it used to be sized from C alone and the batch wrote 2 - 4x past it."""
    test_wide_fp32_forward_backward_vs_oracle(4, 128, 0.5, C, 256, n_layers=1)


@pytest.mark.parametrize("B,T,p,C,H", [(4, 128, 0.5, 64, 64), (4, 628, 0.5, 64, 64), (24, 628, 0.5, 64, 64), (4, 864, 0.5, 64, 64),
                                       (4, 628, 0.5, 128, 256), (8, 216, 0.0, 128, 256), (24, 628, 0.5, 128, 256)])
def test_bf16_operands_forward_backward_vs_fp32_oracle(B, T, p, C, H):
    """bf16 MFMA operands (fp32 accumulation, fp32 everything else) against the FP32 oracle: the measured error of the
    posteriors and of every gradient is printed and held to BF16_POST_TOL / BF16_GRAD_TOL."""
    r = _fwd_bwd(B, T, p, C, H, "bf16")
    es, _ = gu.report("strong (bf16 operands)", r["s"], r["so"])
    ew, _ = gu.report("weak (bf16 operands)", r["w"], r["wo"])
    worst, name = _grad_errors(r["g"], r["go"])
    print(f"[bf16] C={C} H={H} B={B} T={T}: posterior err strong {es:.2e} weak {ew:.2e}; worst gradient err/typ {worst:.2e} ({name})")
    tol = BF16_POST_TOL_BASE if (C, H) == (64, 64) else BF16_POST_TOL
    assert es < tol and ew < tol
    assert r["loss"] == pytest.approx(r["lo"], rel=5e-3)
    assert worst < BF16_GRAD_TOL, (name, worst)
    # BatchNorm statistics are fp32 sums of the conv outputs, which carry the operand rounding
    for k, v in r["bno"].items():
        if not k.endswith("num_batches_tracked"):
            np.testing.assert_allclose(r["bn"][k].numpy(), v.numpy(), rtol=2e-2, atol=5e-3, err_msg=k)


X3_POST_TOL = 1e-3           # the north star's own bound ("within 1e-3 fp32"); measured ~1e-6: products are exact to ~2^-16
X3_GRAD_TOL = 1e-2           # of the gradient's typical magnitude


@pytest.mark.parametrize("B,T,p,C,H", [(4, 128, 0.5, 64, 64), (4, 628, 0.5, 64, 64), (24, 628, 0.5, 64, 64), (5, 150, 0.25, 64, 64),
                                       (4, 128, 0.5, 128, 256), (4, 628, 0.5, 128, 256), (24, 628, 0.5, 128, 256),
                                       (5, 150, 0.25, 128, 256), (4, 22, 0.5, 128, 256), (4, 216, 0.5, 64, 256), (4, 216, 0.0, 128, 64)])
def test_bf16x3_split_operands_hold_the_north_star_tolerance(B, T, p, C, H):
    """sed_dims.dtype = SED_DTYPE_BF16X3: every operand of the 3x3 convolutions (forward and dgrad) is carried as hi + lo
    bf16 halves and a product is three bf16 MFMAs (csrc/bconv.hip).  Against the FP32 oracle on identical inputs and Philox
    masks: posteriors within the north star's 1e-3 - asserted AT 1e-3, on the base and the wide model incl. BASELINE
    configs[4]'s per-GPU shape (24, 628) - and every gradient within 1e-2 of its typical magnitude."""
    r = _fwd_bwd(B, T, p, C, H, "bf16x3")
    es, _ = gu.report("strong (bf16x3)", r["s"], r["so"])
    ew, _ = gu.report("weak (bf16x3)", r["w"], r["wo"])
    worst, name = _grad_errors(r["g"], r["go"])
    print(f"[bf16x3] C={C} H={H} B={B} T={T}: posterior err strong {es:.2e} weak {ew:.2e}; worst gradient err/typ {worst:.2e} ({name})")
    assert es < X3_POST_TOL and ew < X3_POST_TOL
    assert r["loss"] == pytest.approx(r["lo"], rel=1e-4)
    assert worst < X3_GRAD_TOL, (name, worst)
    for k, v in r["bno"].items():
        if not k.endswith("num_batches_tracked"):
            np.testing.assert_allclose(r["bn"][k].numpy(), v.numpy(), rtol=1e-3, atol=1e-4, err_msg=k)


F16_POST_TOL = 1e-3          # the north star's bound; measured 1 - 3e-4 (tests/bf16_budget.py predicts 1.1e-4 base / 3.1e-4 wide)
# SED_DTYPE_F16's OWN gradient bounds (round 6; until then the bf16 mode's 1.4e-1 was re-used): the measured maximum over the
# twelve geometries below per layer class (profiles/r06_f16_gradient_errors.md: worst element / rms of the gradient), + 35 %.
#   conv blocks   6.7e-2  (batchnorm0.bias / glu0.linear.bias at (24, 628): 64-element vectors, each a sum over 965 k pixels with
#                          cancellation; every other conv-block tensor <= 5.1e-2) - the conv-block BACKWARD is the bf16 mode's:
#                          bf16 gradient tensors and operands (8-bit significands; fp16's range does not cover 1e-9 .. 1e-4)
#   recurrences   2.6e-2 at H = 64 (fp32 recurrence; the error is what arrives from p2), 4.2e-2 at H = 256 (bf16 dgate operands)
#   heads         5.7e-3 (fp32 kernels on an fp16-accurate forward)
# The bf16 mode on the same shapes: 1.0e-1 / 8.1e-2 / 4.0e-2.
F16_GRAD_TOL = {"cnn": 9.0e-2, "rnn64": 3.6e-2, "rnn256": 5.7e-2, "heads": 7.8e-3}


@pytest.mark.parametrize("B,T,p,C,H", [(4, 128, 0.5, 64, 64), (4, 628, 0.5, 64, 64), (24, 628, 0.5, 64, 64), (4, 864, 0.5, 64, 64),
                                       (5, 150, 0.25, 64, 64), (8, 216, 0.0, 64, 64),
                                       (4, 128, 0.5, 128, 256), (4, 628, 0.5, 128, 256), (24, 628, 0.5, 128, 256),
                                       (5, 150, 0.25, 128, 256), (4, 216, 0.5, 64, 256), (4, 216, 0.0, 128, 64)])
def test_f16_forward_chain_holds_the_north_star_tolerance_at_bf16_speed(B, T, p, C, H):
    """sed_dims.dtype = SED_DTYPE_F16 (round 5): the bf16 mode with its forward chain in fp16 - operands of every forward
    GEMM-shaped operator and the activations the forward hands on (11-bit significand, same MFMA rate and bytes as bf16) - and
    the bf16 mode's backward unchanged (it reads bf16 copies the forward kernels write).  Against the FP32 oracle on identical
    inputs and Philox masks: posteriors within the north star's 1e-3, asserted AT 1e-3 on the base and the wide model incl.
    BASELINE configs[4]'s per-GPU shape (24, 628); gradients at the mode's OWN measured bounds per layer class (F16_GRAD_TOL)."""
    r = _fwd_bwd(B, T, p, C, H, "f16")
    es, _ = gu.report("strong (f16 forward)", r["s"], r["so"])
    ew, _ = gu.report("weak (f16 forward)", r["w"], r["wo"])
    cls = {}
    worst, name = _grad_errors(r["g"], r["go"], cls)
    print(f"[f16] C={C} H={H} B={B} T={T}: posterior err strong {es:.2e} weak {ew:.2e}; worst gradient err/typ {worst:.2e} ({name}); "
          f"per class cnn {cls['cnn']:.2e} rnn {cls['rnn']:.2e} heads {cls['heads']:.2e}")
    assert es < F16_POST_TOL and ew < F16_POST_TOL
    assert r["loss"] == pytest.approx(r["lo"], rel=1e-3)
    assert cls["cnn"] < F16_GRAD_TOL["cnn"], (name, cls)
    assert cls["rnn"] < F16_GRAD_TOL["rnn%d" % H], (name, cls)
    assert cls["heads"] < F16_GRAD_TOL["heads"], (name, cls)
    for k, v in r["bno"].items():
        if not k.endswith("num_batches_tracked"):
            np.testing.assert_allclose(r["bn"][k].numpy(), v.numpy(), rtol=4e-3, atol=1e-3, err_msg=k)


@pytest.mark.parametrize("B,T,p,C,H,dtype", [(8, 216, 0.5, 64, 64, "bf16"), (24, 628, 0.5, 64, 64, "f16"), (4, 216, 0.5, 128, 256, "bf16"),
                                             (5, 150, 0.0, 64, 64, "bf16")])
def test_block0_saved_gates_against_the_recomputing_backward(B, T, p, C, H, dtype):
    """Round 6, measured and NOT the default (debug bit 28 turns it on): in the bf16 family the differentiated
    forward can store block 0's GLU gate sigmoid(z) as one byte per element and k_blk0_bwd<..., SG = 1> read it instead of
    recomputing z on the MFMA + exp2 + rcp.  The posteriors are untouched, every gradient outside block 0 agrees to rounding
    noise, and block 0's parameter gradients move by the quantisation only (<= 1 / 510 per gate, unbiased, inside sums over
    ~10^5 - 10^6 pixels): asserted within 3e-2 of each gradient's rms (measured 1.2e-2 on the worst element of conv0.weight),
    inside the distance of either form from the fp32 oracle."""
    from dcase2019_task4_amd import _lib
    l = _lib.lib()
    res = {}
    for bit in (1 << 28, 0):
        prev = l.sed_debug_set(bit)
        try:
            r = _fwd_bwd(B, T, p, C, H, dtype)
        finally:
            l.sed_debug_set(prev)
        res[bit] = r
    a, b = res[1 << 28], res[0]                      # a: saved gates, b: the default (recomputed)
    assert float((a["s"] - b["s"]).abs().max()) < 1e-3 and float((a["w"] - b["w"]).abs().max()) < 1e-3
    blk0 = ("cnn.cnn.conv0.", "cnn.cnn.batchnorm0.", "cnn.cnn.glu0.")
    worst = 0.0
    for n, ga in a["g"].items():
        gb = b["g"][n]
        if n.startswith(blk0):
            if n.endswith("conv0.bias"):
                continue                                # (exactly-zero gradient in front of a train-mode BatchNorm)
            rms = float(gb.double().norm()) / np.sqrt(gb.numel()) + 1e-30
            d = float((ga - gb).abs().max()) / rms
            worst = max(worst, d)
            print(f"[saved gates] {n:32s} max|saved - recomputed| / rms {d:.2e}")
            assert d < 3e-2, (n, d)
        elif dtype == "bf16" or True:
            # (the rest of the backward never sees block 0's gates; fp64 atomics make bf16-family runs agree to rounding noise only)
            rms = float(gb.double().norm()) / np.sqrt(gb.numel()) + 1e-30
            assert float((ga - gb).abs().max()) / rms < 5e-3, n
    print(f"[saved gates] {dtype} C={C} H={H} B={B} T={T}: worst block-0 gradient difference / rms {worst:.2e}")


def test_f16_stores_saturate_instead_of_overflowing():
    """fp16's range ends at 65504.  Nothing in this model comes near it, but nothing forbids it either (a BatchNorm makes the scale
    of the convolution in front of it free): with conv1's weights scaled by 1e5 its outputs reach ~1e6.  The fp16 stores saturate,
    the BatchNorm behind them absorbs it: posteriors and gradients stay finite (an Inf would turn the whole step into NaNs)."""
    model, params = gu.make_model(0, dropout=0.5, mfma_dtype="f16")
    with torch.no_grad():
        dict(model.named_parameters())["cnn.cnn.conv1.weight"].mul_(1e5)
    model.train()
    x = synth.make_input(41, 4, 216)
    s, w = model(x.cuda(), seed=gu.seed_tensor(123))
    (s.sum() + w.sum()).backward()
    torch.cuda.synchronize()
    y1 = model.ctx_view("y1")
    assert float(y1.abs().max()) > 6.0e4, "the test must actually reach the end of fp16's range"
    assert torch.isfinite(s).all() and torch.isfinite(w).all() and torch.isfinite(y1).all()
    for n, p in model.named_parameters():
        assert torch.isfinite(p.grad).all(), n


@pytest.mark.parametrize("C,H,dtype", [(128, 256, "f32"), (64, 64, "bf16"), (128, 256, "bf16x3"), (64, 64, "f16"), (128, 256, "f16")])
def test_generic_eval_forward_vs_oracle(C, H, dtype):
    """Eval mode (running statistics, no dropout), B = 1 (the reference's evaluation loop) and B = 3."""
    for B, T in ((1, 628), (3, 864)):
        model, params = gu.make_model(3, dropout=0.5, C=C, H=H, mfma_dtype=dtype)
        st = ref_cpu.new_bn_state([C] * 3)
        rs = np.random.RandomState(17)
        for k in st:
            if k.endswith("running_mean"):
                st[k] = torch.tensor(rs.normal(0, 0.2, st[k].shape), dtype=torch.float32)
            elif k.endswith("running_var"):
                st[k] = torch.tensor(rs.uniform(0.5, 1.5, st[k].shape), dtype=torch.float32)
        gu.set_bn(model, st)
        model.eval()
        x = synth.make_input(50 + B, B, T)
        with torch.no_grad():
            s, w = model(x.cuda())
        so, wo = ref_cpu.crnn_forward(params, x, False, st, None, n_layers_RNN=2)
        tol = {"f32": POST_TOL, "bf16": BF16_POST_TOL, "bf16x3": X3_POST_TOL, "f16": F16_POST_TOL}[dtype]
        es, _ = gu.report(f"eval strong {dtype}", s.cpu(), so.detach())
        ew, _ = gu.report(f"eval weak {dtype}", w.cpu(), wo.detach())
        assert s.shape == (B, T // 8, 10) and es < tol and ew < tol


def test_wide_crnn_vs_real_reference_goldens(golden_dir):
    """G10: the HIP path at the wide geometry (fp32) against outputs of the REAL reference CRNN / main.train built with
    nb_filters = [128] * 3, n_RNN_cell = 256 (oracle/gen_golden.py wide_): eval posteriors at T = 628, train-mode forward +
    BatchNorm buffers, and two fused steps (meters, parameters of student and EMA teacher)."""
    from dcase2019_task4_amd.train import MeanTeacherStep
    g = np.load(os.path.join(golden_dir, "g10_wide.npz"))
    C, H = 128, 256
    model, _ = gu.make_model(0, dropout=0.5, C=C, H=H)
    rs = np.random.RandomState(5000)
    st = ref_cpu.new_bn_state([C] * 3)
    for i in range(3):
        st[f"cnn.cnn.batchnorm{i}.running_mean"] = torch.tensor(rs.normal(0, 0.2, C), dtype=torch.float32)
        st[f"cnn.cnn.batchnorm{i}.running_var"] = torch.tensor(rs.uniform(0.5, 1.5, C), dtype=torch.float32)
    gu.set_bn(model, st)
    model.eval()
    with torch.no_grad():
        s, w = model(synth.make_input(628, 2, 628).cuda())
    es, _ = gu.report("G10 eval strong", s.cpu(), g["eval_strong"])
    ew, _ = gu.report("G10 eval weak", w.cpu(), g["eval_weak"])
    assert es < POST_TOL and ew < POST_TOL
    model, _ = gu.make_model(0, dropout=0, C=C, H=H)
    model.train()
    with torch.no_grad():
        for it in range(2):
            s, w = model(synth.make_input(10 + it, 4, 216).cuda())
            assert gu.report(f"G10 train strong{it}", s.cpu(), g[f"train_strong{it}"])[0] < POST_TOL
            assert gu.report(f"G10 train weak{it}", w.cpu(), g[f"train_weak{it}"])[0] < POST_TOL
    for k, v in model.named_buffers():
        np.testing.assert_allclose(v.cpu().numpy(), g["tb_" + k.replace(".", "_")], rtol=3e-5, atol=3e-6, err_msg=k)
    B, T = 8, 216
    student, _ = gu.make_model(0, dropout=0, C=C, H=H)
    teacher, _ = gu.make_model(1, dropout=0, C=C, H=H)
    student.train(); teacher.train()
    _, wm, sm = synth.make_target(0, B, T // 8)
    step = MeanTeacherStep(student, teacher, B, T, 2 * 100 // 2, wm, sm, use_graph=False)
    key = {"weak_class_loss": "meter_weak_class_loss", "weak_ema_loss": "meter_Weak_EMA_loss", "strong_loss": "meter_Strong_loss",
           "strong_ema_loss": "meter_Strong_EMA_loss", "cons_strong": "meter_Consistency_strong",
           "cons_weak": "meter_Consistency_weak", "loss": "meter_Loss"}
    for it in range(2):
        tgt, _, _ = synth.make_target(it, B, T // 8)
        step.step(synth.make_input(20 + it, B, T).cuda(), synth.make_input(30 + it, B, T).cuda(), tgt.cuda())
        m = step.meters()
        for k, gk in key.items():
            assert m[k] == pytest.approx(float(g[gk][it]), rel=1e-4, abs=1e-9), (it, k)
    for (n, p), (_, pe) in zip(student.named_parameters(), teacher.named_parameters()):
        k = n.replace(".", "_")
        tol = 1e-2 if (".conv" in n and n.endswith("bias")) else 3e-5
        np.testing.assert_allclose(p.detach().flatten()[:16].cpu().numpy(), g["pS_head_" + k], atol=tol, err_msg=n)
        np.testing.assert_allclose(pe.detach().flatten()[:16].cpu().numpy(), g["pT_head_" + k], atol=tol, err_msg=n)
        assert float(p.detach().double().sum()) == pytest.approx(float(g["pS_sum_" + k]), abs=tol * p.numel())


@pytest.mark.parametrize("use_graph,B,T", [(False, 4, 128), (True, 4, 128), (False, 4, 628), (True, 4, 1040)])
def test_wide_fused_steps_vs_oracle_trajectory(use_graph, B, T):
    """Two fused mean-teacher steps of the wide CRNN (fp32, dropout 0.5) against MeanTeacherOracle with the same masks.
    T = 628 / 1040: 78 / 130 GRU frames = 3 / 5 chunks of k_heads_bwd<512>, one workgroup each (the loss partials of a multi-chunk
    launch live in the workspace; the clip-level loss terms enter through chunk 0 only) - meters, gradients and parameters must
    not notice."""
    from dcase2019_task4_amd.train import MeanTeacherStep
    from tests.test_gpu_parity import _assert_params_close
    C, H = 128, 256
    student, ps = gu.make_model(0, dropout=0.5, C=C, H=H)
    teacher, pt = gu.make_model(1, dropout=0.5, C=C, H=H)
    student.train(); teacher.train()
    _, wm, sm = synth.make_target(0, B, T // 8)
    st = MeanTeacherStep(student, teacher, B, T, 50, wm, sm, use_graph=use_graph, seed=42)
    if use_graph:
        st._warm = 2
    mt = ref_cpu.MeanTeacherOracle(ps, pt)
    for it in range(2):
        state = st.read_state()
        x, xe = synth.make_input(50 + it, B, T), synth.make_input(60 + it, B, T)
        tgt, _, _ = synth.make_target(it, B, T // 8)
        st.step(x.cuda(), xe.cuda(), tgt.cuda())
        mo, go, _ = mt.step(x, xe, tgt, wm, sm, 50, gu.oracle_masks(state.seed_student, B, T, 0.5, C, H),
                            gu.oracle_masks(state.seed_teacher, B, T, 0.5, C, H))
        m = st.meters()
        for k in ("loss", "weak_class_loss", "strong_loss", "weak_ema_loss", "strong_ema_loss"):
            assert m[k] == pytest.approx(mo[k], rel=2e-4), (it, k)
        if it == 0:
            gh = {n: st.grads[o0:o1].view(shp).cpu() for n, (o0, o1, shp) in zip(go.keys(), student._layout)}
            worst, name = _grad_errors(gh, go)
            assert worst < 1e-3, (name, worst)
    for n, p in student.named_parameters():
        _assert_params_close(p.detach().cpu().numpy(), mt.p[n].detach().numpy(), n, 2)
    for n, p in teacher.named_parameters():
        _assert_params_close(p.detach().cpu().numpy(), mt.pe[n].numpy(), n, 2)


def test_bf16_step_is_bitwise_reproducible_and_dtype_is_explicit():
    """Same step twice from the same state -> identical bits (no order-dependent reductions in the generic kernels);
    and the bf16 path is only ever taken when the caller asked for it (different results from the fp32 path)."""
    from dcase2019_task4_amd.train import MeanTeacherStep
    B, T = 8, 216
    tgt, wm, sm = synth.make_target(1, B, T // 8)
    outs = {}
    for dtype in ("f32", "bf16"):
        s, _ = gu.make_model(0, dropout=0.5, mfma_dtype=dtype)
        t, _ = gu.make_model(1, dropout=0.5, mfma_dtype=dtype)
        s.train(); t.train()
        st = MeanTeacherStep(s, t, B, T, 40, wm, sm, seed=99, use_graph=False)
        assert st.dims.dtype == (1 if dtype == "bf16" else 0)
        st.load_batch(synth.make_input(60, B, T).cuda(), synth.make_input(70, B, T).cuda(), tgt.cuda())
        sd = st.state_dict()
        ref = None
        for rep in range(6):
            st.load_state_dict(sd)
            st.run()
            torch.cuda.synchronize()
            cur = [st.grads.clone(), st.strong.clone(), s._flat.clone(), t._flat.clone(), s._bn_flat.clone()]
            if ref is None:
                ref = cur
            else:
                for k, (a, b) in enumerate(zip(cur, ref)):
                    assert torch.equal(a, b), (dtype, rep, k, float((a - b).abs().max()))
        outs[dtype] = ref
    assert not torch.equal(outs["f32"][1], outs["bf16"][1])
    assert float((outs["f32"][1] - outs["bf16"][1]).abs().max()) < BF16_POST_TOL


# ---- 50-step trajectories (every reduced-precision test was one forward / backward or two steps) ------
_TRAJ = {}
TRAJ_STEPS, TRAJ_B, TRAJ_T = 50, 8, 216
# loss of step k relative to the fp32 oracle's loss of step k, maximum over the 50 steps, and the strong posteriors after step 50.
# Measured (round 4): f32 3.5e-6 / 1.2e-5, bf16x3 3.7e-6 / 1.3e-5, bf16 4.9e-4 / 3.5e-3; asserted with head-room.
# Round 6: the f16 mode has its OWN bounds (until then it borrowed bf16's): measured 1.06e-4 / 9.2e-4 - its fp16 forward keeps the
# 50-step drift 4 x below the bf16 mode's although both share the bf16 backward.
TRAJ_TOL = {"f32": 5e-5, "bf16x3": 5e-5, "bf16": 2e-3, "f16": 4e-4}
TRAJ_POST_TOL = {"f32": 1e-4, "bf16x3": 1e-4, "bf16": 7e-3, "f16": 2e-3}


def _hip_trajectory(dtype):
    from dcase2019_task4_amd.train import MeanTeacherStep
    B, T = TRAJ_B, TRAJ_T
    student, ps = gu.make_model(0, dropout=0.5, mfma_dtype=dtype)
    teacher, pt = gu.make_model(1, dropout=0.5, mfma_dtype=dtype)
    student.train(); teacher.train()
    x, xe = synth.make_input(11, B, T), synth.make_input(12, B, T)
    tgt, wm, sm = synth.make_target(3, B, T // 8)
    st = MeanTeacherStep(student, teacher, B, T, 40, wm, sm, seed=2024, use_graph=False)
    st.load_batch(x.cuda(), xe.cuda(), tgt.cuda())
    seeds, losses, weak = [], [], []
    for _ in range(TRAJ_STEPS):
        s = st.read_state()
        seeds.append((s.seed_student, s.seed_teacher))
        st.run()
        m = st.meters()
        losses.append(m["loss"]); weak.append(m["weak_class_loss"])
    return dict(seeds=seeds, loss=np.array(losses), weak=np.array(weak), post=st.strong.cpu().numpy(), ps=ps, pt=pt,
                batch=(x, xe, tgt, wm, sm))


def _oracle_trajectory(hip):
    """The fp32 oracle on the same batch with the SAME Philox dropout masks the device drew, step by step (computed once)."""
    if "oracle" not in _TRAJ:
        x, xe, tgt, wm, sm = hip["batch"]
        mt = ref_cpu.MeanTeacherOracle(hip["ps"], hip["pt"])
        losses, post = [], None
        for k, (ss, stt) in enumerate(hip["seeds"]):
            mo, _, (so, _, _, _) = mt.step(x, xe, tgt, wm, sm, 40, gu.oracle_masks(ss, TRAJ_B, TRAJ_T, 0.5),
                                           gu.oracle_masks(stt, TRAJ_B, TRAJ_T, 0.5))
            losses.append(mo["loss"]); post = so.numpy()
        _TRAJ["oracle"] = dict(loss=np.array(losses), post=post, seeds=hip["seeds"])
    assert _TRAJ["oracle"]["seeds"] == hip["seeds"], "the dropout key sequence must not depend on the arithmetic mode"
    return _TRAJ["oracle"]


@pytest.mark.parametrize("dtype", ["f32", "bf16x3", "bf16", "f16"])
def test_fifty_step_loss_trajectory_tracks_the_fp32_oracle(dtype):
    """50 mean-teacher steps (B = 8, T = 216, dropout 0.5, the same Philox masks on both sides, Adam + EMA + consistency ramp)
    on the device in each arithmetic mode against the fp32 CPU oracle: the loss of EVERY step within a stated relative
    bound, and the student's strong posteriors after the 50th step within a stated absolute bound.  What it shows: the 10 %
    worst-element gradient error SED_DTYPE_BF16 is allowed (BF16_GRAD_TOL) does not bend the trajectory."""
    hip = _hip_trajectory(dtype)
    ora = _oracle_trajectory(hip)
    rel = np.abs(hip["loss"] - ora["loss"]) / np.abs(ora["loss"])
    perr = np.abs(hip["post"] - ora["post"]).max()
    print(f"[trajectory {dtype}] loss {ora['loss'][0]:.4f} -> {ora['loss'][-1]:.4f} (oracle); max rel deviation {rel.max():.2e} "
          f"at step {rel.argmax()}, last step {rel[-1]:.2e}; posteriors after step 50: {perr:.2e}")
    assert np.isfinite(hip["loss"]).all()
    assert ora["loss"][-1] < 0.9 * ora["loss"][0], "the run must actually train"
    assert rel.max() < TRAJ_TOL[dtype], (dtype, rel.max(), int(rel.argmax()))
    assert perr < TRAJ_POST_TOL[dtype]
