"""CPU-side checks of the C-ABI boundary: the shared library builds/loads and exports exactly the
symbols include/dcase_sed.h declares; argument validation paths that need no GPU."""
import ctypes as C
import os
import re

import pytest

from dcase2019_task4_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(REPO, "include", "dcase_sed.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sed_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _header_functions() == _lib.exported_symbols()


def test_library_exports_every_declared_symbol():
    l = _lib.lib()
    for name in _header_functions():
        assert hasattr(l, name), name
    assert l.sed_version() >= 100


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.SedDims) == 44 and _lib.SedDims.dtype.offset == 40
    assert C.sizeof(_lib.SedStepState) == 6 * 8 + 6 * 8 + 4 * 4
    assert _lib.SedStepState.seed_student.offset == 32 and _lib.SedStepState.seed_teacher.offset == 40


def test_param_layout_matches_reference_order():
    from oracle import ref_cpu
    import numpy as np
    d = _lib.make_dims(24, 628)
    offs = _lib.param_layout(d)
    shapes = list(ref_cpu.param_shapes().values())
    assert len(offs) == len(shapes) + 1 == 39
    sizes = [int(np.prod(s)) for s in shapes]
    assert [offs[i + 1] - offs[i] for i in range(len(shapes))] == sizes
    assert offs[-1] == 214356
    d1 = _lib.make_dims(24, 628, n_layers_rnn=1)
    assert _lib.param_layout(d1)[-1] == 214356 - 2 * (192 * 128 + 192 * 64 + 2 * 192)
    # the wide CRNN of BASELINE.json configs[4]: nb_filters 3 x 128, n_RNN_cell 256 (2 132 628)
    dw = _lib.make_dims(24, 628, C_=128, H=256, dtype=_lib.DTYPE_BF16)
    offs_w = _lib.param_layout(dw)
    shapes_w = list(ref_cpu.param_shapes(nb_filters=(128, 128, 128), n_RNN_cell=256).values())
    assert [offs_w[i + 1] - offs_w[i] for i in range(len(shapes_w))] == [int(np.prod(s)) for s in shapes_w]
    assert offs_w[-1] == 2132628


def test_unsupported_configurations_fail_loudly():
    l = _lib.lib()
    bad = _lib.make_dims(24, 628, F=128)
    assert l.sed_crnn_ctx_bytes(C.byref(bad)) == 0
    assert b"64" in l.sed_last_error()
    assert l.sed_param_count(C.byref(_lib.make_dims(24, 628, nclass=40))) < 0
    assert l.sed_mel_spec_ws_bytes(1, 160000, 255, 1024, 64) == 0
    ok = _lib.make_dims(24, 628)
    assert l.sed_crnn_ctx_bytes(C.byref(ok)) > 0 and l.sed_crnn_bwd_ws_bytes(C.byref(ok)) > 0
    for kw in (dict(C_=128, H=256), dict(dtype=_lib.DTYPE_BF16), dict(C_=128, H=256, dtype=_lib.DTYPE_BF16)):
        d = _lib.make_dims(24, 628, **kw)
        assert l.sed_crnn_ctx_bytes(C.byref(d)) > 0 and l.sed_crnn_bwd_ws_bytes(C.byref(d)) > 0, kw
    for kw in (dict(C_=96), dict(H=128), dict(dtype=7)):
        assert l.sed_crnn_ctx_bytes(C.byref(_lib.make_dims(24, 628, **kw))) == 0, kw


def test_module_refuses_configs_outside_the_hot_path_and_cpu_tensors():
    import torch
    from dcase2019_task4_amd.crnn import CRNN
    kw = dict(n_in_channel=1, nclass=10, attention=True, n_RNN_cell=64, n_layers_RNN=2, activation="glu", dropout=0.5,
              kernel_size=3 * [3], padding=3 * [1], stride=3 * [1], nb_filters=[64, 64, 64], pooling=list(3 * ((2, 4),)))
    m = CRNN(**kw)
    names = [n for n, _ in m.named_parameters()]
    from oracle import ref_cpu
    assert names == list(ref_cpu.param_shapes().keys())
    sd = m.state_dict()
    assert set(sd) == {"cnn", "rnn", "dense", "dense_softmax"}
    assert "conv0.weight" in sd["cnn"] and "batchnorm2.num_batches_tracked" in sd["cnn"]
    assert "rnn.weight_hh_l1_reverse" in sd["rnn"]
    assert m.hot_path
    with pytest.raises(_lib.SedError):
        m(torch.zeros(2, 1, 64, 64))               # hot-path configuration, CPU tensor: NO fallback of any kind
    with pytest.raises(NotImplementedError):
        CRNN(**dict(kw, rnn_type="LSTM"))          # (the reference's own refusal, CRNN.py:26-27)


def test_constructor_variants_outside_the_hot_path_run_the_reference_graph_on_stock_torch(golden_dir):
    """`activation="Relu"` (the class default), leakyrelu, cg, `attention=False` (weak = strong.mean(1)), other widths /
    poolings / cell counts are ACCEPTED and served by stock torch operators (CRNN.hot_path False) - against G11, outputs of the REAL
    reference built with the same constructor arguments (oracle/gen_golden.py g11): eval and train-mode posteriors, the gradient
    norms of a loss, the updated BatchNorm running mean; parameter names and order as the reference's (`dense_softmax` only with
    attention).  The same fixtures pin the oracle's functional restatement (ref_cpu.crnn_variant_forward).  The fused step refuses
    such a module."""
    import numpy as np
    import torch
    import warnings
    from dcase2019_task4_amd.crnn import CRNN
    from dcase2019_task4_amd.train import MeanTeacherStep
    from oracle import gen_golden, ref_cpu, synth
    g = np.load(os.path.join(golden_dir, "g11_variants.npz"))
    for k, (tag, kwv) in enumerate(gen_golden.VARIANTS.items()):
        full = dict(n_in_channel=1, nclass=10, dropout=0, kernel_size=3 * [3], padding=3 * [1], stride=3 * [1],
                    nb_filters=[64, 64, 64], pooling=list(3 * ((2, 4),)))
        full.update(kwv)
        m = CRNN(**full)
        assert not m.hot_path
        names = [n for n, _ in m.named_parameters()]
        assert names == list(g[f"{tag}_param_names"]), tag
        params = synth.make_params_for([(n, tuple(p.shape)) for n, p in m.named_parameters()], seed=k)
        bn = gen_golden.synth_bn(30 + k, nb=full["nb_filters"])
        gen_golden.load_params(m, params, bn)
        T = 64 if full["pooling"][0][0] == 2 else 16
        x = synth.make_input(300 + k, 3, T)
        m.eval()
        with warnings.catch_warnings(record=True) as wrn:
            warnings.simplefilter("always")
            with torch.no_grad():
                s, w = m(x)
        assert any("outside the MI355X hot path" in str(v.message) for v in wrn)
        np.testing.assert_allclose(s.numpy(), g[f"{tag}_eval_strong"], atol=2e-6)
        np.testing.assert_allclose(w.numpy(), g[f"{tag}_eval_weak"], atol=2e-6)
        # the oracle's restatement of the same variant, same fixtures
        bn_o = {kk: v.clone() for kk, v in bn.items()}
        so, wo = ref_cpu.crnn_variant_forward(params, x, bn_o, False, activation=full["activation"], attention=full["attention"],
                                              pooling=full["pooling"], n_layers_RNN=full["n_layers_RNN"], n_RNN_cell=full["n_RNN_cell"])
        np.testing.assert_allclose(so.numpy(), g[f"{tag}_eval_strong"], atol=2e-6)
        np.testing.assert_allclose(wo.numpy(), g[f"{tag}_eval_weak"], atol=2e-6)
        m.train()
        s, w = m(x)
        ((s * s).mean() + w.sum()).backward()
        np.testing.assert_allclose(s.detach().numpy(), g[f"{tag}_train_strong"], atol=2e-6)
        np.testing.assert_allclose(w.detach().numpy(), g[f"{tag}_train_weak"], atol=2e-6)
        gn = np.array([float(p.grad.double().norm()) for _, p in m.named_parameters()])
        np.testing.assert_allclose(gn, g[f"{tag}_grad_norms"], rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(dict(m.named_buffers())["cnn.cnn.batchnorm0.running_mean"].numpy(), g[f"{tag}_bn_mean0"], atol=2e-6)
        sd = m.state_dict()
        assert ("dense_softmax" in sd) == bool(full["attention"])
        with pytest.raises(_lib.SedError):
            MeanTeacherStep(m, None, 3, T, 10, slice(1), slice(2, 3))


def test_p2p_flag_never_overtakes_the_payload_in_the_compiled_kernel():
    """csrc/p2p.hip hands data to peer GPUs with write-through stores followed by a flag store.  The flag may only be raised
    once every wave's payload stores are ACKNOWLEDGED: an explicit `s_waitcnt vmcnt(0)` on every wave, then the workgroup
    barrier, then the flag.  On gfx950 the barrier itself carries no vmcnt wait (the compiler emits `s_waitcnt lgkmcnt(0);
    s_barrier` - round 5 relied on it), so this checks the COMPILER'S OUTPUT: in every instantiation, each flag store is
    preceded by the marked drain, with exactly one s_barrier and no vector-memory store in between."""
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(REPO, "dcase2019_task4_amd", "csrc", "p2p.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "p2p.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-DSED_AB", "-S", "--cuda-device-only", src, "-o", out],
                       check=True, capture_output=True)
        text = open(out).read()
    kernels = re.split(r"^(_Z15k_p2p_allreduceILi\d+E[^:\n]*):", text, flags=re.M)
    assert len(kernels) >= 2 * 5 + 1, "expected five instantiations of k_p2p_allreduce"
    for name, body in zip(kernels[1::2], kernels[2::2]):
        body = body.split(".Lfunc_end")[0]
        lines = [ln.strip() for ln in body.split("\n")]
        flags = [i for i, ln in enumerate(lines) if "p2p_flag_store" in ln]
        drains = [i for i, ln in enumerate(lines) if "p2p_signal_drain" in ln]
        assert len(flags) >= 2 and len(drains) >= 2, (name, len(flags), len(drains))
        for f in flags:
            before = [d for d in drains if d < f]
            assert before, (name, "flag store without a drain in front of it")
            between = lines[before[-1] + 1:f]
            assert sum(ln.startswith("s_barrier") for ln in between) == 1, (name, between)
            assert not any(re.match(r"(global|flat|buffer)_(store|atomic)", ln) for ln in between), (name, between)
        for d in drains:
            assert lines[d].startswith("s_waitcnt vmcnt(0)"), lines[d]
