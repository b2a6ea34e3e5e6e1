"""CRNN nn.Module for MI355X - host-side mirror of baseline/models/CRNN.py (+ CNN.py, RNN.py).

Same constructor signature, ``forward(x[B,1,T,F]) -> (strong[B,T//8,nclass], weak[B,nclass])``,
nested ``state_dict()`` / ``load`` / ``save`` / ``load_cnn`` contract and submodule class names
(``weights_init`` in baseline/utils/utils.py:205-224 dispatches on ``__class__.__name__``
substrings), so ``CRNN(**cfg.crnn_kwargs)`` drops into baseline/main.py:279-287 and
baseline/TestModel.py:30-36 unchanged.  On the hot path the torch submodules below are PARAMETER
CONTAINERS only: ``forward`` never calls them.  All arithmetic runs in the hand-written HIP kernels
behind the C-ABI (include/dcase_sed.h) on one flat fp32 parameter buffer the containers' parameters
are views of, and there is no CPU / stock-torch fallback for it: without the library (or on a CPU
tensor) a hot-path module raises.

Hot path (``CRNN.hot_path``): activation="glu", attention=True, BGRU, n_in_channel=1, three 3x3 conv blocks of EQUAL
width with (2,4) pooling, 1 or 2 GRU layers, and
  * nb_filters 3 x 64, n_RNN_cell 64 - cfg.crnn_kwargs (baseline/config.py:53-58), the specialised fp32 kernel set;
  * nb_filters 3 x 64 or 3 x 128, n_RNN_cell 64 or 256 - the generic kernel set (BASELINE.json configs[4]'s wide CRNN).
``mfma_dtype`` (extra keyword, or ``set_mfma_dtype``) states the arithmetic of the GEMM-shaped operators of conv blocks
1 and 2: "f32" (default, exact fp32 MFMA) or "bf16" (bf16 operands, fp32 accumulation - include/dcase_sed.h sed_dims.dtype).

Every OTHER argument combination the reference's constructor accepts (activation "Relu" - the class default - / "leakyrelu" /
"cg", attention=False, other filter counts / kernels / poolings / cell counts; SURVEY.md 8(b): accept-and-fallback) builds the
reference's module tree and runs its graph on stock torch operators (``hot_path`` False, one warning; pinned by golden G11
against the real reference).  The fused step (train.MeanTeacherStep) and the C-ABI refuse such a module.
"""
import ctypes as C
import warnings

import torch
import torch.nn as nn

from . import _lib


class GLU(nn.Module):
    """Container for baseline/models/CNN.py:5-16 (Linear over channels x sigmoid gate)."""

    def __init__(self, input_num):
        super().__init__()
        self.sigmoid = nn.Sigmoid()
        self.linear = nn.Linear(input_num, input_num)

    def forward(self, x):
        """CNN.py:11-16 - only executed by the stock-torch path of configurations outside the hot path (CRNN.hot_path False);
        on the hot path the GLU is fused into the HIP conv-block kernels."""
        lin = self.linear(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        return lin * self.sigmoid(x)


class ContextGating(nn.Module):
    """baseline/models/CNN.py:19-30 (activation="cg"): never on the hot path, stock torch operators."""

    def __init__(self, input_num):
        super().__init__()
        self.sigmoid = nn.Sigmoid()
        self.linear = nn.Linear(input_num, input_num)

    def forward(self, x):
        lin = self.linear(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        return x * self.sigmoid(lin)


class CNN(nn.Module):
    """Container for baseline/models/CNN.py:33-89: holds conv{i}/batchnorm{i}/glu{i}/dropout{i}/pooling{i}."""

    def __init__(self, n_in_channel, activation="Relu", conv_dropout=0, kernel_size=[3, 3, 3], padding=[1, 1, 1],
                 stride=[1, 1, 1], nb_filters=[64, 64, 64], pooling=[(1, 4), (1, 4), (1, 4)]):
        super().__init__()
        # The HIP kernels implement config.py:53-58's block (and BASELINE.json configs[4]'s width): GLU, n_in_channel = 1, three
        # 3x3 / stride 1 / pad 1 convolutions of 64 or 128 filters, pooling (2, 4) x 3.  Everything else the reference's constructor
        # accepts (CNN.py:35-67) is ACCEPTED and runs on stock torch operators (accept-and-fallback for what is not
        # on the hot path); `hot` says which of the two this container is.
        act = activation.lower()
        self.hot = (act == "glu" and n_in_channel == 1 and list(nb_filters) in ([64, 64, 64], [128, 128, 128]) and
                    list(kernel_size) == [3, 3, 3] and list(padding) == [1, 1, 1] and list(stride) == [1, 1, 1] and
                    [tuple(p) for p in pooling] == [(2, 4)] * 3 and conv_dropout is not None)
        self.nb_filters = list(nb_filters)
        cnn = nn.Sequential()
        for i in range(len(nb_filters)):
            n_in = n_in_channel if i == 0 else nb_filters[i - 1]
            cnn.add_module(f"conv{i}", nn.Conv2d(n_in, nb_filters[i], kernel_size[i], stride[i], padding[i]))
            cnn.add_module(f"batchnorm{i}", nn.BatchNorm2d(nb_filters[i], eps=0.001, momentum=0.99))
            if act == "leakyrelu":
                cnn.add_module(f"relu{i}", nn.LeakyReLU(0.2))
            elif act == "relu":
                cnn.add_module(f"relu{i}", nn.ReLU())
            elif act == "glu":
                cnn.add_module(f"glu{i}", GLU(nb_filters[i]))
            elif act == "cg":
                cnn.add_module(f"cg{i}", ContextGating(nb_filters[i]))
            if conv_dropout is not None:
                cnn.add_module(f"dropout{i}", nn.Dropout(conv_dropout))
            cnn.add_module(f"pooling{i}", nn.AvgPool2d(pooling[i]))
        self.cnn = cnn

    def load(self, filename=None, parameters=None):
        if filename is not None:
            self.cnn.load_state_dict(torch.load(filename))
        elif parameters is not None:
            self.cnn.load_state_dict(parameters)
        else:
            raise NotImplementedError("load is a filename or a list of parameters (state_dict)")

    def state_dict(self, destination=None, prefix='', keep_vars=False):
        return self.cnn.state_dict(destination=destination, prefix=prefix, keep_vars=keep_vars)

    def save(self, filename):
        torch.save(self.cnn.state_dict(), filename)

    def forward(self, x):
        """CNN.py:85-89 on stock torch operators - reached only when CRNN.hot_path is False."""
        return self.cnn(x)


class BidirectionalGRU(nn.Module):
    """Container for baseline/models/RNN.py:7-16."""

    def __init__(self, n_in, n_hidden, dropout=0, num_layers=1):
        super().__init__()
        self.rnn = nn.GRU(n_in, n_hidden, bidirectional=True, dropout=dropout, batch_first=True, num_layers=num_layers)

    def forward(self, input_feat):
        """RNN.py:14-16 on stock torch operators - reached only when CRNN.hot_path is False."""
        recurrent, _ = self.rnn(input_feat)
        return recurrent


class _CRNNFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward = sed_crnn_forward, backward = sed_crnn_backward."""

    @staticmethod
    def forward(ctx, module, x, train, seed_t, *params):
        l = _lib.lib()
        dims = module._dims(x)
        B, T3, NC = dims.B, dims.T // 8, dims.nclass
        strong = torch.empty(B, T3, NC, device=x.device, dtype=torch.float32)
        weak = torch.empty(B, NC, device=x.device, dtype=torch.float32)
        ctx_bytes = l.sed_crnn_ctx_bytes(C.byref(dims))
        if ctx_bytes == 0:
            raise _lib.SedError(l.sed_last_error().decode())
        cbuf = _lib.scratch(ctx_bytes, x.device)
        _lib.check(l.sed_crnn_buffers_init(C.byref(dims), _lib.ptr(cbuf), ctx_bytes, None, 0, _lib.stream_ptr()),
                   "sed_crnn_buffers_init")
        _lib.check(l.sed_crnn_forward(C.byref(dims), _lib.ptr(module._flat), _lib.ptr(module._bn_flat),
                                      _lib.ptr(module._bn_tracked), _lib.ptr(x), int(train), 1, _lib.ptr(seed_t),
                                      _lib.ptr(cbuf), ctx_bytes, _lib.ptr(strong), _lib.ptr(weak), _lib.stream_ptr()),
                   "sed_crnn_forward")
        module._last_ctx = (cbuf, dims)          # test/debug hook (sed_crnn_ctx_view)
        if ctx.needs_input_grad[1]:
            raise _lib.SedError("the gradient w.r.t. the input features is not implemented (the reference never asks for "
                                "it: main.py:91 feeds a plain batch tensor)")
        if not train and any(ctx.needs_input_grad[4:]):
            # eval-mode autograd (running BatchNorm statistics, no dropout) is not on the hot path: refuse loudly rather
            # than fail later in backward with a missing context
            ctx.eval_mode = True
        need = train and any(ctx.needs_input_grad[4:])
        if need:
            ctx.module, ctx.dims, ctx.cbuf, ctx.seed_t = module, dims, cbuf, seed_t
            ctx.save_for_backward(x)
            ctx.flat_version = module._flat_gen
        return strong, weak

    @staticmethod
    def backward(ctx, d_strong, d_weak):
        l = _lib.lib()
        if getattr(ctx, "eval_mode", False):
            raise _lib.SedError("backward through CRNN.forward needs module.train(): the eval-mode backward (running "
                                "BatchNorm statistics) is not implemented")
        module, dims = ctx.module, ctx.dims
        (x,) = ctx.saved_tensors
        if ctx.flat_version != module._flat_gen:
            raise _lib.SedError("CRNN parameters were re-laid-out between forward and backward")
        d_strong = d_strong.contiguous().float()
        d_weak = d_weak.contiguous().float()
        gflat = torch.empty_like(module._flat)
        ws_bytes = l.sed_crnn_bwd_ws_bytes(C.byref(dims))
        ws = _lib.scratch(ws_bytes, x.device)
        _lib.check(l.sed_crnn_buffers_init(C.byref(dims), None, 0, _lib.ptr(ws), ws_bytes, _lib.stream_ptr()),
                   "sed_crnn_buffers_init")
        _lib.check(l.sed_crnn_backward(C.byref(dims), _lib.ptr(module._flat), _lib.ptr(x), _lib.ptr(ctx.seed_t),
                                       _lib.ptr(ctx.cbuf), ctx.cbuf.numel(), _lib.ptr(d_strong), _lib.ptr(d_weak),
                                       _lib.ptr(gflat), _lib.ptr(ws), ws_bytes, 3, _lib.stream_ptr()),
                   "sed_crnn_backward")
        grads = []
        for i, (o0, o1, shp) in enumerate(module._layout):
            grads.append(gflat[o0:o1].view(shp) if ctx.needs_input_grad[4 + i] else None)
        return (None, None, None, None) + tuple(grads)


class CRNN(nn.Module):
    """Drop-in for baseline/models/CRNN.py:10-84 on MI355X (see module docstring for scope)."""

    def __init__(self, n_in_channel, nclass, attention=False, activation="Relu", dropout=0, train_cnn=True,
                 rnn_type='BGRU', n_RNN_cell=64, n_layers_RNN=1, dropout_recurrent=0, **kwargs):
        super().__init__()
        mfma_dtype = kwargs.pop("mfma_dtype", "f32")
        if rnn_type != 'BGRU':
            raise NotImplementedError("Only BGRU supported for CRNN for now")          # (the reference's own error, CRNN.py:26-27)
        self.attention = attention
        self.cnn = CNN(n_in_channel, activation, dropout, **kwargs)
        # hot_path: the configuration the HIP kernels implement (cfg.crnn_kwargs and BASELINE.json configs[4]'s widths).  Every
        # other combination the reference's constructor accepts - activation "Relu" (the class default) / "leakyrelu" / "cg",
        # attention=False (weak = strong.mean(1), CRNN.py:82-83), other filter counts, kernels, poolings, cell counts, classes,
        # recurrent dropout - is accepted too and runs the reference's graph on STOCK TORCH operators; forward
        # warns once.  Nothing on the hot path ever takes that route: a hot-path module still raises on a CPU tensor or a
        # missing library, and MeanTeacherStep / the HIP inference path refuse a module that is not hot_path.
        self.hot_path = bool(self.cnn.hot and attention and n_RNN_cell in (64, 256) and n_layers_RNN in (1, 2) and
                             dropout_recurrent == 0 and 1 <= nclass <= 16)
        self._warned_stock = False
        if not train_cnn:
            for param in self.cnn.parameters():
                param.requires_grad = False
        self.train_cnn = train_cnn
        self.rnn = BidirectionalGRU(self.cnn.nb_filters[-1], n_RNN_cell, dropout=dropout_recurrent,
                                    num_layers=n_layers_RNN)
        self.dropout = nn.Dropout(dropout)
        self.dense = nn.Linear(n_RNN_cell * 2, nclass)
        self.sigmoid = nn.Sigmoid()
        if self.attention:                             # (CRNN.py:29-31: the attention layer only exists with attention=True)
            self.dense_softmax = nn.Linear(n_RNN_cell * 2, nclass)
            self.softmax = nn.Softmax(dim=-1)
        self._nclass, self._n_layers, self._p_drop = nclass, n_layers_RNN, float(dropout)
        self._C, self._H = int(self.cnn.nb_filters[-1]), int(n_RNN_cell)
        self._dtype = _lib.DTYPE_F32
        self.set_mfma_dtype(mfma_dtype)
        # flat storage (built lazily on the first forward / after every .to()/.cuda())
        self._flat = None
        self._bn_flat = None
        self._bn_tracked = None
        self._layout = None
        self._flat_gen = 0
        self._last_ctx = None
        self._seed_base = None
        self._seed_calls = 0

    # ---- reference API: checkpoint format (CRNN.py:33-57) ---------------------------------------
    def load_cnn(self, parameters):
        self.cnn.load(parameters)
        if not self.train_cnn:
            for param in self.cnn.parameters():
                param.requires_grad = False

    def load(self, filename=None, parameters=None):
        if filename is not None:
            parameters = torch.load(filename)
        if parameters is None:
            raise NotImplementedError("load is a filename or a list of parameters (state_dict)")
        self.cnn.load(parameters=parameters["cnn"])
        self.rnn.load_state_dict(parameters["rnn"])
        self.dense.load_state_dict(parameters["dense"])
        if "dense_softmax" in parameters and self.attention:      # superset key: the reference drops the attention layer
            self.dense_softmax.load_state_dict(parameters["dense_softmax"])

    def state_dict(self, destination=None, prefix='', keep_vars=False):
        sd = {"cnn": self.cnn.state_dict(destination=destination, prefix=prefix, keep_vars=keep_vars),
              "rnn": self.rnn.state_dict(destination=destination, prefix=prefix, keep_vars=keep_vars),
              "dense": self.dense.state_dict(destination=destination, prefix=prefix, keep_vars=keep_vars)}
        if self.attention:
            sd["dense_softmax"] = self.dense_softmax.state_dict(destination=destination, prefix=prefix, keep_vars=keep_vars)
        return sd

    def save(self, filename):
        torch.save({k: v for k, v in self.state_dict().items()}, filename)

    def set_mfma_dtype(self, name):
        """sed_dims.dtype: "f32" (exact fp32 MFMA), "bf16" (bf16 operands / fp32 accumulation in the conv-block GEMMs) or
        "bf16x3" (split bf16 operands, three MFMAs per product: ~2^-16 relative error, holds the 1e-3 posterior bound)."""
        if name not in _lib.DTYPES:
            raise ValueError(f"mfma_dtype must be one of {sorted(_lib.DTYPES)}, got {name!r}")
        self._dtype = _lib.DTYPES[name]
        return self

    # ---- flat storage -----------------------------------------------------------------------------
    def make_dims(self, B, T, F=64, p_drop=None):
        if not self.hot_path:
            raise _lib.SedError("this CRNN configuration is outside the MI355X hot path (CRNN.hot_path is False): it runs on stock "
                                "torch operators through forward() only - no flat buffers, no fused step, no HIP inference")
        return _lib.make_dims(B, T, F, self._C, self._H, self._nclass, self._n_layers,
                              self._p_drop if p_drop is None else p_drop, 1e-3, 0.99, self._dtype)

    def _dims(self, x):
        B, _, T, F = x.shape
        return self.make_dims(B, T, F)

    def _bn_modules(self):
        return [getattr(self.cnn.cnn, f"batchnorm{i}") for i in range(3)]

    def flatten_parameters_(self, device=None):
        """Pack every parameter / BN buffer into the flat buffers the kernels read (no-op when already
        packed on ``device``).  Parameters become views of ``self._flat`` in named_parameters() order,
        which is the order sed_param_layout defines."""
        params = list(self.parameters())
        device = device if device is not None else params[0].device
        if self._layout is None:
            offs = _lib.param_layout(self.make_dims(1, 16, 64, 0.0))
            assert len(offs) == len(params) + 1, "parameter order does not match sed_param_layout"
            self._layout = [(offs[i], offs[i + 1], tuple(p.shape)) for i, p in enumerate(params)]
            for (o0, o1, shp), p in zip(self._layout, params):
                assert o1 - o0 == p.numel(), "parameter shape does not match sed_param_layout"
        ok = self._flat is not None and self._flat.device == device
        if ok:
            base = self._flat.data_ptr()
            for (o0, _, _), p in zip(self._layout, params):
                if p.data_ptr() != base + 4 * o0 or p.device != device or p.dtype != torch.float32:
                    ok = False
                    break
            bns = self._bn_modules()
            Cn = self._C
            for i, bn in enumerate(bns):
                if bn.running_mean.data_ptr() != self._bn_flat.data_ptr() + 4 * (2 * i) * Cn or \
                        bn.running_var.data_ptr() != self._bn_flat.data_ptr() + 4 * (2 * i + 1) * Cn or \
                        bn.num_batches_tracked.data_ptr() != self._bn_tracked.data_ptr() + 8 * i:
                    ok = False
                    break
        if ok:
            return
        with torch.no_grad():
            total = self._layout[-1][1]
            flat = torch.empty(total, device=device, dtype=torch.float32)
            for (o0, o1, shp), p in zip(self._layout, params):
                flat[o0:o1].copy_(p.data.reshape(-1).to(device=device, dtype=torch.float32))
                p.data = flat[o0:o1].view(shp)
                p.grad = None
            Cn = self._C
            bn_flat = torch.empty(3 * 2 * Cn, device=device, dtype=torch.float32)
            trk = torch.empty(3, device=device, dtype=torch.int64)
            for i, bn in enumerate(self._bn_modules()):
                bn_flat[(2 * i) * Cn:(2 * i + 1) * Cn].copy_(bn.running_mean.to(device))
                bn_flat[(2 * i + 1) * Cn:(2 * i + 2) * Cn].copy_(bn.running_var.to(device))
                trk[i] = bn.num_batches_tracked.to(device)
                bn._buffers["running_mean"] = bn_flat[(2 * i) * Cn:(2 * i + 1) * Cn]
                bn._buffers["running_var"] = bn_flat[(2 * i + 1) * Cn:(2 * i + 2) * Cn]
                bn._buffers["num_batches_tracked"] = trk[i]
            self._flat, self._bn_flat, self._bn_tracked = flat, bn_flat, trk
            self._flat_gen += 1

    def _next_seed(self, device):
        """Philox key for the dropout masks of one train-mode forward (stand-in for torch's global RNG
        used by nn.Dropout, CNN.py:59-61 / CRNN.py:74): drawn once from torch's CPU generator, so
        torch.manual_seed() makes runs repeatable, then advanced per call."""
        if self._seed_base is None:
            self._seed_base = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        self._seed_calls += 1
        s = (self._seed_base + 0x9E3779B97F4A7C15 * self._seed_calls) & 0x7FFFFFFFFFFFFFFF
        return torch.tensor([s], dtype=torch.int64, device=device)

    # ---- forward (CRNN.py:59-84) -------------------------------------------------------------------
    def forward(self, x, seed=None):
        """x: float32 [B, 1, T, 64] on the GPU -> (strong [B, T//8, nclass], weak [B, nclass]).
        ``seed`` (int64 device tensor [1]) pins the dropout stream for tests."""
        if not self.hot_path:
            return self._stock_forward(x)
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError(f"expected input [B, 1, T, F], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise _lib.SedError("CRNN.forward needs a GPU tensor: the HIP path is the only path (no CPU fallback)")
        if x.shape[3] // 64 != 1 or x.shape[3] != 64:
            warnings.warn("Output shape is: {}".format((x.shape[0], x.shape[2] // 8, 64 * (x.shape[3] // 64))))
            raise NotImplementedError("hot path implements freq == 1 after pooling, i.e. 64 mel bins (CRNN.py:64-70)")
        x = x.contiguous().float()
        self.flatten_parameters_(x.device)
        use_drop = self.training and self._p_drop > 0
        seed_t = (seed if seed is not None else self._next_seed(x.device)) if use_drop else None
        params = list(self.parameters())
        strong, weak = _CRNNFunction.apply(self, x, self.training, seed_t, *params)
        return strong, weak

    def _stock_forward(self, x):
        """baseline/models/CRNN.py:59-84 for a configuration OUTSIDE the hot path, on stock torch operators (whatever device the
        module and x live on).  Same graph, same warning for freq != 1 (which the reference then feeds, chan * freq wide, into a
        GRU built for chan features - torch's size error is the reference's behaviour too)."""
        if not self._warned_stock:
            self._warned_stock = True
            warnings.warn("CRNN: this configuration is outside the MI355X hot path (activation='glu', attention=True, 3 x 64 / 128 "
                          "filters, pooling (2,4) x 3, n_RNN_cell 64 / 256): running the reference graph on stock torch operators")
        x = self.cnn(x)
        bs, chan, frames, freq = x.size()
        if freq != 1:
            warnings.warn("Output shape is: {}".format((bs, frames, chan * freq)))
            x = x.permute(0, 2, 1, 3).contiguous().view(bs, frames, chan * freq)
        else:
            x = x.squeeze(-1).permute(0, 2, 1)
        x = self.dropout(self.rnn(x))
        strong = self.sigmoid(self.dense(x))
        if self.attention:
            sof = torch.clamp(self.softmax(self.dense_softmax(x)), min=1e-7, max=1)
            weak = (strong * sof).sum(1) / sof.sum(1)
        else:
            weak = strong.mean(1)
        return strong, weak

    def check_recurrence(self):
        """Raise if the wide model's cluster recurrence of the LAST forward timed out on a cross-workgroup wait (it then
        used a stale hidden state).  Eval / debugging aid for the drop-in path; MeanTeacherStep.check_health is the
        training-loop counterpart."""
        if self._H == 256 and self._last_ctx is not None:
            n = int(self.ctx_view("gru_err").view(torch.int32)[0].item())
            if n:
                raise _lib.SedError(f"cluster GRU recurrence: {n} cross-workgroup waits timed out; the outputs are invalid")

    def ctx_view(self, name):
        """Test hook: float32 (or float64 for 'mom0'/'stat*') view of an intermediate of the last forward."""
        cbuf, dims = self._last_ctx
        off, nb = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.lib().sed_crnn_ctx_view(C.byref(dims), name.encode(), C.byref(off), C.byref(nb)), "sed_crnn_ctx_view")
        raw = cbuf[off.value:off.value + nb.value]
        if name in ("mom0", "stat1", "stat2"):
            return raw.view(torch.float64)
        if dims.dtype in (_lib.DTYPE_BF16, _lib.DTYPE_F16) and name in ("p0", "y1", "p1", "y2"):
            # SED_DTYPE_BF16 stores the conv-block activations as bf16; SED_DTYPE_F16's views are the bf16 COPIES the backward
            # reads (a training forward writes them next to the fp16 tensors of the forward chain)
            return raw.view(torch.bfloat16).float()
        return raw.view(torch.float32)
