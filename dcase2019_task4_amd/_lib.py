"""ctypes binding of libdcase_sed_mi355.so (the C-ABI declared in include/dcase_sed.h).

There is NO fallback: if the library is missing or a call fails, this raises.  The product path is
the HIP path or nothing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdcase_sed_mi355.so")
# SED_LIB: a variant build of the SAME sources with experiment knobs (tools/build_variant.sh -> build/variants/, never shipped).
# Such builds may compute wrong results on purpose (timing experiments), so they load only with SED_ALLOW_VARIANT=1.
if os.environ.get("SED_LIB"):
    if os.environ.get("SED_ALLOW_VARIANT") != "1":
        raise ImportError("SED_LIB is set but SED_ALLOW_VARIANT != 1: refusing to load a variant build of the library")
    LIB_PATH = os.environ["SED_LIB"]


class SedError(RuntimeError):
    pass


class SedDims(C.Structure):
    """sed_dims (include/dcase_sed.h) - mirrors cfg.crnn_kwargs, baseline/config.py:53-58."""
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("F", C.c_int32), ("C", C.c_int32), ("H", C.c_int32),
                ("nclass", C.c_int32), ("n_layers_rnn", C.c_int32), ("p_drop", C.c_float),
                ("bn_eps", C.c_float), ("bn_momentum", C.c_float), ("dtype", C.c_int32)]


FORK_CALLBACK = C.CFUNCTYPE(None, C.c_void_p)     # void (*fn)(void* user): sed_crnn_fork_callback
DTYPE_F32, DTYPE_BF16, DTYPE_BF16X3, DTYPE_F16 = 0, 1, 2, 3
FFT_F64, FFT_F32 = 0, 1
FFT_DTYPES = {"f64": FFT_F64, "float64": FFT_F64, "f32": FFT_F32, "float32": FFT_F32}
DTYPES = {"f32": DTYPE_F32, "fp32": DTYPE_F32, "float32": DTYPE_F32, "bf16": DTYPE_BF16, "bfloat16": DTYPE_BF16,
          "bf16x3": DTYPE_BF16X3, "f16": DTYPE_F16, "fp16": DTYPE_F16, "float16": DTYPE_F16}


class SedStepState(C.Structure):
    """sed_step_state (include/dcase_sed.h); lives in device memory, this is its host mirror."""
    _fields_ = [("global_step", C.c_int64), ("opt_step", C.c_int64), ("rampup_length", C.c_int64),
                ("base_seed", C.c_uint64), ("seed_student", C.c_uint64), ("seed_teacher", C.c_uint64),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("ema_decay", C.c_double), ("max_cons_cost", C.c_double),
                ("cons_weight", C.c_float), ("ema_alpha", C.c_float), ("adam_step_size", C.c_float),
                ("adam_sqrt_bc2", C.c_float)]


_P = C.c_void_p
_SIGS = {
    "sed_last_error": (C.c_char_p, []),
    "sed_resample": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                               C.c_int, C.c_void_p]),
    "sed_scaler_stats": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]),
    "sed_postprocess": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_void_p]),
    "sed_version": (C.c_int, []),
    "sed_param_count": (C.c_int, [C.POINTER(SedDims)]),
    "sed_param_layout": (C.c_int, [C.POINTER(SedDims), C.POINTER(C.c_int64)]),
    "sed_crnn_ctx_bytes": (C.c_size_t, [C.POINTER(SedDims)]),
    "sed_crnn_bwd_ws_bytes": (C.c_size_t, [C.POINTER(SedDims)]),
    "sed_crnn_forward": (C.c_int, [C.POINTER(SedDims), _P, _P, _P, _P, C.c_int, C.c_int, _P, _P, C.c_size_t, _P, _P, _P]),
    "sed_crnn_moments": (C.c_int, [C.POINTER(SedDims), _P, _P, C.c_size_t, _P]),
    "sed_crnn_backward": (C.c_int, [C.POINTER(SedDims), _P, _P, _P, _P, C.c_size_t, _P, _P, _P, _P, C.c_size_t, C.c_int, _P]),
    "sed_crnn_buffers_init": (C.c_int, [C.POINTER(SedDims), _P, C.c_size_t, _P, C.c_size_t, _P]),
    "sed_crnn_ctx_view": (C.c_int, [C.POINTER(SedDims), C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "sed_mt_loss": (C.c_int, [C.POINTER(SedDims), _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "sed_mt_loss_backward": (C.c_int, [C.POINTER(SedDims), _P, _P, _P, _P, C.c_size_t, _P, _P, _P, C.c_int, C.c_int, C.c_int,
                                       C.c_int, _P, C.c_int, _P, _P, _P, _P, _P, C.c_size_t, C.c_int, _P]),
    "sed_mt_step_backward": (C.c_int, [C.POINTER(SedDims), _P, _P, _P, _P, C.c_size_t, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int,
                                       C.c_int, _P, C.c_int, _P, _P, _P, _P, _P, C.c_size_t, C.c_int, _P]),
    "sed_p2p_buffer_bytes": (C.c_size_t, [C.c_longlong]),
    "sed_p2p_alloc": (C.c_int, [C.c_size_t, C.c_int, C.POINTER(C.c_void_p), _P, C.POINTER(C.c_int)]),
    "sed_p2p_open": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "sed_p2p_close": (C.c_int, [_P]),
    "sed_p2p_free": (C.c_int, [_P]),
    "sed_p2p_can_access": (C.c_int, [C.c_int]),
    "sed_p2p_errors": (C.c_int, [_P, C.POINTER(C.c_uint)]),
    "sed_p2p_configure": (C.c_int, [_P, C.c_double, _P, C.c_int]),
    "sed_p2p_allreduce": (C.c_int, [_P, C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_longlong, C.c_int, _P]),
    "sed_adam_ema": (C.c_int, [C.c_int64, _P, _P, _P, _P, _P, _P, C.c_float, _P]),
    "sed_ema_update": (C.c_int, [C.c_int64, _P, _P, C.c_float, _P]),
    "sed_step_state_init": (C.c_int, [_P, C.c_uint64, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_double, C.c_double, _P]),
    "sed_step_state_advance": (C.c_int, [_P, _P]),
    "sed_step_state_update": (C.c_int, [_P, C.c_uint64, C.c_double, C.c_int, _P]),
    "sed_step_state_set_global_step": (C.c_int, [_P, C.c_int64, _P]),
    "sed_stream_prepare": (C.c_int, [_P]),
    "sed_stream_release": (C.c_int, [_P]),
    "sed_crnn_fork_callback": (C.c_int, [_P, _P, _P]),
    "sed_mel_spec_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sed_mel_spec": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, C.c_size_t, _P]),
    "sed_mel_tables": (C.c_int, [C.c_int, _P, _P, C.c_int, _P, C.c_size_t, _P]),
    "sed_mel_frames": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P, C.c_size_t, C.c_int, C.c_int, _P]),
    "sed_seed_advance": (C.c_int, [_P, _P]),
    "sed_logmel_transform_ws_bytes": (C.c_size_t, [C.c_int]),
    "sed_logmel_transform": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_size_t, C.c_int, _P]),
    "sed_selftest": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "sed_debug_set": (C.c_int, [C.c_int]),
    "sed_build_flags": (C.c_int, []),
    "sed_kernel_replay": (C.c_int, [C.c_char_p, C.POINTER(SedDims), _P, _P, _P, _P, C.c_size_t, _P, _P, C.c_size_t, _P]),
}

_lib = None


def exported_symbols():
    """Names every build must export (checked on CPU by tests/test_abi.py)."""
    return sorted(_SIGS)


def lib():
    """Load (once) and return the library; raises SedError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SedError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C dcase2019_task4_amd/csrc` (needs hipcc). There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
        if os.environ.get("SED_DEBUG"):          # timing experiments only (see sed_debug_set)
            l.sed_debug_set(int(os.environ["SED_DEBUG"]))
    return _lib


def scratch(nbytes, device):
    """A caller-owned scratch / context buffer for the library.  SED_POISON=1 (the -m gpu test session sets it) fills it with
    0xFF bytes first - NaN in fp32, bf16 and fp16 alike - so that any kernel that reads a byte no kernel wrote shows up as a NaN
    instead of depending on what the allocator's block held before (round 5: the bf16 copy of an unpooled row)."""
    import torch
    t = torch.empty(nbytes, device=device, dtype=torch.uint8)
    if os.environ.get("SED_POISON") == "1" and nbytes:
        t.fill_(0xFF)
    return t


def check(status, what):
    if status != 0:
        msg = lib().sed_last_error()
        raise SedError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device (or host) pointer of a torch tensor / None as c_void_p."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_dims(B, T, F=64, C_=64, H=64, nclass=10, n_layers_rnn=2, p_drop=0.5, bn_eps=1e-3, bn_momentum=0.99, dtype=DTYPE_F32):
    return SedDims(B, T, F, C_, H, nclass, n_layers_rnn, p_drop, bn_eps, bn_momentum, dtype)


def param_layout(dims):
    l = lib()
    n = l.sed_param_count(C.byref(dims))
    if n <= 0:
        raise SedError(f"sed_param_count: {l.sed_last_error().decode()}")
    off = (C.c_int64 * (n + 1))()
    check(l.sed_param_layout(C.byref(dims), off), "sed_param_layout")
    return list(off)
