"""Audio decode + resample in front of the feature extractor: the reference's ``read_audio``
(baseline/utils/utils.py:175-193) with the same signature and return value.

``soundfile`` and ``librosa`` are not dependencies here: RIFF/WAVE files are parsed with numpy (PCM 8/16/24/32-bit and
IEEE float 32/64, any channel count, scaled to [-1, 1) as libsndfile does) and the resampling - librosa.resample's
``kaiser_best`` = resampy's windowed-sinc interpolation - runs on the GPU (``sed_resample``).  The filter table is
built on the host once per rate pair, from resampy's published construction (see oracle/resample_np.py for the
restatement it is tested against; parity with librosa itself is unpinned: the package is absent from this image).
"""
import math
import struct

import numpy as np
import scipy.signal
import torch

from . import _lib

_KAISER_BEST = dict(num_zeros=64, precision=9, rolloff=0.9475937167399596, beta=14.769656459379492)


def _sinc_window(num_zeros, precision, rolloff, beta):
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = scipy.signal.windows.kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


class Resampler:
    """librosa.resample(y, orig_sr, target_sr) for batches of equal-length clips, on the GPU, in float64."""

    def __init__(self, orig_sr, target_sr, device="cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SedError("Resampler needs a GPU device (no CPU fallback)")
        self.orig_sr, self.target_sr = int(orig_sr), int(target_sr)
        self.ratio = float(target_sr) / float(orig_sr)
        win, self.num_table = _sinc_window(**_KAISER_BEST)
        if self.ratio < 1:
            win = win * self.ratio
        self.win = torch.tensor(win, dtype=torch.float64, device=self.device)
        self._treg = {}

    def n_out(self, n_in):
        return int(math.ceil(n_in * self.ratio))

    def __call__(self, waves):
        """waves: [n_clips, n_in] or [n_in] (numpy / tensor) -> float64 cuda tensor [n_clips, ceil(n_in * ratio)]."""
        x = torch.as_tensor(waves).to(self.device, torch.float64).contiguous()
        squeeze = x.dim() == 1
        if squeeze:
            x = x[None]
        n, n_in = x.shape
        n_out = self.n_out(n_in)
        y = torch.empty(n, n_out, dtype=torch.float64, device=self.device)
        if n_in not in self._treg:
            # resampy's time_register: a running sum, rounded step by step (numpy.cumsum accumulates sequentially)
            n_res = int(n_in * self.ratio)
            tr = np.cumsum(np.r_[0.0, np.full(max(n_res - 1, 0), 1.0 / self.ratio)])
            self._treg[n_in] = torch.tensor(tr, dtype=torch.float64, device=self.device)
        _lib.check(_lib.lib().sed_resample(_lib.ptr(x), n, n_in, self.ratio, _lib.ptr(self.win), self.win.numel(),
                                           self.num_table, _lib.ptr(self._treg[n_in]), _lib.ptr(y), n_out,
                                           _lib.stream_ptr()), "sed_resample")
        return y[0] if squeeze else y


def read_wav(path):
    """(float64 array [n] or [n, channels], sample rate), values scaled like soundfile.read's default dtype."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
            if fmt[0] == 0xFFFE and len(body) >= 26:            # WAVE_FORMAT_EXTENSIBLE: real tag in the sub-format GUID
                fmt = (struct.unpack("<H", body[24:26])[0],) + fmt[1:]
        elif cid == b"data":
            payload = body
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    tag, ch, fs, _, _, bits = fmt
    if tag == 1:
        if bits == 8:
            a = (np.frombuffer(payload, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
        elif bits == 16:
            a = np.frombuffer(payload, dtype="<i2").astype(np.float64) / 32768.0
        elif bits == 24:
            b = np.frombuffer(payload[:len(payload) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            a = (v - ((v & 0x800000) << 1)).astype(np.float64) / 8388608.0
        elif bits == 32:
            a = np.frombuffer(payload, dtype="<i4").astype(np.float64) / 2147483648.0
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif tag == 3:
        a = np.frombuffer(payload, dtype="<f4" if bits == 32 else "<f8").astype(np.float64)
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag}")
    a = a[:len(a) // ch * ch]
    return (a.reshape(-1, ch) if ch > 1 else a), fs


_resamplers = {}


def read_audio(path, target_fs=None):
    """baseline/utils/utils.py:175-193: (mono float64 numpy array at target_fs, sampling rate)."""
    audio, fs = read_wav(path)
    if audio.ndim > 1:
        audio = np.mean(audio, axis=1)
    if target_fs is not None and fs != target_fs:
        key = (fs, target_fs)
        if key not in _resamplers:
            _resamplers[key] = Resampler(fs, target_fs)
        audio = _resamplers[key](audio).cpu().numpy()
        fs = target_fs
    return audio, fs
