"""-m gpu tests of the audio decode + resample step: sed_resample against the numpy restatement
of resampy's kaiser_best interpolation (oracle/resample_np.py; parity with librosa itself is unpinned - the package is
absent), fp64 on both sides, tolerance 1e-12 of the signal amplitude; read_audio against WAV files written here."""
import struct

import numpy as np
import pytest
import torch

from oracle import resample_np, synth

pytestmark = pytest.mark.gpu


def _write_wav(path, data, fs, bits=16, fmt_tag=1):
    data = np.atleast_2d(data.T).T if data.ndim == 1 else data          # [n, ch]
    n, ch = data.shape
    if fmt_tag == 3:
        raw = data.astype("<f4").tobytes()
        bits = 32
    elif bits == 16:
        raw = np.clip(np.round(data * 32768.0), -32768, 32767).astype("<i2").tobytes()
    elif bits == 24:
        v = np.clip(np.round(data * 8388608.0), -8388608, 8388607).astype(np.int32).reshape(-1)
        raw = b"".join(int(x & 0xFFFFFF).to_bytes(3, "little") for x in v)
    else:
        raise ValueError(bits)
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(raw), b"WAVE", b"fmt ", 16, fmt_tag, ch, fs,
                      fs * ch * bits // 8, ch * bits // 8, bits, b"data", len(raw))
    with open(path, "wb") as f:
        f.write(hdr + raw)


@pytest.mark.parametrize("orig,target,n", [(44100, 16000, 8820), (48000, 16000, 4801), (16000, 44100, 1600),
                                           (22050, 16000, 3001), (44100, 16000, 37)])
def test_resample_vs_oracle(orig, target, n):
    from dcase2019_task4_amd.audio import Resampler
    waves = np.stack([synth.make_wave(i, n) * (1.0 + 0.3 * i) for i in range(3)])
    got = Resampler(orig, target)(waves).cpu().numpy()
    assert got.shape == (3, int(np.ceil(n * target / orig)))
    for i in range(3):
        want = resample_np.resample(waves[i], orig, target)
        assert want.shape == got[i].shape
        np.testing.assert_allclose(got[i], want, rtol=0, atol=1e-12 * max(1.0, np.abs(want).max()))


def test_resample_properties_at_full_clip_length():
    """Size-independent checks on a full 10-s 44.1 kHz clip (441 000 -> 160 000 samples): a 1 kHz tone keeps its
    frequency and (up to resampy's integer-step gain, 0.4 %) its amplitude; linearity; equal rates are the identity."""
    from dcase2019_task4_amd.audio import Resampler
    rs = Resampler(44100, 16000)
    t = np.arange(441000) / 44100.0
    a = np.sin(2 * np.pi * 1000.0 * t)
    b = 0.5 * np.sin(2 * np.pi * 3000.0 * t + 0.3)
    ya, yb, yab = (rs(v).cpu().numpy() for v in (a, b, a + 2.0 * b))
    assert ya.shape == (160000,)
    tt = np.arange(160000) / 16000.0
    core = slice(2000, 158000)
    gain = np.dot(ya[core], np.sin(2 * np.pi * 1000.0 * tt[core])) / np.dot(np.sin(2 * np.pi * 1000.0 * tt[core]),
                                                                            np.sin(2 * np.pi * 1000.0 * tt[core]))
    assert abs(gain - 1.0) < 6e-3
    resid = np.abs(ya[core] - gain * np.sin(2 * np.pi * 1000.0 * tt[core])).max()
    print('gain', gain, 'residual', resid)
    assert resid < 2e-3        # resampy's truncated table step leaves ~1e-4..1e-3 of interpolation error
    np.testing.assert_allclose(yab, ya + 2.0 * yb, atol=1e-12)
    tone9k = rs(np.sin(2 * np.pi * 9000.0 * t)).cpu().numpy()          # above the new Nyquist: rejected
    print('9 kHz leak', np.abs(tone9k[core]).max())
    assert np.abs(tone9k[core]).max() < 5e-3


def test_read_audio_matches_reference_semantics(tmp_path):
    """read_audio(path, target_fs): channel mean, then resample; (audio float64, fs) like utils.py:175-193."""
    from dcase2019_task4_amd.audio import read_audio, read_wav
    n = 22050
    st = np.stack([0.4 * synth.make_wave(1, n), 0.4 * synth.make_wave(2, n)], axis=1)
    p16, p24, pf = str(tmp_path / "a16.wav"), str(tmp_path / "a24.wav"), str(tmp_path / "af.wav")
    _write_wav(p16, st, 44100, 16)
    _write_wav(p24, st[:, 0], 44100, 24)
    _write_wav(pf, st, 44100, fmt_tag=3)
    raw, fs = read_wav(p16)
    assert fs == 44100 and raw.shape == (n, 2) and raw.dtype == np.float64
    np.testing.assert_array_equal(raw, np.clip(np.round(st * 32768.0), -32768, 32767) / 32768.0)
    np.testing.assert_array_equal(read_wav(p24)[0], np.round(st[:, 0] * 8388608.0) / 8388608.0)
    np.testing.assert_array_equal(read_wav(pf)[0], st.astype(np.float32).astype(np.float64))
    same, fs0 = read_audio(p16)                            # target_fs None: native rate, channels averaged
    assert fs0 == 44100
    np.testing.assert_array_equal(same, raw.mean(axis=1))
    audio, fs1 = read_audio(p16, 16000)
    assert fs1 == 16000 and audio.dtype == np.float64 and audio.shape == (8000,)
    np.testing.assert_allclose(audio, resample_np.resample(raw.mean(axis=1), 44100, 16000), atol=1e-12)
