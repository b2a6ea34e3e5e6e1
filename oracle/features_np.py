"""TEST INFRASTRUCTURE ONLY - numpy restatement of the reference's feature path.

Follows DatasetDcase2019Task4.calculate_mel_spec (baseline/DatasetDcase2019Task4.py:197-231) and the
per-sample transform chain fixed by get_transforms (baseline/utils/utils.py:397-412):
AugmentGaussianNoise -> ApplyLog -> PadOrTrunc -> ToTensor -> Normalize
(baseline/DataLoad.py:262-287, 189-207, 210-259, 290-321, 324-350) and Scaler
(baseline/utils/Scaler.py:34-105).

PARITY UNPINNED for the librosa calls (stft / filters.mel / amplitude_to_db): librosa is an
unpinned, un-vendored dependency of the reference (environment.yml:17, README.md:29 ">= 0.6.3")
and is absent here.  Those three functions restate librosa's published algorithm (0.6-0.10 agree on
it for these arguments) and are cross-checked in tests against torch.stft / numpy.fft only.
Everything else in this file is pinned against the imported reference (oracle/gen_golden.py).
"""
import numpy as np


def hamming_window(n):
    """np.hamming(n) (DatasetDcase2019Task4.py:209): symmetric, 0.54 - 0.46 cos(2 pi k/(n-1))."""
    k = np.arange(n, dtype=np.float64)
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * k / (n - 1))


def n_frames(n_samples, hop):
    return 1 + n_samples // hop


def stft_mag(y, n_fft, hop, window):
    """|librosa.stft(y, n_fft, hop_length=hop, window=window, center=True, pad_mode='reflect')|
    (DatasetDcase2019Task4.py:211-218,221).  Returns float64 [1 + n_fft//2, 1 + len(y)//hop]."""
    y = np.asarray(y, dtype=np.float64)
    ypad = np.pad(y, n_fft // 2, mode="reflect")
    nf = 1 + (len(ypad) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(nf)[:, None]
    frames = ypad[idx] * window[None, :]
    return np.abs(np.fft.rfft(frames, axis=1)).T


def hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm=None)
    (called inside melspectrogram, DatasetDcase2019Task4.py:220-225). float32 [n_mels, 1+n_fft//2]."""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz_slaney(np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    return w.astype(np.float32)


def amplitude_to_db(S, amin=1e-5, top_db=80.0):
    """librosa.amplitude_to_db(S) with defaults ref=1.0, amin=1e-5, top_db=80
    (DataLoad.py:206, DatasetDcase2019Task4.py:228): 10 log10(max(amin^2, S^2)), clamped to
    >= max - top_db over the whole array passed in."""
    mag = np.abs(np.asarray(S))
    power = np.square(mag)
    log_spec = 10.0 * np.log10(np.maximum(amin ** 2, power))
    log_spec = log_spec - 10.0 * np.log10(np.maximum(amin ** 2, 1.0))
    return np.maximum(log_spec, log_spec.max() - top_db)


def calculate_mel_spec(audio, sr, n_window, hop_length, n_mels, f_min, f_max, save_log_feature=False):
    """DatasetDcase2019Task4.calculate_mel_spec (:197-231). Returns float32 [frames, n_mels]."""
    win = hamming_window(n_window)
    mag = stft_mag(audio, n_window, hop_length, win)
    mel = mel_filterbank(sr, n_window, n_mels, f_min, f_max).astype(np.float64) @ mag
    if save_log_feature:
        mel = amplitude_to_db(mel)
    return mel.T.astype(np.float32)


def pad_trunc_seq(x, max_len):
    """DataLoad.pad_trunc_seq (:210-230): zero-pad (after the log!) or truncate along axis 0."""
    length = len(x)
    if length < max_len:
        pad = np.zeros((max_len - length,) + x.shape[1:])
        return np.concatenate((x, pad), axis=0)
    if length > max_len:
        return x[0:max_len]
    return x


def scaler_stats(samples):
    """Scaler.means + calculate_scaler (Scaler.py:34-97): per-last-axis mean / mean-of-square in
    float64 averaged over samples of identical shape; std = sqrt(E[x^2] - E[x]^2)."""
    mean = None
    msq = None
    cnt = 0
    for s in samples:
        a = np.asarray(s)
        m = a
        q = a ** 2
        while m.ndim != 1:
            m = np.mean(m, axis=0, dtype=np.float64)
            q = np.mean(q, axis=0, dtype=np.float64)
        mean = m if mean is None else mean + m
        msq = q if msq is None else msq + q
        cnt += 1
    mean = mean / cnt
    msq = msq / cnt
    return mean, msq, np.sqrt(msq - mean ** 2)


def transform_chain(mel_lin, max_frames, mean=None, std=None, noise=None):
    """get_transforms(frames, scaler, augment_type="noise" if noise is not None) applied to one
    linear-mel clip [frames, n_mels] (utils.py:397-412).

    noise: the |N(0, 0.25)| draw of AugmentGaussianNoise (DataLoad.py:283-285) made explicit, or None.
    Returns float32 [1, max_frames, n_mels] (student) and, with noise, the teacher copy as well."""
    def one(a):
        a = amplitude_to_db(a.T).T                                  # ApplyLog  DataLoad.py:206
        a = pad_trunc_seq(a, max_frames)                             # PadOrTrunc
        a = a.astype(np.float32)[None]                               # ToTensor(.float(), unsqueeze 0)
        if mean is not None:
            a = ((a - mean) / std).astype(np.float32)                # Normalize -> Scaler.normalize (f64 math, torch.Tensor() -> f32)
        return a
    if noise is None:
        return one(np.asarray(mel_lin))
    clean = np.asarray(mel_lin)
    noisy = clean + noise                                            # DataLoad.py:285
    return one(clean), one(noisy)
