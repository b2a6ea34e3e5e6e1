"""TEST INFRASTRUCTURE ONLY - numpy statement of the resampler behind the reference's read_audio.

baseline/utils/utils.py:175-193 calls ``librosa.resample(audio, orig_sr, target_sr)``; in the librosa versions the
reference targets (>= 0.6.3, README.md:29; unpinned in environment.yml:17) that is resampy's band-limited sinc
interpolation with the ``kaiser_best`` filter, followed by ``fix_length`` to ceil(n * ratio) samples.  Neither package
is in this image, so this file restates resampy's published algorithm (resampy/filters.py ``sinc_window``,
resampy/interpn.py ``resample_f``, resampy/core.py ``resample``) - **parity unpinned**; the kaiser_best parameters are
the ones resampy generated its shipped table with (num_zeros 64, precision 9, rolloff 0.9475937167399596, Kaiser beta
14.769656459379492).

Which librosa uses which resampler (from librosa's changelog; matters to anyone re-generating features today):
  * librosa 0.6.x - 0.9.x: ``resample(..., res_type='kaiser_best')`` is the default  -> this file.
  * librosa >= 0.10.0: the default became ``res_type='soxr_hq'`` (libsoxr); ``kaiser_best`` is still available on
    request (and needs resampy installed).  A reference checkout run against librosa >= 0.10 therefore resamples with
    a DIFFERENT filter than the one restated here; the reference's README ("librosa >= 0.6.3", 2019) predates 0.10.
The STFT / mel / dB restatements in features_np.py are unaffected by this version split.
"""
import numpy as np
import scipy.signal

KAISER_BEST = dict(num_zeros=64, precision=9, rolloff=0.9475937167399596, beta=14.769656459379492)


def sinc_window(num_zeros=64, precision=9, rolloff=0.945, beta=14.769656459379492):
    """Right half of a Kaiser-windowed sinc, 2**precision samples per zero crossing (resampy.filters.sinc_window)."""
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = scipy.signal.windows.kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


def filter_table(ratio):
    """(interp_win scaled for this ratio, num_table) as resampy.core.resample prepares them."""
    win, num_table = sinc_window(**KAISER_BEST)
    if ratio < 1:
        win = win * ratio
    return win, num_table


def resample_f(x, n_out, ratio, interp_win, num_table):
    """resampy.interpn.resample_f for a 1-D float64 signal, one output sample per loop trip."""
    interp_delta = np.zeros_like(interp_win)
    interp_delta[:-1] = np.diff(interp_win)
    scale = min(1.0, ratio)
    time_increment = 1.0 / ratio
    index_step = int(scale * num_table)
    nwin = interp_win.shape[0]
    n_orig = x.shape[0]
    y = np.zeros(n_out, dtype=np.float64)
    time_register = 0.0
    for t in range(n_out):
        n = int(time_register)
        frac = scale * (time_register - n)
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        i_max = min(n + 1, (nwin - offset) // index_step)
        i = np.arange(i_max)
        w = interp_win[offset + i * index_step] + eta * interp_delta[offset + i * index_step]
        acc = float(np.dot(w, x[n - i]))
        frac = scale - frac
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        k_max = min(n_orig - n - 1, (nwin - offset) // index_step)
        k = np.arange(k_max)
        w = interp_win[offset + k * index_step] + eta * interp_delta[offset + k * index_step]
        acc += float(np.dot(w, x[n + k + 1]))
        y[t] = acc
        time_register += time_increment
    return y


def resample(y, orig_sr, target_sr):
    """librosa.resample(y, orig_sr, target_sr) (res_type='kaiser_best', fix=True, scale=False)."""
    y = np.asarray(y, dtype=np.float64)
    if orig_sr == target_sr:
        return y
    ratio = float(target_sr) / orig_sr
    n_fixed = int(np.ceil(y.shape[-1] * ratio))
    n_out = int(y.shape[-1] * ratio)                       # resampy's own length
    win, num_table = filter_table(ratio)
    out = resample_f(y, n_out, ratio, win, num_table)
    if n_fixed > n_out:                                    # librosa.util.fix_length pads with zeros
        out = np.concatenate([out, np.zeros(n_fixed - n_out)])
    return out[:n_fixed]
