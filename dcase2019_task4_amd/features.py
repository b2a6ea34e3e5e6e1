"""Feature front-end on MI355X - host-side mirror of the reference's feature interfaces.

* ``FeatureExtractor.calculate_mel_spec(audio)`` keeps the signature of
  ``DatasetDcase2019Task4.calculate_mel_spec`` (baseline/DatasetDcase2019Task4.py:197-231):
  numpy waveform in, float32 ``[frames, n_mels]`` linear-mel out; ``calculate_mel_spec_batch`` is the
  same for a batch of clips resident on the GPU.  The STFT + mel projection run in
  ``sed_mel_spec`` (csrc/feat.hip).
* ``LogMelTransform`` is the batched GPU form of ``get_transforms(frames, scaler, augment_type)``
  (baseline/utils/utils.py:397-412): [teacher noise] -> amplitude_to_db -> pad/trunc -> channel
  axis -> normalise, in ``sed_logmel_transform``.
* ``Scaler`` mirrors baseline/utils/Scaler.py (statistics pass stays in numpy: it is a one-off
  reduction before training).

The mel filterbank and the Hamming window are built on the host at construction time exactly as
librosa / numpy do (float64 math, float32 filterbank); they are inputs of the kernel, not part of
the per-clip work.
"""
import ctypes as C
import json
import os
import math
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib


@dataclass
class FeatureConfig:
    """baseline/config.py:17-25 (defaults = the reference's 44.1 kHz setting)."""
    sample_rate: int = 44100
    n_window: int = 2048
    hop_length: int = 511
    n_mels: int = 64
    max_len_seconds: float = 10.0
    f_min: float = 0.0
    f_max: float = 22050.0

    @property
    def max_frames(self):
        return math.ceil(self.max_len_seconds * self.sample_rate / self.hop_length)

    @classmethod
    def baseline_16k(cls):
        """The 16 kHz / hop 255 setting behind BASELINE.json's 64-mel x 628-frame shape."""
        return cls(sample_rate=16000, n_window=2048, hop_length=255, n_mels=64, f_min=0.0, f_max=8000.0)


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3)
    with np.errstate(divide="ignore", invalid="ignore"):
        log = 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) / (np.log(6.4) / 27.0)
    return np.where(f >= 1000.0, log, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    lin = (200.0 / 3) * m
    log = 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0))
    return np.where(m >= 15.0, log, lin)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """Slaney-scale triangular filterbank, peak 1 (librosa.filters.mel(htk=False, norm=None)),
    float32 [n_mels, 1 + n_fft//2]."""
    freqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    pts = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    width = np.diff(pts)
    ramps = pts[:, None] - freqs[None, :]
    fb = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        fb[i] = np.maximum(0.0, np.minimum(-ramps[i] / width[i], ramps[i + 2] / width[i + 1]))
    return fb.astype(np.float32)


class FeatureExtractor:
    """GPU feature extractor with the reference's calculate_mel_spec call signature."""

    def __init__(self, cfg=None, device="cuda", save_log_feature=False, fft_dtype="f64"):
        self.cfg = cfg or FeatureConfig()
        if fft_dtype not in _lib.FFT_DTYPES:
            raise ValueError(f"fft_dtype must be one of {sorted(_lib.FFT_DTYPES)}")
        self.fft_dtype = fft_dtype
        if self.cfg.n_window != 2048:
            raise NotImplementedError("hot path implements n_window = 2048 (config.py:18)")
        if save_log_feature:
            raise NotImplementedError("main.py stores linear features (save_log_feature=False, main.py:199-201); "
                                      "the log is applied by LogMelTransform")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SedError("FeatureExtractor needs a GPU device (no CPU fallback)")
        c = self.cfg
        self.mel_basis = torch.tensor(mel_filterbank(c.sample_rate, c.n_window, c.n_mels, c.f_min, c.f_max), device=self.device)
        self.window = torch.tensor(np.hamming(c.n_window).astype(np.float32), device=self.device)
        self._tables = {}

    def n_frames(self, n_samples):
        return 1 + n_samples // self.cfg.hop_length

    def tables(self, exact_window=True):
        """The (window, mel_basis)-dependent tables of the STFT kernel, built once per extractor (sed_mel_tables) - the
        reference builds np.hamming / librosa.filters.mel once per call of calculate_mel_spec, on the host."""
        ws = self._tables.get(bool(exact_window))
        if ws is None:
            l = _lib.lib()
            c = self.cfg
            ws = torch.empty(l.sed_mel_spec_ws_bytes(1, 0, c.hop_length, c.n_window, c.n_mels), device=self.device, dtype=torch.uint8)
            # window=None -> the kernel builds np.hamming(2048) itself in float64 (what librosa multiplies by)
            _lib.check(l.sed_mel_tables(c.n_window, None if exact_window else _lib.ptr(self.window), _lib.ptr(self.mel_basis),
                                        c.n_mels, _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "sed_mel_tables")
            self._tables[bool(exact_window)] = ws
        return ws

    def calculate_mel_spec_batch(self, waves, exact_window=True, workgroups=0, fft_dtype=None):
        """waves: float tensor [n_clips, n_samples] (any device) -> float32 cuda [n_clips, frames, n_mels].
        ``fft_dtype``: "f64" (the reference's arithmetic, default) or "f32" (stated reduced-precision mode, SED_FFT_F32)."""
        l = _lib.lib()
        c = self.cfg
        waves = torch.as_tensor(waves).to(self.device, torch.float32).contiguous()
        if waves.dim() == 1:
            waves = waves[None]
        n, ns = waves.shape
        frames = self.n_frames(ns)
        ws = self.tables(exact_window)
        out = torch.empty(n, frames, c.n_mels, device=self.device, dtype=torch.float32)
        _lib.check(l.sed_mel_frames(_lib.ptr(waves), n, ns, c.hop_length, c.n_window, _lib.ptr(self.mel_basis), c.n_mels,
                                    _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.FFT_DTYPES[fft_dtype or self.fft_dtype],
                                    int(workgroups), _lib.stream_ptr()), "sed_mel_frames")
        return out

    def calculate_mel_spec(self, audio):
        """DatasetDcase2019Task4.calculate_mel_spec(audio) -> np.ndarray float32 [frames, n_mels]."""
        out = self.calculate_mel_spec_batch(torch.as_tensor(np.asarray(audio, dtype=np.float32))[None])
        return out[0].cpu().numpy()


class FeatureCache:
    """The reference's on-disk feature cache (DatasetDcase2019Task4.py:183-195, 233-269): one
    ``<feature_dir>/<splitext(filename)[0]>.npy`` per audio file holding float32 [frames, n_mels] (linear mel,
    save_log_feature=False), so that baseline/main.py consumes GPU-extracted features unchanged."""

    def __init__(self, feature_dir, extractor=None):
        self.feature_dir = feature_dir
        self.extractor = extractor
        os.makedirs(feature_dir, exist_ok=True)

    def path(self, filename):
        return os.path.join(self.feature_dir, os.path.splitext(filename)[0] + ".npy")

    def get_feature_file(self, filename):
        """DatasetDcase2019Task4.get_feature_file: the ``get_feature_file_func`` DataLoadDf expects."""
        return np.load(self.path(filename))

    def extract_features(self, filenames, waves, batch_size=64, overwrite=False):
        """Compute and store the features of ``waves`` (equal-length float arrays, one per filename) ``batch_size`` clips
        per launch; files that already exist are kept (DatasetDcase2019Task4.py:255), returns the number written."""
        if self.extractor is None:
            raise ValueError("FeatureCache.extract_features needs a FeatureExtractor")
        todo = [i for i, f in enumerate(filenames) if overwrite or not os.path.exists(self.path(f))]
        for i0 in range(0, len(todo), batch_size):
            idx = todo[i0:i0 + batch_size]
            mel = self.extractor.calculate_mel_spec_batch(torch.as_tensor(np.stack([np.asarray(waves[i], dtype=np.float32)
                                                                                    for i in idx]))).cpu().numpy()
            for k, i in enumerate(idx):
                np.save(self.path(filenames[i]), mel[k])
        return len(todo)


class Scaler:
    """baseline/utils/Scaler.py: per-mel mean / mean-of-square in float64 over a dataset."""

    def __init__(self):
        self.mean_ = None
        self.mean_of_square_ = None
        self.std_ = None

    def calculate_scaler(self, dataset):
        n = 0
        for sample in dataset:
            x = sample[0] if isinstance(sample, (tuple, list)) and len(sample) == 2 else sample
            a = x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
            m, q = a, a ** 2
            while m.ndim != 1:
                m = np.mean(m, axis=0, dtype=np.float64)
                q = np.mean(q, axis=0, dtype=np.float64)
            self.mean_ = m if self.mean_ is None else self.mean_ + m
            self.mean_of_square_ = q if self.mean_of_square_ is None else self.mean_of_square_ + q
            n += 1
        self.mean_ = self.mean_ / n
        self.mean_of_square_ = self.mean_of_square_ / n
        self.std_ = np.sqrt(self.mean_of_square_ - self.mean_ ** 2)
        return self.mean_, self.std_

    def calculate_scaler_device(self, batches):
        """The same statistics from device batches: ``batches`` yields cuda float32 tensors [..., n_mels] of equal
        clip shape (e.g. the [N, 1, T, 64] output of LogMelTransform(frames) without a scaler, as main.py:212,249
        feeds the reference's Scaler).  One streaming fp64 pass on the GPU instead of a host loop over the set."""
        l = _lib.lib()
        sums, rows, n_mels = None, 0, None
        for x in batches:
            if x.device.type != "cuda":
                raise _lib.SedError("calculate_scaler_device needs GPU tensors (no CPU fallback)")
            x = x.contiguous().float()
            if sums is None:
                n_mels = x.shape[-1]
                sums = torch.zeros(2 * n_mels, dtype=torch.float64, device=x.device)
            r = x.numel() // n_mels
            _lib.check(l.sed_scaler_stats(_lib.ptr(x), r, n_mels, _lib.ptr(sums), _lib.stream_ptr()), "sed_scaler_stats")
            rows += r
        sums = sums.cpu().numpy()
        self.mean_ = sums[:n_mels] / rows
        self.mean_of_square_ = sums[n_mels:] / rows
        self.std_ = np.sqrt(self.mean_of_square_ - self.mean_ ** 2)
        return self.mean_, self.std_

    def normalize(self, batch):
        if torch.is_tensor(batch):
            return torch.Tensor((batch.cpu().numpy() - self.mean_) / self.std_)
        return (batch - self.mean_) / self.std_

    def state_dict(self):
        return {"mean_": self.mean_.tolist(), "mean_of_square_": self.mean_of_square_.tolist()}

    def load_state_dict(self, sd):
        self.mean_ = np.array(sd["mean_"])
        self.mean_of_square_ = np.array(sd["mean_of_square_"])
        self.std_ = np.sqrt(self.mean_of_square_ - self.mean_ ** 2)

    def save(self, path):
        with open(path, "w") as f:
            json.dump(self.state_dict(), f)

    def load(self, path):
        with open(path) as f:
            self.load_state_dict(json.load(f))


class LogMelTransform:
    """Batched GPU form of get_transforms(frames, scaler, add_axis_conv=True, augment_type)."""

    def __init__(self, frames, scaler=None, augment_type=None, device="cuda", seed=0, math_dtype="f64"):
        self.math_dtype = _lib.FFT_DTYPES[math_dtype]            # "f32": the front-end's stated fp32 mode (SED_FFT_F32)
        if augment_type not in (None, "noise"):
            raise NotImplementedError("only augment_type='noise' exists (utils.py:404-406)")
        self.frames = int(frames)
        self.noise = augment_type == "noise"
        self.device = torch.device(device)
        self.mean = self.std = None
        if scaler is not None:
            self.mean = torch.tensor(np.asarray(scaler.mean_), dtype=torch.float64, device=self.device)
            self.std = torch.tensor(np.asarray(scaler.std_), dtype=torch.float64, device=self.device)
        self._seed = int(seed)
        self._calls = 0

    def __call__(self, mel, seed=None):
        """mel: float32 cuda [n_clips, frames, n_mels] linear mel -> clean [n,1,T,n_mels] (, noisy)."""
        l = _lib.lib()
        mel = mel.to(self.device, torch.float32).contiguous()
        n, fr, nm = mel.shape
        clean = torch.empty(n, 1, self.frames, nm, device=self.device, dtype=torch.float32)
        noisy = torch.empty_like(clean) if self.noise else None
        seed_t = None
        if self.noise:
            if seed is None:
                self._calls += 1
                seed = (self._seed * 0x9E3779B97F4A7C15 + self._calls) & 0x7FFFFFFFFFFFFFFF
            seed_t = torch.tensor([seed], dtype=torch.int64, device=self.device)
        ws = torch.empty(l.sed_logmel_transform_ws_bytes(n), device=self.device, dtype=torch.uint8)
        _lib.check(l.sed_logmel_transform(_lib.ptr(mel), n, fr, nm, self.frames, _lib.ptr(self.mean), _lib.ptr(self.std),
                                          _lib.ptr(seed_t), _lib.ptr(clean), _lib.ptr(noisy), _lib.ptr(ws), ws.numel(),
                                          self.math_dtype, _lib.stream_ptr()),
                   "sed_logmel_transform")
        return (clean, noisy) if self.noise else clean


class _SampleTransforms:
    """What get_transforms returns: a callable with the reference's per-sample protocol (DataLoad.py:157-186 Compose and the
    transforms it chains): ``sample`` is a tuple / list whose LAST element is the label, every other element a
    [frames, n_mels] feature array; the result is a list of tensors in the reference's order - ``[x, label]``, or
    ``[x, x_noisy, label]`` with augment_type="noise" (AugmentGaussianNoise returns (clean, noisy, label), DataLoad.py:262-287).
    The arithmetic runs in sed_logmel_transform on the GPU, the tensors come back on the host like the reference's (a
    torch DataLoader collates them; main.py moves the batch to the GPU).  One launch + one copy per sample: the drop-in form,
    not the fast one - LogMelTransform (batched) and WaveformFrontEnd are."""

    def __init__(self, frames, scaler, add_axis_conv, augment_type, device, seed):
        self.add_axis_conv = bool(add_axis_conv)
        self.tr = LogMelTransform(frames, scaler=scaler, augment_type=augment_type, device=device, seed=seed)
        self._pid = None                # the process that last checked where it runs (a fork / spawn changes it)

    @staticmethod
    def worker_seed(seed, worker_id):
        """The noise seed of DataLoader worker `worker_id`: splitmix64 of (seed, worker id + 1).  Every worker holds its own
        copy of this object, whose call counter restarts at the same value - with the bare seed all workers would draw the
        same noise for their n-th sample."""
        z = (int(seed) + 0x9E3779B97F4A7C15 * (int(worker_id) + 1)) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return (z ^ (z >> 31)) & 0x7FFFFFFFFFFFFFFF

    def _enter_worker(self):
        """The reference hands this callable to DataLoadDf and runs it inside DataLoader workers (config.py: num_workers = 12,
        fork).  The arithmetic here is a GPU kernel and there is no CPU path: a FORKED worker cannot use the parent's HIP
        context, so that case is refused with a message that says what to do instead of torch's re-initialisation error;
        a spawned worker works (its own context, one launch + one copy per sample) and gets a noise stream of its own."""
        info = torch.utils.data.get_worker_info()
        self._pid = os.getpid()
        if info is None:
            return
        if torch.cuda._is_in_bad_fork():
            raise _lib.SedError(
                "get_transforms(...) runs sed_logmel_transform on the GPU and was called in a forked DataLoader worker, which "
                "cannot use the parent's HIP context.  Use DataLoader(num_workers=0), or multiprocessing_context='spawn' "
                "(one context + one launch per sample and worker), or - the fast path - keep linear-mel features in the "
                "dataset and apply features.LogMelTransform / WaveformFrontEnd to whole batches on the training process.")
        # info.seed = the loader's per-EPOCH base seed + worker id (torch draws a new base seed for every iterator): folding it in
        # gives every epoch its own noise - a worker re-created each epoch (persistent_workers=False) restarts its call counter
        # and would otherwise replay the previous epoch's noise sample for sample.  Reproducible under torch.manual_seed.
        self.tr._seed = self.worker_seed(int(self.tr._seed) ^ (int(info.seed) & 0x7FFFFFFFFFFFFFFF), info.id)

    def __call__(self, sample):
        if self._pid != os.getpid():
            self._enter_worker()
        sample = list(sample)
        label = torch.from_numpy(np.asarray(sample[-1])).float()              # ToTensor: "even labels" (DataLoad.py:316)
        outs = []
        for feat in sample[:-1]:
            mel = torch.as_tensor(np.asarray(feat, dtype=np.float32))[None]
            res = self.tr(mel)
            outs.extend(res if isinstance(res, tuple) else (res,))
        outs = [o[0].cpu() if self.add_axis_conv else o[0, 0].cpu() for o in outs]
        return outs + [label]

    def __getstate__(self):
        # (pickled into spawned workers: the worker decides for itself where it runs)
        st = dict(self.__dict__)
        st["_pid"] = None
        return st


def get_transforms(frames, scaler=None, add_axis_conv=True, augment_type=None, device="cuda", seed=0):
    """utils.get_transforms(frames, scaler=None, add_axis_conv=True, augment_type=None) (baseline/utils/utils.py:397-412):
    noise -> log -> pad / truncate -> tensor (+ channel axis) -> normalise, as one callable to hand to DataLoadDf(transform=...).
    In-process (num_workers=0) or in SPAWNED DataLoader workers; a forked worker is refused with a clear error (no CPU path)."""
    return _SampleTransforms(frames, scaler, add_axis_conv, augment_type, device, seed)


class WaveformFrontEnd:
    """BASELINE.json configs[2]: the mean-teacher step fed from raw waveforms resident in HBM.  Feature extraction =
    calculate_mel_spec for the whole batch (sed_mel_frames) -> the train-time transform chain with the teacher's noisy copy
    (sed_logmel_transform, utils.py:397-412 with augment_type="noise").  Persistent buffers, no allocation and no host
    value on the hot path; the noise key is a device word of the front-end's own, advanced in stream order.

    The features depend on no weight, so they are computed ONE BATCH AHEAD - the way the reference's DataLoader workers
    prepare batch k + 1 while the model trains on batch k (DataLoad.py:47-186 behind torch's DataLoader,
    main.py:238-247).  There are two slots of (x, x_ema, target); ``run()`` trains on the slot extracted last and extracts the
    staged waveforms (``load_batch``) into the other slot, target travelling with its features.  With ``overlap=True``
    (default; needs a step that replays ONE hipGraph per step: single process, or data-parallel with captured
    collectives) the extraction is part of that graph: a second stream, forked after the forwards (or, SED_FE_FORK=gru, from
    inside the student forward between its conv stack and its recurrence: sed_crnn_fork_callback), runs the persistent STFT
    kernel on ``fe_workgroups`` CUs - beside the heads / BiGRU backward, which occupy one workgroup per (clip, direction)
    and cannot share a CU with an STFT workgroup (registers + LDS), so neither delays the other.  Without overlap the same protocol
    runs serially (train, then extract).

    Streaming real data:  ``feed(waves, target)`` per batch and ``flush()`` at the end train on every batch exactly once, in
    order, in both modes (bit-identical results: tests/test_gpu_features.py).  Resident / constant data (bench.py): ``run()``."""

    def __init__(self, step, waves, cfg=None, scaler=None, overlap=True, seed=0, fe_workgroups=None, fft_dtype="f64"):
        self.step = step
        # This front-end writes the step's input slots itself, one batch ahead - so it also computes block 0's patch moments of
        # that batch (sed_crnn_moments) right behind the extraction, on the extraction's own stream: the step's forwards then
        # start at k_blk0_prep (26 + 16 us -> 16 us at the head of the B = 64 step), and the graph gets no new branch.
        # Measured (bench.py, 800 replays, B = 64): waveform-bf16 0.8445 -> 0.8233 ms, waveform-f16 0.8698 -> 0.8442; the fp32 step
        # (1.468 -> 1.491: its conv backward fills every CU, the two extra launches land in front of k_wgrad_wino) keeps the
        # moments at the head of the forward.  SED_FE_MOMENTS=0 / 1 overrides.
        step.moments_ahead = False
        env = os.environ.get("SED_FE_MOMENTS")
        self.moments = (env == "1") if env in ("0", "1") else (step.student._dtype != _lib.DTYPE_F32)
        step._mom_external = self.moments
        self.l = _lib.lib()
        self.fx = FeatureExtractor(cfg or FeatureConfig.baseline_16k(), device=step.device, fft_dtype=fft_dtype)
        c = self.fx.cfg
        self.waves = torch.as_tensor(waves).to(step.device, torch.float32).contiguous()
        n, ns = self.waves.shape
        if n != step.B:
            raise ValueError(f"{n} waveforms for a step of batch {step.B}")
        self.n, self.ns, self.frames = n, ns, self.fx.n_frames(ns)
        self.mel = torch.empty(n, self.frames, c.n_mels, device=step.device, dtype=torch.float32)
        self.ws = self.fx.tables(exact_window=True)
        self.ws_t = torch.empty(self.l.sed_logmel_transform_ws_bytes(n), device=step.device, dtype=torch.uint8)
        self.mean = self.std = None
        if scaler is not None:
            self.mean = torch.tensor(np.asarray(scaler.mean_), dtype=torch.float64, device=step.device)
            self.std = torch.tensor(np.asarray(scaler.std_), dtype=torch.float64, device=step.device)
        self.key = torch.tensor([(int(seed) * 0x9E3779B97F4A7C15 + 0x2545F4914F6CDD1D) & 0x7FFFFFFFFFFFFFFF],
                                dtype=torch.int64, device=step.device)
        _lib.check(self.l.sed_seed_advance(_lib.ptr(self.key), _lib.stream_ptr()), "sed_seed_advance")
        self.overlap = bool(overlap) and step.single_graph and step.teacher is not None
        # how much of the chip the next batch's STFT may take while the step's recurrences run (one workgroup = one CU):
        # B x 2 recurrence workgroups need their CUs first.  Serial extraction uses the whole chip.
        n_cu = torch.cuda.get_device_properties(step.device).multi_processor_count
        if fe_workgroups is None:
            fe_workgroups = int(os.environ.get("SED_FE_WGS", max(32, n_cu - 2 * step.B)))
        self.fe_workgroups = int(fe_workgroups)
        # slot i = (x, x_ema, target); slot 0 are the step's own buffers
        self._slots = [(step.x, step.x_ema, step.target),
                       (torch.empty_like(step.x), torch.empty_like(step.x_ema), torch.empty_like(step.target))]
        self._staged_target = step.target.clone()      # until load_batch stages another one: the target the step holds now
        self._graphs = None
        self._cur = 0                                  # slot the next run() trains on
        self._primed = False
        self._runs = 0
        if self.overlap:
            # lowest priority: the next batch's features must only fill CUs the step leaves idle, never win a CU from it
            lo, _hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, 0)
            self._fe_stream = step._stream("front-end", int(os.environ.get("SED_FE_PRIO", lo)))     # (released by step.close())
            _lib.check(self.l.sed_stream_prepare(C.c_void_p(self._fe_stream.cuda_stream)), "sed_stream_prepare")

    # ---- staging ---------------------------------------------------------------------------------------------------------------
    def load_waves(self, waves):
        self.waves.copy_(torch.as_tensor(waves).reshape(self.waves.shape), non_blocking=True)

    def load_batch(self, waves, target):
        """Stage the NEXT batch to extract: its waveforms and the target that belongs to them."""
        self.load_waves(waves)
        self._staged_target.copy_(torch.as_tensor(target).reshape(self._staged_target.shape), non_blocking=True)

    def features(self, x=None, x_ema=None, target=None, workgroups=0):
        """staged waveforms -> (x, x_ema) on the current stream (default: the step's own input buffers); the staged target
        is copied beside them when `target` is given."""
        c = self.fx.cfg
        st = self.step
        x = st.x if x is None else x
        x_ema = (st.x_ema if st.teacher is not None else None) if x_ema is None else x_ema
        _lib.check(self.l.sed_mel_frames(_lib.ptr(self.waves), self.n, self.ns, c.hop_length, c.n_window,
                                         _lib.ptr(self.fx.mel_basis), c.n_mels, _lib.ptr(self.mel), _lib.ptr(self.ws),
                                         self.ws.numel(), _lib.FFT_DTYPES[self.fx.fft_dtype], int(workgroups), _lib.stream_ptr()),
                   "sed_mel_frames")
        _lib.check(self.l.sed_logmel_transform(_lib.ptr(self.mel), self.n, self.frames, c.n_mels, st.T, _lib.ptr(self.mean),
                                               _lib.ptr(self.std), _lib.ptr(self.key), _lib.ptr(x), _lib.ptr(x_ema),
                                               _lib.ptr(self.ws_t), self.ws_t.numel(), _lib.FFT_DTYPES[self.fx.fft_dtype],
                                               _lib.stream_ptr()),
                   "sed_logmel_transform")
        # the key moves on AFTER the extraction (one tiny kernel at the tail of the chain instead of its head, where it delayed
        # the STFT by a launch); the constructor advanced it once, so extraction k draws with key_0 + (k + 1) strides as before
        _lib.check(self.l.sed_seed_advance(_lib.ptr(self.key), _lib.stream_ptr()), "sed_seed_advance")
        if target is not None:
            target.copy_(self._staged_target, non_blocking=True)
        if self.moments:
            st._moments(x, st.ctx_s)
            if x_ema is not None:
                st._moments(x_ema, st.ctx_t)

    def _point_step_at(self, i):
        st = self.step
        st.x, st.x_ema, st.target = self._slots[i]

    def prime(self):
        """Extract the staged batch into the slot the next run() trains on (start-up: nothing ran ahead of the first step)."""
        self.features(*self._slots[self._cur])
        self._primed = True

    # ---- capture ---------------------------------------------------------------------------------------------------------------
    def _capture(self):
        st = self.step
        torch.cuda.synchronize(st.device)
        graphs = []
        cap = dict(stream=st._cap_stream)
        if st.dp:
            cap["capture_error_mode"] = "thread_local"
        # Where the extraction is forked (SED_FE_FORK): "backward" (default) = after both forwards, beside heads backward + GRU
        # backward (64 / 2 B workgroups: 2 B CUs); "gru" = at the student's recurrence (sed_crnn_fork_callback), beside the GRU
        # forwards too - but those are 4 B workgroups of BOTH models (k_gru4_fwd<128>: one per CU), which at B = 64 then run in
        # two rounds on the CUs the STFT leaves (64 instead of 37 us each): 0.884 against 0.873 ms per step; "start" = head of
        # the step (the extraction simply adds to the throughput-bound conv stacks).
        fork = os.environ.get("SED_FE_FORK", "backward")
        ok = 1.0
        try:
            for i in range(2):
                self._point_step_at(i)
                g = torch.cuda.CUDAGraph()
                if os.environ.get("SED_GRAPH_DUMP"):   # debugging aid: hipGraphDebugDotPrint of the captured step
                    g.enable_debug_mode()
                st._mom_ready = self.moments
                with torch.cuda.graph(g, **cap):
                    cur = torch.cuda.current_stream()
                    nxt = self._slots[1 - i]

                    def extract():
                        self._fe_stream.wait_stream(cur)
                        with torch.cuda.stream(self._fe_stream):
                            self.features(*nxt, workgroups=self.fe_workgroups)

                    if fork == "start":
                        extract()
                        st._step_body()
                    elif fork == "backward":
                        st._step_body(after_forward=extract)
                    else:
                        st._step_body(at_recurrence=extract)
                    cur.wait_stream(self._fe_stream)
                if os.environ.get("SED_GRAPH_DUMP"):
                    g.debug_dump(os.environ["SED_GRAPH_DUMP"] + f".{i}.dot")
                graphs.append(g)
        except Exception as e:                         # noqa: BLE001 - e.g. a collective that cannot be captured on this rank
            if not st.dp:
                raise
            ok, graphs = 0.0, None
            st._capture_error = repr(e)
        finally:
            st._mom_ready = False
            self._point_step_at(0)
        if st.dp:
            import torch.distributed as dist
            flag = torch.tensor([ok], device=st.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=st.pg)
            if flag.item() != 1.0:
                graphs, self.overlap = None, False     # every rank falls back to the serial protocol together
        self._graphs = graphs

    # ---- running ---------------------------------------------------------------------------------------------------------------
    def run(self):
        """One train step on the batch extracted last + extraction of the staged batch."""
        st = self.step
        if not self._primed:
            self.prime()
        if not self.overlap:
            # one slot is enough without overlap: extract-then-train keeps the step's own graph on its own buffers
            self._cur = 0
            st.run()
            self.features(*self._slots[0])
            self._runs += 1
            return
        if self._runs < 2:                             # two eager steps first (as MeanTeacherStep.run does before capturing)
            self._point_step_at(self._cur)
            st._mom_ready = self.moments
            try:
                st._step_body()
            finally:
                st._mom_ready = False
                self._point_step_at(0)
            self.features(*self._slots[1 - self._cur])
        else:
            if self._graphs is None:
                self._capture()
                if not self.overlap:                   # a rank could not capture: serial protocol from here on
                    if self._cur == 1:
                        for a, b in zip(self._slots[0], self._slots[1]):
                            a.copy_(b)
                    return self.run()
            self._graphs[self._cur].replay()
        self._cur ^= 1
        self._runs += 1
        st._warm += 1
        st.steps_done += 1

    def feed(self, waves, target):
        """Streaming interface: every batch fed is trained on exactly once, in order (the last one by ``flush``)."""
        self.load_batch(waves, target)
        if not self._primed:
            self.prime()
            return
        self.run()

    def flush(self):
        """Train on the batch extracted last without extracting another one."""
        if not self._primed:
            return
        st = self.step
        self._point_step_at(self._cur if self.overlap else 0)
        st._mom_ready = self.moments
        try:
            st._step_body() if self.overlap else st.run()
        finally:
            st._mom_ready = False
            self._point_step_at(0)
        if self.overlap:
            st._warm += 1
            st.steps_done += 1
        self._primed = False
