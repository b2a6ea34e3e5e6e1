"""bench.py - 10-s clips/s of one mean-teacher train step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: spawns its own ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one synthetic batch already resident in HBM:
teacher forward + student forward + BCE/MSE losses + student backward (+ gradient all-reduce over
RCCL when N > 1) + fused Adam + EMA, exactly what baseline/main.py:84-157 does per batch.

Workloads (--config):
  mt-f32 (default)  BASELINE.json configs[1]: mean-teacher CRNN of baseline/config.py:53-58, batch 24 per GPU,
                    precomputed log-mel [24,1,628,64] fp32 in HBM, dropout 0.5, BatchNorm in train mode for both
                    models.  N > 1: weak scaling, 24 clips per GPU, every rank keeps the [weak|unlabeled|strong] =
                    [6|12|6] composition (main.py:238-247); the same run also times configs[3]'s 64 clips per GPU
                    (global 512 at N = 8) and reports it under "config3_ddp".
  waveform          configs[2]'s workload in fp32: the step from raw 16 kHz waveforms (STFT + mel + log + normalise on
                    the GPU inside the timed region), batch 64.   waveform-bf16: configs[2] itself (bf16 MFMA operands).
  mt-bf16           configs[1]'s workload with bf16 MFMA operands in the conv-block GEMMs (sed_dims.dtype = bf16).
  wide-f32 / wide-bf16   configs[4]'s model (nb_filters 3 x 128, n_RNN_cell 256), batch 24 per GPU, features in HBM.
Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

B_PER_GPU = 24
B_CONFIG3 = 64
T_FRAMES = 628
N_MELS = 64

# algorithmic work per clip: forward GEMM-shaped FLOPs (2 x MACs of the reference's operators)
FWD_FLOP_PER_CLIP = {
    "conv0": 2 * 64 * 9 * 628 * 64, "glu0": 2 * 64 * 64 * 628 * 64,
    "conv1": 2 * 64 * 576 * 314 * 16, "glu1": 2 * 64 * 64 * 314 * 16,
    "conv2": 2 * 64 * 576 * 157 * 4, "glu2": 2 * 64 * 64 * 157 * 4,
}
STEP_FLOP_PER_CLIP = 3.432e9      # 4 x forward (teacher fwd + student fwd + 2x for backward)
STEP_BYTES_PER_CLIP = 12.0e6      # 7 passes over the block-boundary tensors, fp32
WIDE_STEP_FLOP_PER_CLIP = 14.157e9    # wide CRNN (3 x 128 filters, 256-cell BiGRU)
WIDE_STEP_BYTES_PER_CLIP = 23.9e6
N_PARAMS = {False: 214356, True: 2132628}
WAVEFORM_BYTES_PER_CLIP = 160000 * 4 + 2 * T_FRAMES * N_MELS * 4      # fp32 waveform read + two feature tensors written


def algorithmic_bytes(wide, mfma_dtype, waveform, B):
    """SURVEY.md 8(d), priced by what the mode actually stores: block-boundary tensors 12.0 / 23.9 MB per clip in fp32
    (f32, and bf16x3 - split operands, fp32 storage), 6.0 / 12.0 MB in bf16 (SED_DTYPE_BF16 stores activations as bf16);
    + 0.96 MB per clip when the step starts from waveforms; + 9 words x parameters per STEP of optimiser traffic."""
    per_clip = WIDE_STEP_BYTES_PER_CLIP if wide else STEP_BYTES_PER_CLIP
    if mfma_dtype in ("bf16", "f16"):      # (f16: the same 2-byte tensors; its bf16 copies for the backward are not algorithmic bytes)
        per_clip = 12.0e6 if wide else 6.0e6
    if waveform:
        per_clip += WAVEFORM_BYTES_PER_CLIP
    return per_clip * B + 9 * 4 * N_PARAMS[wide]
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (= the f32 vector peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak
CONFIGS = {   # name -> (wide, mfma dtype, from waveform, default batch per GPU)
    "mt-f32": (False, "f32", False, 24), "mt-bf16": (False, "bf16", False, 24),
    "waveform": (False, "f32", True, 64), "waveform-bf16": (False, "bf16", True, 64),
    "wide-f32": (True, "f32", False, 24), "wide-bf16": (True, "bf16", False, 24),
    "mt-bf16x3": (False, "bf16x3", False, 24), "wide-bf16x3": (True, "bf16x3", False, 24),
    "waveform-bf16x3": (False, "bf16x3", True, 64),
    "mt-f16": (False, "f16", False, 24), "wide-f16": (True, "f16", False, 24), "waveform-f16": (False, "f16", True, 64),
    # round 6: configs[3]'s PER-RANK shape on one GPU (fp32, 64 clips as [16|32|16]: the N = 1 point of its weak-scaling curve),
    # the reference's own geometry (44.1 kHz / hop 511: T = 864, baseline/config.py:17-22), and the all-fp32 twin of the headline
    # (block 0's backward sums as fp32 VALU FMAs instead of split-bf16 MFMA products: SED_STRICT_F32=1)
    "mt-f32-b64": (False, "f32", False, 64), "mt-f32-T864": (False, "f32", False, 24), "mt-f32-strict": (False, "f32", False, 24),
}
CONFIG_FRAMES = {"mt-f32-T864": 864}                # everything else: T_FRAMES
CONFIG_ENV = {"mt-f32-strict": {"SED_STRICT_F32": "1"}}


def block_boundary_elements(T):
    """elements per clip of the block-boundary tensors (x0, pool0, pool1, pool2, gru0, gru1, heads) at T frames, base CRNN."""
    return T * 64 + 64 * (T // 2) * 16 + 64 * (T // 4) * 4 + 64 * (T // 8) + 2 * (T // 8) * 128 + (T // 8) * 20


def frames_scale(T):
    """(flop scale, byte scale) of a base-CRNN step at T frames relative to T = 628 (858.1 -> 1180.8 MFLOP at 864)."""
    if T == T_FRAMES:
        return 1.0, 1.0
    if T == 864:
        return 1180.8 / 858.1, block_boundary_elements(864) / block_boundary_elements(T_FRAMES)
    return T / T_FRAMES, block_boundary_elements(T) / block_boundary_elements(T_FRAMES)
PEAK_HBM_GBS = 8000.0


def synthetic_batch(B, T, seed, device):
    """x ~ N(0,1) (already-normalised log-mel), teacher input = a second draw,
    targets: rows [0,B/4) weak-style, [B/4,3B/4) unlabeled (-1), [3B/4,B) strong-style."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(B, 1, T, N_MELS, generator=g)
    xe = x + 0.1 * torch.randn(B, 1, T, N_MELS, generator=g).abs()
    T3 = T // 8
    tgt = torch.zeros(B, T3, 10)
    nw = B // 4
    tgt[:nw] = (torch.rand(nw, 1, 10, generator=g) < 0.2).float()
    tgt[nw:B - nw] = -1.0
    tgt[B - nw:] = (torch.rand(nw, T3, 10, generator=g) < 0.2).float()
    return x.to(device), xe.to(device), tgt.to(device), slice(nw), slice(B - nw, B)


def weights_init_(m):
    """weights_init (baseline/utils/utils.py:205-224): random init of the reference architecture."""
    for mod in m.modules():
        name = mod.__class__.__name__
        if name.find('Conv2d') != -1:
            torch.nn.init.xavier_uniform_(mod.weight, gain=np.sqrt(2)); mod.bias.data.fill_(0)
        elif name.find('BatchNorm') != -1:
            mod.weight.data.normal_(1.0, 0.02); mod.bias.data.fill_(0)
        elif name.find('GRU') != -1:
            for w in mod.parameters():
                if len(w.size()) > 1:
                    torch.nn.init.orthogonal_(w.data)
        elif name.find('Linear') != -1:
            mod.weight.data.normal_(0, 0.01); mod.bias.data.zero_()


def build_models(device, seed, **over):
    from dcase2019_task4_amd.crnn import CRNN
    kw = dict(n_in_channel=1, nclass=10, attention=True, n_RNN_cell=64, n_layers_RNN=2, activation="glu", dropout=0.5,
              kernel_size=3 * [3], padding=3 * [1], stride=3 * [1], nb_filters=[64, 64, 64], pooling=list(3 * ((2, 4),)))
    kw.update(over)
    torch.manual_seed(seed)
    models = []
    for _ in range(2):
        m = CRNN(**kw)
        weights_init_(m)
        models.append(m.to(device).train())
    return models


def usable_cores():
    """Host cores this process may actually use (affinity mask and cgroup CPU quota, not just cpu_count)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(budget_s=60.0, warm=2, timed_steps=5):
    """The oracle (oracle/ref_cpu.py, a torch-CPU port of the reference step) on this box's host cores: B=24, T=628,
    2 warm-up + 5 timed steps (SURVEY 8(d) / BASELINE.md section 3), cut short only if `budget_s` runs out."""
    from oracle import ref_cpu, synth
    cores = usable_cores()
    torch.set_num_threads(cores)
    B, T = B_PER_GPU, T_FRAMES
    mt = ref_cpu.MeanTeacherOracle(synth.make_params(0), synth.make_params(1))
    x, xe, tgt, wm, sm = synthetic_batch(B, T, 0, "cpu")

    def masks(seed):
        g = torch.Generator().manual_seed(seed)
        mk = lambda *s: (torch.rand(*s, generator=g) < 0.5).float() * 2.0
        return {"drop0": mk(B, T, 64, 64), "drop1": mk(B, T // 2, 16, 64), "drop2": mk(B, T // 4, 4, 64),
                "drop_rnn": mk(B, T // 8, 128)}
    times = []
    t_start = time.perf_counter()
    for it in range(warm + timed_steps):
        t0 = time.perf_counter()
        mt.step(x, xe, tgt, wm, sm, 10500, masks(2 * it), masks(2 * it + 1))     # mask draw timed like nn.Dropout's
        times.append(time.perf_counter() - t0)
        print(f"[cpu_baseline] step {it}: {times[-1]:.2f} s ({cores} threads)", file=sys.stderr, flush=True)
        if it >= warm and time.perf_counter() - t_start > budget_s:
            break
    timed = times[warm:] if len(times) > warm else times[-1:]
    dt = float(np.mean(timed))
    return {"value": round(B / dt, 3), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"oracle/ref_cpu.MeanTeacherOracle, B={B} T={T}, {warm} warm-up + {len(timed)} timed steps, "
                      f"{dt:.2f} s/step, torch {torch.__version__} CPU threads={torch.get_num_threads()}"}


def feature_path(device, n_clips=32):
    """BASELINE.md section 3's feature-path line: waveform -> linear mel (sed_mel_frames, fp64 butterflies) -> log / pad / normalise
    (sed_logmel_transform) for 32 clips of 160 000 samples at 16 kHz, HIP-event timed on the launch stream, beside the
    numpy restatement of the same path (oracle/features_np.py) on the host cores (bounded: 4 clips)."""
    from dcase2019_task4_amd.features import FeatureConfig, FeatureExtractor, LogMelTransform
    from oracle import features_np
    fx = FeatureExtractor(FeatureConfig.baseline_16k(), device=device)
    g = torch.Generator().manual_seed(5)
    wave = (0.1 * torch.randn(n_clips, 160000, generator=g)).to(device)
    tr = LogMelTransform(T_FRAMES, scaler=None, device=device)

    def timed(fn, iters=20):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    mel = fx.calculate_mel_spec_batch(wave)
    ms_stft = timed(lambda: fx.calculate_mel_spec_batch(wave))
    ms_stft32 = timed(lambda: fx.calculate_mel_spec_batch(wave, fft_dtype="f32"))
    ms_tr = timed(lambda: tr(mel))
    ms = timed(lambda: tr(fx.calculate_mel_spec_batch(wave)))
    # algorithmic bytes: the waveform read once (fp32) + the feature tensor written once; the linear mel in between is
    # written and read once more (two kernels)
    by_stft = n_clips * (160000 * 4 + T_FRAMES * N_MELS * 4)
    by_tr = n_clips * (2 * T_FRAMES * N_MELS * 4)
    by = by_stft + by_tr
    w = wave[:4].cpu().numpy().astype(np.float64)
    t0 = time.perf_counter()
    for i in range(w.shape[0]):
        m = features_np.calculate_mel_spec(w[i], 16000, 2048, 255, 64, 0.0, 8000.0)
        features_np.transform_chain(m, T_FRAMES)
    cpu_s = (time.perf_counter() - t0) / w.shape[0]
    per = lambda b, t: {"us": round(t * 1e3, 1), "hbm_gbs_algorithmic": round(b / t * 1e-6, 1),
                        "frac_of_hbm_peak": round(b / t * 1e-6 / PEAK_HBM_GBS, 4)}
    return {"gpu_clips_per_s": round(n_clips / ms * 1e3, 1), "ms_per_32_clips": round(ms, 3),
            "hbm_gbs_algorithmic": round(by / ms * 1e-6, 1), "frac_of_hbm_peak": round(by / ms * 1e-6 / PEAK_HBM_GBS, 4),
            "kernels": {"sed_mel_frames SED_FFT_F64 (k_stft_mel_p<double>: persistent STFT + mel projection)":
                        dict(per(by_stft, ms_stft), bound="fp64 VALU issue + LDS exchange writes (DESIGN.md 3.12): 1024-point complex "
                             "FFT per frame as 16 x 16 x 4 in registers, three exchanges through LDS, tables resident in LDS"),
                        "sed_mel_frames SED_FFT_F32 (k_stft_mel_p<float>, the stated fp32 mode)":
                        dict(per(by_stft, ms_stft32), bound="packed-fp32 VALU issue + LDS exchange writes"),
                        "sed_logmel_transform (k_logmel_max + k_logmel_apply, no augmentation)": dict(per(by_tr, ms_tr),
                        bound="fp64 log10 per element, then HBM")},
            "cpu_clips_per_s": round(1.0 / cpu_s, 2), "cpu_kind": "port (oracle/features_np.py, numpy, 1 process)",
            "cpu_sample": "4 clips of 160 000 samples"}


def kernel_roofline(step, iters=20):
    """HIP-event timing of the step's main kernels, each re-launched on the step's own buffers (same shapes and data
    as in the timed region) on the current stream.  Per kernel: reference ("effective") FLOPs - what the reference's
    operator costs - AND executed FLOPs - what this kernel actually issues on the MFMA pipe (the Winograd kernels
    issue 16/36 of a direct 3x3 convolution's multiplies; block 0 folds conv0 + BatchNorm + the GLU's Linear into
    one K = 10 convolution)."""
    import ctypes as C
    from dcase2019_task4_amd import _lib
    l = _lib.lib()
    B = step.B
    px0, px1, px2 = B * 628 * 64, B * 314 * 16, B * 157 * 4
    c1, c2 = B * FWD_FLOP_PER_CLIP["conv1"], B * FWD_FLOP_PER_CLIP["conv2"]
    blk0_ref = B * (FWD_FLOP_PER_CLIP["conv0"] + FWD_FLOP_PER_CLIP["glu0"])
    blk0_exec = px0 * 2 * 10 * 128             # one K = 10 convolution to 128 channels (lin | z) on the MFMA
    gru_ref = lambda nin: B * 78 * 2 * 2 * 192 * (nin + 64)
    algo = {  # name -> (reference flops, executed MFMA flops, algorithmic hbm bytes, bound, per-step launches, steps)
        "conv1_fwd": (c1, c1 * 16 / 36, 2 * px1 * 64 * 4 + 9 * 4096 * 4, "mfma", 2),
        "conv1_dgrad": (c1, c1 * 16 / 36, 3 * px1 * 64 * 4 + 9 * 4096 * 4, "mfma", 1),
        "conv1_wgrad": (c1, c1 * 16 / 36, 3 * px1 * 64 * 4, "mfma", 1),
        "conv2_fwd": (c2, c2 * 16 / 36, 2 * px2 * 64 * 4 + 9 * 4096 * 4, "mfma", 2),
        "conv2_dgrad": (c2, c2 * 16 / 36, 3 * px2 * 64 * 4 + 9 * 4096 * 4, "mfma", 1),
        "conv2_wgrad": (c2, c2 * 16 / 36, 3 * px2 * 64 * 4, "mfma", 1),
        "glu1_fwd": (B * FWD_FLOP_PER_CLIP["glu1"], B * FWD_FLOP_PER_CLIP["glu1"], px1 * 64 * 4 + px1 * 8 * 4, "mfma+valu", 2),
        "glu1_bwd": (3 * B * FWD_FLOP_PER_CLIP["glu1"], 3 * B * FWD_FLOP_PER_CLIP["glu1"], 2 * px1 * 64 * 4 + px1 * 8 * 4, "mfma+valu", 1),
        "blk0_fwd": (blk0_ref, blk0_exec, B * 628 * 64 * 4 + px1 * 64 * 4, "valu", 2),
        "blk0_bwd": (2 * blk0_ref, blk0_exec, B * 628 * 64 * 4 + px1 * 64 * 4, "valu", 1),
        "gru0_fwd": (gru_ref(64), B * 78 * 2 * 2 * 192 * 64, B * 78 * (64 + 128 + 512) * 4, "latency", 2),
        "gru1_fwd": (gru_ref(128), B * 78 * 2 * 2 * 192 * 128, B * 78 * (128 + 128 + 512) * 4, "latency", 2),
        "gru1_bwd": (2 * gru_ref(128), B * 78 * 2 * 2 * 192 * 128, B * 78 * (128 + 512 + 384 * 2 + 128 + 256) * 4, "latency", 1),
        "gru0_bwd": (2 * gru_ref(64), B * 78 * 2 * 2 * 192 * 64, B * 78 * (128 + 512 + 384 * 2 + 128 + 128) * 4, "latency", 1),
    }
    out = {}
    st = _lib.stream_ptr()
    for name, (fl, fx, by, bound, per_step) in algo.items():
        def call():
            _lib.check(l.sed_kernel_replay(name.encode(), C.byref(step.dims), _lib.ptr(step.student._flat), _lib.ptr(step.x),
                                           step._seed_s, _lib.ptr(step.ctx_s), step.ctx_bytes, _lib.ptr(step.grads),
                                           _lib.ptr(step.ws), step.ws_bytes, st), name)
        call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        row = {"us": round(us, 2), "launches_per_step": per_step, "bound": bound,
               "effective_tflops": round(fl / us * 1e-6, 2), "executed_mfma_tflops": round(fx / us * 1e-6, 2),
               "effective_frac_of_f32_peak": round(fl / us * 1e-6 / PEAK_F32_MFMA_TFLOPS, 4),
               "executed_frac_of_f32_peak": round(fx / us * 1e-6 / PEAK_F32_MFMA_TFLOPS, 4),
               "gbs": round(by / us * 1e-3, 1), "flops": int(fl), "executed_flops": int(fx), "bytes": int(by)}
        if bound == "latency":
            row["us_per_time_step"] = round(us / 78.0, 3)
        out[name] = row
    return out


def pmc_step_traffic():
    """HBM bytes per step from the committed PMC passes (profiles/pmc_traffic.json, written by tools/summarize_pmc.py from
    two rocprofv3 --pmc passes of THIS bench command: 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes).
    The per-step figure is the sum over every dispatch of the traced run divided by the steps it ran - the launches per
    step come out of the trace itself ("_per_step"), nothing is maintained by hand here."""
    pmc_file = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if not os.path.exists(pmc_file):
        return None, None
    table = json.load(open(pmc_file))
    per_step = table.get("_per_step")
    return (int(per_step["bytes"]) if per_step else None), table


def free_port():
    """A TCP port nothing listens on right now (bind to port 0 on the loopback and let the kernel pick)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args):
    """--gpus N without a launcher: re-exec through torch.distributed.run (one rank per GPU, RCCL)."""
    n_dev = torch.cuda.device_count()
    # SED_BENCH_SHARE_GPU=1 (code-path test on a box with fewer GPUs than ranks, tests/test_gpu_dp.py): the ranks share the
    # visible GPUs round-robin and talk over gloo - RCCL refuses two ranks on one device, the library's own all-reduce does not.
    # Such a run says nothing about scaling; its JSON line carries "shared_gpu": true.
    if n_dev < args.gpus and os.environ.get("SED_BENCH_SHARE_GPU") != "1":
        raise SystemExit(f"--gpus {args.gpus} but only {n_dev} GPU(s) are visible")
    port = int(os.environ.get("MASTER_PORT", "0")) or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"[bench] spawning {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=env))


def collective_bench(pg, device, world, rank, iters=50):
    """The gradient all-reduce ALONE, so that a scaling record explains itself: the two message sizes of the step (857 424 B =
    cfg.crnn_kwargs' flat gradient, 8 530 512 B = the wide model's) through the library's peer kernel (csrc/p2p.hip) and through
    the process group's all_reduce (RCCL), each captured into a hipGraph and replayed `iters` times (dist.graph_time_us: HIP events
    on the launch stream, MAX over ranks).  us per call, algorithm bandwidth (bytes / time) and bus bandwidth (x 2 (W-1) / W)."""
    import torch.distributed as dist
    from dcase2019_task4_amd import dist as sdist
    out = {"world": world, "iters": iters, "backend": dist.get_backend(pg), "sizes": {}}
    n_big = N_PARAMS[True]
    p2p = None
    try:
        p2p = sdist.PeerAllReduce.create(n_big, device, pg)
        if p2p is None:
            out["p2p_unavailable"] = str(sdist.PeerAllReduce.last_error)
        for label, n in (("base_857KB", N_PARAMS[False]), ("wide_8.5MB", n_big)):
            row = {"bytes": 4 * n}

            def fill(tag, us):
                row[tag + "_us"] = round(us, 2)
                row[tag + "_algbw_gbs"] = round(4 * n / us * 1e-3, 1)
                row[tag + "_busbw_gbs"] = round(4 * n / us * 1e-3 * 2 * (world - 1) / world, 1)
            if p2p is not None:
                try:
                    fill("p2p", p2p.time_us(n, iters))
                except Exception as e:                  # noqa: BLE001
                    row["p2p_error"] = repr(e)[:200]
            if dist.get_backend(pg) == "nccl" and world > 1:
                try:
                    fill("rccl", sdist.graph_time_us(lambda t: dist.all_reduce(t, group=pg), n, device, pg, iters))
                except Exception as e:                  # noqa: BLE001
                    row["rccl_error"] = repr(e)[:200]
            out["sizes"][label] = row
        if p2p is not None:
            out["p2p_timeouts"] = p2p.errors(reduce=True)
    except Exception as e:                              # noqa: BLE001 - the headline line must not depend on this leg
        out["error"] = repr(e)[:300]
    finally:
        if p2p is not None:
            p2p.close()
    if rank == 0:
        print(f"[bench] collective-only leg: {json.dumps(out)}", file=sys.stderr, flush=True)
    return out


def time_steps(step, steps, world, device):
    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(device)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step.run()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    barrier()
    per_rank = [elapsed]
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        tl = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(tl, t)
        per_rank = [float(v.item()) for v in tl]
        elapsed = max(per_rank)
    time_steps.per_rank = per_rank
    # SURVEY 8(d)'s prescription beside the wall clock: a hipEvent pair on the launch stream around the same number of steps,
    # recorded right behind up to 50 more (untimed) replays with NO host synchronisation in between - the first dozen replays
    # after a synchronise run 3 - 10 % slow (tools/first_replays.py), which is what a 20-step wall-clock region mostly measures
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(min(steps, 50)):
        step.run()
    e0.record()
    for _ in range(steps):
        step.run()
    e1.record()
    torch.cuda.synchronize(device)
    ev = e0.elapsed_time(e1) * 1e-3
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([ev], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ev = float(t.item())
    barrier()
    time_steps.events_s = ev
    return elapsed


def replicas_bit_identical(step, world, device):
    """After the timed steps every rank must hold bit-identical students, teachers and Adam moments (each rank saw different clips
    and masks; only the all-reduced gradient couples them): a 64-bit XOR / sum fingerprint of the raw bits of each buffer, gathered
    and compared.  Reported in the JSON line - an all-reduce that were not bit-reproducible across ranks would show here."""
    import torch.distributed as dist
    bufs = [step.student._flat, step.exp_avg, step.exp_avg_sq] + ([step.teacher._flat] if step.teacher is not None else [])
    fp = []
    for t in bufs:
        bits = t.contiguous().view(torch.int32).to(torch.int64)
        fp += [bits.sum(), (bits * (torch.arange(bits.numel(), device=bits.device) % 8191 + 1)).sum()]
    mine = torch.stack(fp)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    return bool(all(torch.equal(allv[0], v) for v in allv[1:]))


def same_on_all_ranks(step, world, device):
    """Every rank must have ended on the same data-parallel schedule (a rank that fell back alone would deadlock or, worse,
    reduce different buckets): gather (schedule, captured) and fail loudly on any difference."""
    mine = (step.dp_schedule, bool(step.dp_capture))
    if world == 1:
        return [mine]
    import torch.distributed as dist
    allv = [None] * world
    dist.all_gather_object(allv, mine)
    if any(v != allv[0] for v in allv):
        raise SystemExit(f"[bench] data-parallel schedules differ across ranks: {allv}")
    return allv


ARITH = {"f32": "fp32 (exact fp32 MFMA; block 0's backward sums on split bf16 operands, DESIGN.md 3.10)",
         "bf16": "SED_DTYPE_BF16: bf16 MFMA operands + bf16 activation storage in the conv blocks (3x3 convolutions forward / "
                 "dgrad / wgrad, GLU Linear, block 0), bf16 GRU weight-gradient GEMMs, bf16 W_hh / projections at H = 256; "
                 "fp32 accumulation everywhere, fp32 recurrence at H = 64, heads, BatchNorm statistics, losses, Adam",
         "bf16x3": "SED_DTYPE_BF16X3: split bf16 operands (hi + lo, three bf16 MFMAs per product, fp32 accumulation and fp32 "
                   "storage) in the conv-block GEMMs, fp32 elsewhere",
         "f16": "SED_DTYPE_F16: SED_DTYPE_BF16 with its FORWARD chain in fp16 - fp16 MFMA operands (same rate as bf16, 11-bit "
                "significand) in conv block 0, the 3x3 convolutions, the GLU Linear and, at H = 256, the gi projections and the W_hh / h "
                "operands of the recurrence; p0 / y1 / p1 / y2 handed on as fp16; the backward is the bf16 mode's (bf16 gradient "
                "tensors and operands, reading bf16 copies of the activations that the forward kernels write as well); posteriors "
                "asserted within the north star's 1e-3 of the fp32 oracle"}


def workload_string(wide, mfma_dtype, waveform, B, fft="f32", T=T_FRAMES, strict=False):
    mdl = "wide CRNN (nb_filters 3 x 128, n_RNN_cell 256)" if wide else "CRNN (baseline/main.py config)"
    if strict:
        return (f"mean-teacher {mdl} train step, batch {B} per GPU, precomputed log-mel [{B},1,{T},64] fp32 resident in HBM, dropout 0.5, "
                "STRICT fp32: exact fp32 MFMA products everywhere and block 0's backward sums as fp32 VALU FMAs (no split-bf16 "
                "product anywhere in the step)")
    if waveform:
        return (f"mean-teacher {mdl} train step from raw 16 kHz waveforms (STFT [{fft} butterflies] + mel + noise + log + "
                f"normalise on the GPU inside the timed region, one batch ahead inside the step's hipGraph), batch {B} per GPU, "
                f"{ARITH[mfma_dtype]}")
    return (f"mean-teacher {mdl} train step, batch {B} per GPU, precomputed log-mel [{B},1,{T},64] fp32 resident in HBM, "
            f"dropout 0.5, {ARITH[mfma_dtype]}")


def strict_f32(config):
    return os.environ.get("SED_STRICT_F32") == "1" or CONFIG_ENV.get(config, {}).get("SED_STRICT_F32") == "1"


def make_runner(config, device, rank, pg=None, use_graph=True, batch=None, seed=1234, pool_streams=True):
    """Models + step (+ waveform front-end) of one workload of CONFIGS on synthetic data resident in HBM."""
    from dcase2019_task4_amd.train import MeanTeacherStep
    wide, mfma_dtype, waveform, b_default = CONFIGS[config]
    B = batch or b_default
    T = CONFIG_FRAMES.get(config, T_FRAMES)
    if strict_f32(config):
        from dcase2019_task4_amd import _lib
        _lib.lib().sed_debug_set(_lib.lib().sed_debug_set(0) | (1 << 27))
    model_kw = dict(mfma_dtype=mfma_dtype)
    if wide:
        model_kw.update(nb_filters=[128, 128, 128], n_RNN_cell=256)
    student, teacher = build_models(device, seed=0, **model_kw)        # identical replicas on every rank
    x, xe, tgt, wm, sm = synthetic_batch(B, T, 1000 + rank, device)
    step = MeanTeacherStep(student, teacher, B, T, rampup_length=210 * 100 // 2, weak_mask=wm, strong_mask=sm,
                           seed=seed, use_graph=use_graph, process_group=pg, pool_streams=pool_streams)
    step.load_batch(x, xe, tgt)
    runner = step
    if waveform:
        from dcase2019_task4_amd.features import FeatureConfig, WaveformFrontEnd
        g = torch.Generator().manual_seed(77 + rank)
        wave = (0.1 * torch.randn(B, 160000, generator=g)).to(device)
        runner = WaveformFrontEnd(step, wave, FeatureConfig.baseline_16k(), fft_dtype=os.environ.get("SED_FE_FFT", "f32"))
    return runner, step, B


def step_roofline(wide, mfma_dtype, waveform, B, ms_per_step, T=T_FRAMES):
    """SURVEY 8(d)'s whole-step figures for one measured line."""
    fs, bs = frames_scale(T)
    flop = (WIDE_STEP_FLOP_PER_CLIP if wide else STEP_FLOP_PER_CLIP) * B * fs
    by = (algorithmic_bytes(wide, mfma_dtype, waveform, B) - 9 * 4 * N_PARAMS[wide]) * bs + 9 * 4 * N_PARAMS[wide]
    peak = PEAK_F32_MFMA_TFLOPS if mfma_dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
    tf = flop / (ms_per_step * 1e-3) * 1e-12
    return {"bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
            "frac_of_f32_mfma_peak": round(tf / PEAK_F32_MFMA_TFLOPS, 4), "algorithmic_flops": int(flop),
            "algorithmic_bytes": int(by), "algorithmic_gbs": round(by / (ms_per_step * 1e-3) * 1e-9, 1),
            "hbm_frac": round(by / (ms_per_step * 1e-3) * 1e-9 / PEAK_HBM_GBS, 4)}


def extra_config_legs(device, steps=300):
    """The other single-GPU BASELINE.json workloads, timed in the SAME driver run as the headline line (round 3's numbers
    for them were builder-printed): configs[2] (waveform-bf16, batch 64), configs[4]'s model at its per-GPU shape (wide-bf16,
    wide-bf16x3), and configs[1]'s workload in the two reduced-precision modes.  `steps` hipGraph replays each after warm-up.
    Every leg runs in a process of its own (this script with --config <c> --no-extras --no-cpu-baseline): a process that has
    already built and replayed four other steps holds their streams, and a later step's graph branches then share hardware
    queues with them - measured in-process, the fifth leg came out at 0.90 ms against 0.66 ms alone."""
    out = {}
    for name in ("mt-f32-b64", "mt-f32-T864", "mt-f32-strict", "waveform-bf16", "waveform-f16", "wide-bf16", "wide-f16", "wide-bf16x3",
                 "mt-bf16", "mt-f16", "mt-bf16x3"):
        try:
            wide, mfma_dtype, waveform, B = CONFIGS[name]
            cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--steps", str(steps), "--warmup", "8", "--no-extras",
                   "--no-cpu-baseline"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-300:])
            d = json.loads(r.stdout.strip().split("\n")[-1])
            ms = d["ms_per_step"]
            out[name] = {"value": d["value"], "unit": "clips/s", "ms_per_step": ms, "ms_per_step_events": d.get("ms_per_step_events"),
                         "steps": steps, "dtype": mfma_dtype,
                         "global_batch": B, "loss": d.get("loss"), "workload": d["config"]["workload"],
                         "roofline": step_roofline(wide, mfma_dtype, waveform, B, ms, CONFIG_FRAMES.get(name, T_FRAMES))}
            if waveform and os.environ.get("SED_FE_FFT", "f32") != "f64":
                # the same leg with the parity-mode front-end (fp64 butterflies = librosa's arithmetic; WaveformFrontEnd's default)
                r64 = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, SED_FE_FFT="f64"))
                if r64.returncode == 0:
                    d64 = json.loads(r64.stdout.strip().split("\n")[-1])
                    out[name]["f64_front_end"] = {"value": d64["value"], "ms_per_step": d64["ms_per_step"],
                                                  "workload": d64["config"]["workload"]}
            print(f"[bench] extra config {name}: {ms:.4f} ms/step, {out[name]['value']} clips/s", file=sys.stderr, flush=True)
        except Exception as e:                          # noqa: BLE001 - the headline line must not depend on these legs
            out[name] = {"error": repr(e)[:300]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="mt-f32", choices=sorted(CONFIGS))
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the kernel table, feature-path and config3 legs")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--trace-only-this-config", action="store_true",
                    help="for rocprofv3 runs (tools/collect_profiles.sh): keep the headline's kernel-table leg (solo re-launches) "
                         "but skip the extra_configs child processes and the feature-path leg, so that the trace holds ONE "
                         "process and only this workload's kernels")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the only product path")
    share_gpu = os.environ.get("SED_BENCH_SHARE_GPU") == "1" and torch.cuda.device_count() < world
    dev_index = local_rank % torch.cuda.device_count() if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    pg = None
    dist_info = None
    if world > 1 or os.environ.get("SED_FORCE_DP") == "1":      # SED_FORCE_DP: one-rank RCCL group (path test on a 1-GPU box)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        pg = dist.group.WORLD
        dist_info = {"backend": dist.get_backend(pg), "world_size": dist.get_world_size(pg), "shared_gpu": share_gpu,
                     "rccl": ".".join(str(v) for v in torch.cuda.nccl.version()),
                     # RCCL picks algorithm / protocol per message size unless pinned; what this run pinned (nothing by default)
                     "env": {k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS",
                                                            "RCCL_MSCCL_ENABLE", "HSA_ENABLE_IPC_MODE_LEGACY")}}
        if rank == 0:
            print(f"[bench] process group up: {dist_info}", file=sys.stderr, flush=True)

    from dcase2019_task4_amd.train import MeanTeacherStep
    wide, mfma_dtype, waveform, b_default = CONFIGS[args.config]
    headline = args.config == "mt-f32"
    T_cfg = CONFIG_FRAMES.get(args.config, T_FRAMES)
    runner, step, B = make_runner(args.config, device, rank, pg, use_graph=not args.no_graph, batch=args.batch)
    x, xe, tgt, wm, sm = synthetic_batch(B, T_cfg, 1000 + rank, device)          # (the same batch again, for the A/B legs below)
    for _ in range(max(args.warmup, 3)):       # >= 3: two eager warm-ups + graph capture/first replay
        runner.run()
    elapsed = time_steps(runner, args.steps, world, device)
    events_s = time_steps.events_s
    per_rank_s = list(time_steps.per_rank)
    meters = step.meters()
    assert np.isfinite(meters["loss"]), meters
    step.check_health()
    schedules = same_on_all_ranks(step, world, device) if step.dp else None
    replicas_identical = replicas_bit_identical(step, world, device) if (step.dp and world > 1) else None

    # N > 1: the same workload under the other data-parallel schedules, so that ONE driver run yields the comparison
    # (captured overlap = the default above | single all-reduce, eager | overlap with eager collectives)
    ab_legs = None
    if world > 1 and headline and not args.no_extras and args.batch is None:
        ab_legs = {}
        # (the headline runs the default: overlap schedule, collectives captured, the library's own peer-memory all-reduce when
        # its self-check passes - config.dp_collective says which ran; these legs price the alternatives)
        for tag, sched, cap, coll in (("overlap_captured_rccl", "overlap", None, "pg"), ("single_eager_rccl", "single", "0", "pg"),
                                      ("overlap_eager_rccl", "overlap", "0", "pg"), ("single_eager_p2p", "single", None, "p2p")):
            try:
                if cap is not None:
                    os.environ["SED_DP_CAPTURE"] = cap
                sa, ta = build_models(device, seed=0)
                stp = MeanTeacherStep(sa, ta, B, T_FRAMES, rampup_length=210 * 100 // 2, weak_mask=wm, strong_mask=sm,
                                      seed=1234, use_graph=not args.no_graph, process_group=pg, dp_schedule=sched,
                                      collective=coll)
                stp.load_batch(x, xe, tgt)
                for _ in range(5):
                    stp.run()
                na = max(50, args.steps // 4)
                ela = time_steps(stp, na, world, device)
                same_on_all_ranks(stp, world, device)
                ab_legs[tag] = {"value": round(B * world * na / ela, 1), "unit": "clips/s", "ms_per_step": round(ela / na * 1e3, 4),
                                "steps": na, "dp_schedule": stp.dp_schedule, "dp_collectives": "captured" if stp.dp_capture else "eager",
                                "dp_collective": stp.collective}
                stp.close()
                del stp
            except Exception as e:                      # noqa: BLE001 - the headline line must not depend on these legs
                ab_legs[tag] = {"error": repr(e)[:300]}
            finally:
                os.environ.pop("SED_DP_CAPTURE", None)

    coll_bench = None
    if step.dp and headline and not args.no_extras and args.batch is None:
        coll_bench = collective_bench(pg, device, world, rank)

    config3 = None
    if world > 1 and headline and not args.no_extras and args.batch is None:
        # BASELINE.json configs[3]: global batch 512 at N = 8 = 64 clips per GPU ([16|32|16] per rank)
        try:                                        # the headline line must not depend on this extra leg
            s3, t3 = build_models(device, seed=0)
            x3, xe3, tg3, wm3, sm3 = synthetic_batch(B_CONFIG3, T_FRAMES, 2000 + rank, device)
            step3 = MeanTeacherStep(s3, t3, B_CONFIG3, T_FRAMES, rampup_length=210 * 100 // 2, weak_mask=wm3, strong_mask=sm3,
                                    seed=4321, use_graph=not args.no_graph, process_group=pg)
            step3.load_batch(x3, xe3, tg3)
            for _ in range(5):
                step3.run()
            n3 = max(50, args.steps // 4)
            el3 = time_steps(step3, n3, world, device)
            config3 = {"workload": "BASELINE.json configs[3]: 64 clips per GPU", "global_batch": B_CONFIG3 * world,
                       "value": round(B_CONFIG3 * world * n3 / el3, 1), "unit": "clips/s", "ms_per_step": round(el3 / n3 * 1e3, 4),
                       "steps": n3}
            del step3
        except Exception as e:                      # noqa: BLE001
            config3 = {"error": repr(e)[:300]}

    config4 = None
    if world > 1 and headline and not args.no_extras and args.batch is None:
        # BASELINE.json configs[4]: the wide CRNN in bf16 under the data-parallel step, 24 clips per GPU
        try:
            r4, step4, B4 = make_runner("wide-bf16", device, rank, pg, use_graph=not args.no_graph, seed=2468)
            for _ in range(5):
                r4.run()
            n4 = max(50, args.steps // 4)
            el4 = time_steps(r4, n4, world, device)
            same_on_all_ranks(step4, world, device)
            step4.check_health()
            ms4 = el4 / n4 * 1e3
            config4 = {"workload": "BASELINE.json configs[4]: " + workload_string(True, "bf16", False, B4), "global_batch": B4 * world,
                       "value": round(B4 * world * n4 / el4, 1), "unit": "clips/s", "ms_per_step": round(ms4, 4), "steps": n4,
                       "dtype": "bf16", "dp_schedule": step4.dp_schedule,
                       "dp_collectives": "captured" if step4.dp_capture else "eager", "dp_collective": step4.collective,
                       "gradient_bytes_per_step": 4 * N_PARAMS[True]}
            del r4, step4
        except Exception as e:                      # noqa: BLE001
            config4 = {"error": repr(e)[:300]}

    # the driver times 20 steps (15 ms): a steady-state figure of the same workload beside it
    steady = None
    if world == 1 and headline and not args.no_extras and args.steps < 1000:
        for _ in range(20):
            runner.run()
        n_ss = 3000
        el_ss = time_steps(runner, n_ss, world, device)
        steady = {"steps": n_ss, "ms_per_step": round(el_ss / n_ss * 1e3, 4), "ms_per_step_events": round(time_steps.events_s / n_ss * 1e3, 4),
                  "value": round(B * n_ss / el_ss, 1), "unit": "clips/s",
                  "note": "same step, same buffers, 3000 hipGraph replays timed the same way (the headline's timed region is "
                          "K steps as the driver asks: 20 steps = 15 ms is dominated by the first replays)"}

    extras = None
    if world == 1 and headline and not args.no_extras and args.batch is None and rank == 0 and not args.trace_only_this_config:
        extras = extra_config_legs(device)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        clips = B * world * args.steps / elapsed
        wl = workload_string(wide, mfma_dtype, waveform, B, os.environ.get("SED_FE_FFT", "f32"), T_cfg, strict_f32(args.config))
        res = {
            "metric": "10-s clips/sec mean-teacher train step (64-mel x 628)",
            "value": round(clips, 1), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4),
            # (hipEvent pair on the launch stream around the same K steps, recorded behind further replays with no host
            # synchronisation in between - SURVEY 8(d)'s method; ms_per_step above is the wall clock the contract asks for)
            "ms_per_step_events": round(events_s / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": mfma_dtype, "data": "synthetic",
            "config": {"workload": wl, "global_batch": B * world, "frames": T_cfg, "n_mels": N_MELS,
                       "parallelism": f"dp{world}", "hip_graph": not args.no_graph,
                       "dp_schedule": step.dp_schedule if step.dp else None,
                       "dp_collectives": ("captured" if step.dp_capture else "eager") if step.dp else None,
                       "dp_collective": ({"p2p": "library kernel over peer-mapped memory (csrc/p2p.hip: reduce-scatter + all-gather in "
                                                 "one launch, direct xGMI loads / stores, self-checked against the process group)",
                                          "pg": "torch.distributed all_reduce (RCCL picks algorithm / protocol)"}[step.collective]
                                         if step.dp else None),
                       # collective="auto" TIMES both on the step's two buckets at construction and keeps the faster:
                       "dp_collective_record": getattr(step, "collective_record", None) if step.dp else None},
            "loss": round(meters["loss"], 5),
        }
        if dist_info:
            res["distributed"] = dist_info
            dist_info["ms_per_step_per_rank"] = {"min": round(min(per_rank_s) / args.steps * 1e3, 4),
                                                 "max": round(max(per_rank_s) / args.steps * 1e3, 4),
                                                 "all": [round(v / args.steps * 1e3, 4) for v in per_rank_s]}
            dist_info["schedule_per_rank"] = [f"{a}/{'captured' if b else 'eager'}" for a, b in (schedules or [])]
            dist_info["replicas_bit_identical_after_timed_steps"] = replicas_identical
            if ab_legs:
                dist_info["schedule_ab"] = ab_legs
            if getattr(step, "_capture_error", None):
                res["distributed"]["capture_fallback"] = step._capture_error[:200]
        if coll_bench:
            res.setdefault("distributed", {})["collective_only"] = coll_bench
        if config3:
            res["config3_ddp"] = config3
        if config4:
            res["config4_ddp"] = config4
        if steady:
            res["steady_state"] = steady
        if extras:
            res["extra_configs"] = extras
        traffic, table = pmc_step_traffic()
        roof = dict(step_roofline(wide, mfma_dtype, waveform, B, ms, T_cfg), kernel="whole step (one hipGraph)")
        roof.update({
            "traffic": traffic if (world == 1 and headline and B == B_PER_GPU) else None,
            "definition": "SURVEY 8(d): reference GEMM-shaped FLOPs of the step (4 x forward: 3.432 GFLOP per clip, wide 14.157) / "
                          "measured time, against the dense MFMA peak of the operand dtype (the step is compute-bound: 286 "
                          "FLOP/B); algorithmic_bytes = block-boundary tensors at the storage width of the mode + optimiser "
                          "words + waveform bytes when the step starts from audio; `traffic` = HBM bytes per step from the "
                          "committed PMC passes of this bench command (profiles/pmc_traffic.json `_per_step`: every dispatch "
                          "of the traced run, 2 x FETCH_SIZE + WRITE_SIZE per MI355X_MICROARCH.md, summed and divided by the "
                          "steps traced - collected offline by tools/collect_profiles.sh, not inside this run)",
            "flop_convention": "effective = the reference's operator FLOPs (direct 3x3 convolution, 64x64 GLU Linear per "
                               "pixel); executed = what the kernel issues on the MFMA pipe (Winograd F(2x2,3x3): 16/36 of a "
                               "direct convolution's multiplies; block 0: one K = 10 convolution to 128 channels). An "
                               "effective fraction above 1 is an algorithmic saving, not MFMA utilisation - the executed "
                               "fraction beside it is.",
        })
        roof = {k: roof[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_of_f32_mfma_peak", "traffic",
                                     "algorithmic_flops", "algorithmic_bytes", "algorithmic_gbs", "hbm_frac", "definition",
                                     "flop_convention")}
        if mfma_dtype != "f32":
            roof["note"] = ("reduced-precision line: priced against the DENSE bf16 MFMA peak as the contract asks.  The operators "
                            "that run on bf16 operands are listed in config.workload; the recurrences, heads, BatchNorm and all "
                            "element-wise work are not MFMA work at all, so this fraction is NOT an MFMA-utilisation figure "
                            "(frac_of_f32_mfma_peak beside it)")
        if not args.no_extras and headline:
            kr = kernel_roofline(step)
            dom = max(kr, key=lambda k: kr[k]["us"] * kr[k]["launches_per_step"])
            roof["dominant_kernel"] = dict(kr[dom], name=dom,
                                           chosen_by="largest (solo launch time x launches per step) of the table below")
            roof["kernels"] = kr
            roof["timing"] = ("whole step: wall clock around the K timed steps (barrier + synchronize on both sides); "
                              "kernel table: HIP events on the launch stream around 20 re-launches of each kernel on the "
                              "step's own buffers (sed_kernel_replay) after the timed region; conv*_wgrad = the Winograd "
                              "wgrad kernel + its ordered partial-sum reduction (the whole operator)")
        res["roofline"] = roof
        if world == 1 and not args.no_extras and headline and not args.trace_only_this_config:
            try:
                res["feature_path"] = feature_path(device)
            except Exception as e:                      # the headline number must not depend on the extra leg
                res["feature_path"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline and headline:
            res["cpu_baseline"] = cpu_baseline()
    if pg is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: whatever RCCL / the runtime still hold in C stdio buffers goes first
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
