"""Front-end kernel timing: sed_mel_frames (persistent STFT + mel) for 64 clips at several workgroup caps, against the two
earlier kernels (debug bits 21 / 19), and sed_logmel_transform.  HIP-event times, 20 launches each."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dcase2019_task4_amd import _lib
from dcase2019_task4_amd.features import FeatureConfig, FeatureExtractor, LogMelTransform

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
fx = FeatureExtractor(FeatureConfig.baseline_16k(), device="cuda")
wave = (0.1 * torch.randn(B, 160000)).cuda()
l = _lib.lib()
for dt in ("f64", "f32"):
    for wgs in (0, 192, 128, 96, 64):
        print(f"k_stft_mel_p<{dt}>  B={B} workgroups={wgs or 'all'}: {timed(lambda: fx.calculate_mel_spec_batch(wave, workgroups=wgs, fft_dtype=dt)):8.1f} us")
a = fx.calculate_mel_spec_batch(wave, fft_dtype="f64"); b = fx.calculate_mel_spec_batch(wave, fft_dtype="f32")
rel = ((a - b).abs() / a.abs().clamp_min(1e-30)).max().item()
print(f"f32 vs f64: max relative difference {rel:.3e}; max |a| {a.abs().max().item():.3e}, max abs diff {(a - b).abs().max().item():.3e}")
for bit, name in ((1 << 21, "k_stft_mel16 (round 3)"), (1 << 19, "k_stft_mel (round 2)")):
    l.sed_debug_set(bit)
    print(f"{name}: {timed(lambda: fx.calculate_mel_spec_batch(wave)):8.1f} us")
l.sed_debug_set(0)
mel = fx.calculate_mel_spec_batch(wave)
tr = LogMelTransform(628, augment_type="noise")
print(f"sed_logmel_transform (noise): {timed(lambda: tr(mel, seed=5)):8.1f} us")
tr0 = LogMelTransform(628)
print(f"sed_logmel_transform (clean): {timed(lambda: tr0(mel)):8.1f} us")
tr32 = LogMelTransform(628, augment_type="noise", math_dtype="f32")
print(f"sed_logmel_transform (noise, fp32 mode): {timed(lambda: tr32(mel, seed=5)):8.1f} us")
tr032 = LogMelTransform(628, math_dtype="f32")
print(f"sed_logmel_transform (clean, fp32 mode): {timed(lambda: tr032(mel)):8.1f} us")
