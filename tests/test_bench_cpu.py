"""CPU-side checks of bench.py's bookkeeping (no GPU): the algorithmic-byte model per arithmetic mode, the
roofline block of a measured line, the workload strings."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def test_algorithmic_bytes_follow_the_storage_width_of_the_mode():
    import bench
    opt_base, opt_wide = 9 * 4 * 214356, 9 * 4 * 2132628              # 9 words x parameters per step (7.7 MB / 76.8 MB)
    assert bench.algorithmic_bytes(False, "f32", False, 24) == 24 * 12.0e6 + opt_base
    assert bench.algorithmic_bytes(False, "bf16x3", False, 24) == 24 * 12.0e6 + opt_base       # split operands, fp32 storage
    assert bench.algorithmic_bytes(False, "bf16", False, 24) == 24 * 6.0e6 + opt_base
    assert bench.algorithmic_bytes(True, "f32", False, 24) == 24 * 23.9e6 + opt_wide
    assert bench.algorithmic_bytes(True, "bf16", False, 24) == 24 * 12.0e6 + opt_wide
    wav = 160000 * 4 + 2 * 628 * 64 * 4                                # fp32 waveform read + two feature tensors written
    assert bench.algorithmic_bytes(False, "bf16", True, 64) == 64 * (6.0e6 + wav) + opt_base


def test_step_roofline_prices_against_the_peak_of_the_operand_dtype():
    import bench
    r = bench.step_roofline(False, "f32", False, 24, 0.7476)
    assert r["peak"] == 157.3 and abs(r["frac"] - 24 * 3.432e9 / 0.7476e-3 / 157.3e12) < 1e-3
    assert abs(r["frac"] - r["frac_of_f32_mfma_peak"]) < 1e-9
    w = bench.step_roofline(True, "bf16", False, 24, 1.31)
    assert w["peak"] == 2500.0 and w["algorithmic_flops"] == int(14.157e9 * 24)
    assert w["frac"] < w["frac_of_f32_mfma_peak"]


def test_workload_strings_name_what_runs_on_which_operands():
    import bench
    s = bench.workload_string(False, "f32", False, 24)
    assert "block 0's backward sums on split bf16 operands" in s and "batch 24" in s
    s = bench.workload_string(False, "bf16", True, 64, "f32")
    assert "raw 16 kHz waveforms" in s and "f32 butterflies" in s and "SED_DTYPE_BF16" in s
    assert "SED_DTYPE_BF16X3" in bench.workload_string(True, "bf16x3", False, 24)


def test_round6_configs_frames_and_strict_strings():
    import bench
    for name in ("mt-f32-b64", "mt-f32-T864", "mt-f32-strict"):
        assert name in bench.CONFIGS and bench.CONFIGS[name][1] == "f32"
    assert bench.CONFIGS["mt-f32-b64"][3] == 64                       # configs[3]'s per-rank shape: [16 | 32 | 16]
    assert bench.CONFIG_FRAMES["mt-f32-T864"] == 864                  # baseline/config.py:17-22 (44.1 kHz, hop 511)
    assert bench.block_boundary_elements(628) == 428440               # S
    fs, bs = bench.frames_scale(864)
    assert abs(fs - 1180.8 / 858.1) < 1e-12 and 1.37 < bs < 1.38
    r = bench.step_roofline(False, "f32", False, 24, 0.93, 864)
    assert r["algorithmic_flops"] == int(3.432e9 * 24 * fs)
    s = bench.workload_string(False, "f32", False, 24, T=864)
    assert "[24,1,864,64]" in s
    s = bench.workload_string(False, "f32", False, 24, strict=True)
    assert "STRICT fp32" in s and "split" in s
    assert bench.strict_f32("mt-f32-strict") and not bench.strict_f32("mt-f32")
