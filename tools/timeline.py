"""One step's kernel timeline from a rocprofv3 --kernel-trace run (rocpd .db): start offset, duration, queue.
Usage: python tools/timeline.py <dir with the .db> [step index from the end, default 2] [--config <bench.py config>]
Refuses a trace directory with more than one database (= more than one traced process: round 4's "mt-f32" timeline was a child
process's bf16x3 step) and, with --config, a step whose kernel set is not that workload's."""
import glob
import os
import sqlite3
import sys

# kernels a workload's step must / must not contain (substrings of the kernel names)
KERNEL_SETS = {
    "mt-f32": (("k_conv_wino", "k_wgrad_wino", "k_glu_pool_fwd", "k_gru4_"), ("k_bconv", "k_bglu", "k_gconv", "k_grec", "k_gclu", "k_stft")),
    "mt-f32-b64": (("k_conv_wino", "k_wgrad_wino", "k_glu_pool_fwd", "k_gru4_"), ("k_bconv", "k_bglu", "k_gconv", "k_grec", "k_gclu", "k_stft")),
    "mt-f32-T864": (("k_conv_wino", "k_wgrad_wino", "k_glu_pool_fwd", "k_gru4_"), ("k_bconv", "k_bglu", "k_gconv", "k_grec", "k_gclu", "k_stft")),
    "mt-f32-strict": (("k_conv_wino", "k_wgrad_wino", "k_glu_pool_fwd", "k_gru4_"), ("k_bconv", "k_bglu", "k_gconv", "k_grec", "k_gclu", "k_stft")),
    "mt-bf16": (("k_bconv", "k_bglu_fwd", "k_gwgrad_bf16", "k_gru4_"), ("k_conv_wino", "k_grec", "k_gclu", "k_stft")),
    "mt-bf16x3": (("k_bconv", "k_gru4_"), ("k_conv_wino", "k_grec", "k_gclu", "k_stft")),
    "waveform-bf16": (("k_bconv", "k_gru4_", "k_stft_mel_p", "k_logmel"), ("k_conv_wino", "k_grec", "k_gclu")),
    "mt-f16": (("k_bconv<2", "k_bglu_fwd", "k_gru4_"), ("k_conv_wino", "k_grec", "k_gclu", "k_stft")),
    "waveform-f16": (("k_bconv<2", "k_gru4_", "k_stft_mel_p", "k_logmel"), ("k_conv_wino", "k_grec", "k_gclu")),
    "wide-f16": (("k_bconv<2", "k_grec_"), ("k_gru4_", "k_gclu", "k_stft")),
    "wide-f32": (("k_gclu_",), ("k_gru4_", "k_grec", "k_stft")),
    "wide-bf16": (("k_bconv", "k_grec_"), ("k_gru4_", "k_gclu", "k_stft")),
    "wide-bf16x3": (("k_bconv",), ("k_gru4_", "k_stft")),
}


def main():
    argv = list(sys.argv[1:])
    config = None
    if "--config" in argv:
        i = argv.index("--config")
        config = argv[i + 1]
        del argv[i:i + 2]
    dbs = glob.glob(os.path.join(argv[0], "**", "*.db"), recursive=True)
    if len(dbs) != 1:
        sys.exit(f"timeline.py: {len(dbs)} databases under {argv[0]} - the trace must hold exactly one process "
                 "(bench.py --trace-only-this-config / --no-extras)")
    back = int(argv[1]) if len(argv) > 1 else 2
    cur = sqlite3.connect(dbs[0]).cursor()
    rows = list(cur.execute("select name, start, end, queue_id, stream_id from kernels order by start"))
    # a step ends with its Adam + EMA kernel (the step state is advanced inside the loss / heads-backward kernel)
    adv = [i for i, r in enumerate(rows) if "k_adam_ema" in r[0]]
    a, b = adv[-back - 1] + 1, adv[-back] + 1
    names = [r[0] for r in rows[a:b]]
    if config is not None:
        need, never = KERNEL_SETS[config]
        missing = [k for k in need if not any(k in n for n in names)]
        alien = [k for k in never if any(k in n for n in names)]
        if missing or alien:
            sys.exit(f"timeline.py: this is not a {config} step (missing {missing}, foreign {alien})")
    t0 = rows[a][1]
    print(f"step of {b - a} kernels, {(rows[b - 1][2] - t0) / 1e3:.1f} us" + (f"  [{config}]" if config else ""))
    busy_end = t0
    for name, s, e, q, st in rows[a:b]:
        gap = (s - busy_end) / 1e3
        busy_end = max(busy_end, e)
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q} {'idle %.1f' % gap if gap > 1.0 else '':10s} {name.split('(')[0][:60]}")


if __name__ == "__main__":
    main()
