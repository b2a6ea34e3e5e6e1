"""Does a step built late in a process run as fast as one built first?  Builds bench.py's five extra workloads one after the
other IN ONE PROCESS and times each (300 replays); compare with `bench.py --config <c>` alone."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
for name in ("waveform-bf16", "wide-bf16", "wide-bf16x3", "mt-bf16", "mt-bf16x3", "mt-f32"):
    runner, step, B = bench.make_runner(name, dev, 0)
    for _ in range(8):
        runner.run()
    el = bench.time_steps(runner, 300, 1, dev)
    print(f"{name}: {el / 300 * 1e3:.4f} ms/step", flush=True)
    del runner, step
    torch.cuda.empty_cache()
