"""Does a step built late in a process run as fast as one built first?  Builds bench.py's workloads one after the other IN ONE
PROCESS and times each (300 replays); compare with `bench.py --config <c>` alone.
  default     the process-wide stream pool (train.shared_stream): every step reuses the same four streams
  --no-pool   every step owns its streams (pool_streams=False) and releases the library's helper streams in step.close()
              (sed_stream_release): the pool is then an optimisation, not a requirement
  --leak      own streams and NO close(): what rounds 1-3 did (helper streams pile up; the later steps slow down)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
mode = "no-pool" if "--no-pool" in sys.argv else ("leak" if "--leak" in sys.argv else "pool")
print(f"mode: {mode}", flush=True)
for name in ("waveform-bf16", "wide-bf16", "wide-bf16x3", "mt-bf16", "mt-bf16x3", "mt-f32"):
    runner, step, B = bench.make_runner(name, dev, 0, pool_streams=(mode == "pool"))
    for _ in range(8):
        runner.run()
    el = bench.time_steps(runner, 300, 1, dev)
    print(f"{name}: {el / 300 * 1e3:.4f} ms/step", flush=True)
    if mode == "no-pool":
        step.close()
    del runner, step
    torch.cuda.empty_cache()
