"""Per-replay time of the first replays after the driver's warm-up (bench.py --steps 20 --warmup 5 reports ~2.5 % more per step
than a 3 000-step run): is it a one-time cost in the first replays or a slower clock over the whole burst?
    python tools/first_replays.py [warmup] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    dev = torch.device("cuda", 0)
    runner, step, B = bench.make_runner("mt-f32", dev, 0, None, use_graph=True)
    for _ in range(max(W, 3)):
        runner.run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(n):
        runner.run()
        ev[i + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    print(f"warmup {W}: wall {wall * 1e3 / n:.4f} ms/step over {n}; first 20: {sum(ms[:20]) / 20:.4f}; last 20: {sum(ms[-20:]) / 20:.4f}")
    print("per replay (ms):", " ".join(f"{v:.3f}" for v in ms))
    # the same again after a long run
    for _ in range(2000):
        runner.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        runner.run()
    torch.cuda.synchronize()
    print(f"20 steps after 2000 more replays: {(time.perf_counter() - t0) * 1e3 / 20:.4f} ms/step")
    time.sleep(2.0)
    t0 = time.perf_counter()
    for _ in range(20):
        runner.run()
    torch.cuda.synchronize()
    print(f"20 steps after a 2 s idle pause: {(time.perf_counter() - t0) * 1e3 / 20:.4f} ms/step")

if __name__ == "__main__":
    main()
