"""Host time of one step launch (hipGraphLaunch through MeanTeacherStep.run) against the step's GPU time: is the replay loop ever
host-bound?   python tools/host_launch_cost.py [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "mt-f32"
    dev = torch.device("cuda", 0)
    runner, step, B = bench.make_runner(cfg, dev, 0, None, use_graph=True)
    for _ in range(30):
        runner.run()
    torch.cuda.synchronize()
    # (a) host time per call while the queue is short (first calls after a synchronize) and in a long run
    ts = []
    t_prev = time.perf_counter()
    for i in range(200):
        runner.run()
        t = time.perf_counter()
        ts.append(t - t_prev)
        t_prev = t
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    print(f"[{cfg}] host time per run() call, us: first 10: " + " ".join(f"{v * 1e6:.0f}" for v in ts[:10]))
    print(f"[{cfg}] calls 10-50 mean {sum(ts[10:50]) / 40 * 1e6:.0f} us; calls 150-200 mean {sum(ts[150:]) / 50 * 1e6:.0f} us (a full queue blocks the host: then this is the GPU time)")
    # (b) GPU time: a long run
    t0 = time.perf_counter()
    for _ in range(2000):
        runner.run()
    torch.cuda.synchronize()
    print(f"[{cfg}] 2000 replays: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per step")

if __name__ == "__main__":
    main()
