"""Does a compute-bound kernel run slower right after a synchronize?  (The first dozen step replays after one do: tools/first_replays.py.)
A fixed fp32 GEMM (rocBLAS through torch - measurement tool only, not the product path) timed back to back with events."""
import time
import torch

def burst(a, b, n, tag):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        torch.mm(a, b)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    print(f"{tag}: " + " ".join(f"{v * 1e3:.0f}" for v in ms[:24]) + f" ... last 8 mean {sum(ms[-8:]) / 8 * 1e3:.0f} us")

def main():
    a = torch.randn(4096, 4096, device="cuda")
    b = torch.randn(4096, 4096, device="cuda")
    for _ in range(50):
        torch.mm(a, b)
    burst(a, b, 60, "right after a warm run + synchronize")
    burst(a, b, 60, "again (synchronize only)")
    time.sleep(0.01)
    burst(a, b, 60, "after 10 ms idle")
    time.sleep(1.0)
    burst(a, b, 60, "after 1 s idle")

if __name__ == "__main__":
    main()
