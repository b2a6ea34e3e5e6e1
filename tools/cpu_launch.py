import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from dcase2019_task4_amd.train import MeanTeacherStep
dev = torch.device("cuda", 0)
student, teacher = bench.build_models(dev, 0)
x, xe, tgt, wm, sm = bench.synthetic_batch(bench.B_PER_GPU, bench.T_FRAMES, 1000, dev)
step = MeanTeacherStep(student, teacher, bench.B_PER_GPU, bench.T_FRAMES, 10500, wm, sm, use_graph=True)
step.load_batch(x, xe, tgt)
for _ in range(10): step.run()
torch.cuda.synchronize()
for n in (1, 5, 50):
    t0 = time.perf_counter()
    for _ in range(n): step.run()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n}: cpu enqueue {1e3*(t1-t0)/n:.3f} ms/step, total {1e3*(t2-t0)/n:.3f} ms/step")
