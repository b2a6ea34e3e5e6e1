import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dcase2019_task4_amd.features import FeatureConfig, FeatureExtractor
from dcase2019_task4_amd import _lib
fx = FeatureExtractor(FeatureConfig.baseline_16k(), device="cuda")
wave = (0.1*torch.randn(64,160000)).cuda()
for _ in range(3): fx.calculate_mel_spec_batch(wave)
torch.cuda.synchronize()
l=_lib.lib(); n=1024*16; buf=(C.c_ulonglong*n)()
fn=l.sed_debug_ts_feat; fn.argtypes=[C.POINTER(C.c_ulonglong), C.c_int]; fn.restype=C.c_int
assert fn(buf,n)==0
ts=np.frombuffer(buf,dtype=np.uint64).reshape(1024,16).astype(np.int64); ts=ts[ts[:,0]>0]
print(len(ts),"wgs")
for k in range(1,7):
    d=ts[:,k]-ts[:,k-1]; print(f"stamp {k-1}->{k}: mean {d.mean():9.1f} min {d.min()} max {d.max()}")
if (ts[:,9]>0).all():
    cyc=(ts[:,7]-ts[:,0]).astype(float); wall=(ts[:,9]-ts[:,8]).astype(float)/100.0
    print(f"frame loop: {cyc.mean():.0f} shader cycles in {wall.mean():.1f} us of wall clock = {cyc.mean()/wall.mean()/1e3:.3f} GHz")
    pro=(ts[:,8]-ts[:,10]).astype(float)/100.0
    t0=ts[:,10].min(); end=(ts[:,9]-t0).astype(float)/100.0; start=(ts[:,10]-t0).astype(float)/100.0
    print(f"prologue (entry -> loop): mean {pro.mean():.1f} us max {pro.max():.1f}; workgroup start offsets: mean {start.mean():.1f} max {start.max():.1f} us; last loop end {end.max():.1f} us after the first entry")
