import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import torch, bench
from dcase2019_task4_amd.train import MeanTeacherStep
dev = torch.device("cuda", 0)
for kw in (dict(mfma_dtype="bf16"), dict(nb_filters=[128]*3, n_RNN_cell=256, mfma_dtype="f32")):
    for teacher in (False, True):
        s, t = bench.build_models(dev, 0, **kw)
        x, xe, tgt, wm, sm = bench.synthetic_batch(8, 216, 1, dev)
        st = MeanTeacherStep(s, t if teacher else None, 8, 216, 100, wm, sm, use_graph=True, overlap_streams=True)
        st.load_batch(x, xe, tgt)
        for i in range(4):
            print("cfg", kw, "teacher", teacher, "step", i, flush=True)
            st.run()
            torch.cuda.synchronize()
print("done")
