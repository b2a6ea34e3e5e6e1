"""Run-to-run determinism probe: the same step (same inputs, same state) several times; reports which gradient
tensors / outputs differ bitwise between repetitions and by how much.  Usage (GPU box): python tools/determinism.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth  # noqa: E402
from tests import gpu_util as gu  # noqa: E402
from dcase2019_task4_amd.train import MeanTeacherStep  # noqa: E402


def main():
    B, T = 8, 216
    tgt, wm, sm = synth.make_target(1, B, T // 8)
    s, _ = gu.make_model(0, dropout=0.5)
    t, _ = gu.make_model(1, dropout=0.5)
    s.train(); t.train()
    st = MeanTeacherStep(s, t, B, T, 40, wm, sm, seed=1234, use_graph=os.environ.get("GRAPH", "0") == "1")
    st.load_batch(synth.make_input(60, B, T).cuda(), synth.make_input(70, B, T).cuda(), tgt.cuda())
    if st.use_graph:
        st._warm = 2
    sd = st.state_dict()
    names = [n for n, _ in s.named_parameters()]
    ref = None
    for rep in range(int(os.environ.get("REPS", "12"))):
        st.load_state_dict(sd)
        st.run()
        torch.cuda.synchronize()
        cur = {"grads": st.grads.clone(), "strong": st.strong.clone(), "strong_ema": st.strong_ema.clone(),
               "params": s._flat.clone(), "bn": s._bn_flat.clone(), "bn_t": t._bn_flat.clone()}
        if ref is None:
            ref = cur
            continue
        for k in cur:
            if not torch.equal(cur[k], ref[k]):
                d = (cur[k] - ref[k]).abs()
                msg = f"rep {rep}: {k} differs: max {float(d.max()):.3e}, n {int((d > 0).sum())}"
                if k == "grads":
                    bad = [names[i] for i, (o0, o1, _) in enumerate(s._layout) if not torch.equal(cur[k][o0:o1], ref[k][o0:o1])]
                    msg += f" in {bad}"
                print(msg)
    print("done")


if __name__ == "__main__":
    main()
