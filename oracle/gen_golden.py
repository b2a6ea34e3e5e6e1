"""TEST INFRASTRUCTURE ONLY - generates tests/golden/*.npz by IMPORTING the reference.

Run in the build container only (needs /root/reference; the GPU box never runs this):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden.py

The reference's non-arithmetic third-party imports that are absent from this image (librosa,
soundfile, dcase_util, sed_eval, youtube_dl) are stubbed in sys.modules; the only stubbed function
that does arithmetic is ``librosa.amplitude_to_db``, which is bound to the oracle's restatement
(so G6 pins Scaler / PadOrTrunc / ToTensor / Normalize, not the dB formula - that stays "parity
unpinned", see oracle/__init__.py).  Fixtures hold reference OUTPUTS only; inputs and weights are
regenerated from oracle/synth.py seeds.
"""
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference/baseline")
    for n in ["librosa", "soundfile", "dcase_util", "dcase_util.data", "dcase_util.containers",
              "sed_eval", "youtube_dl", "youtube_dl.utils"]:
        sys.modules[n] = types.ModuleType(n)
    from oracle import postprocess_np
    sys.modules["dcase_util.data"].DecisionEncoder = postprocess_np.DecisionEncoder          # restatements (parity
    sys.modules["dcase_util.data"].ProbabilityEncoder = postprocess_np.ProbabilityEncoder    # unpinned): see that file
    sys.modules["dcase_util.containers"].AudioContainer = object
    sys.modules["youtube_dl.utils"].ExtractorError = Exception
    sys.modules["youtube_dl.utils"].DownloadError = Exception
    from oracle import features_np
    sys.modules["librosa"].amplitude_to_db = features_np.amplitude_to_db
    import main  # noqa
    import config as cfg
    from models.CRNN import CRNN
    return main, cfg, CRNN


def load_params(model, params, bn=None):
    import torch
    sd = dict(model.named_parameters())
    with torch.no_grad():
        for k, v in params.items():
            sd[k].copy_(v)
        if bn is not None:
            bufs = dict(model.named_buffers())
            for k, v in bn.items():
                bufs[k].copy_(v)


def synth_bn(seed, nb=(64, 64, 64)):
    import torch
    rs = np.random.RandomState(5000 + seed)
    st = {}
    for i, c in enumerate(nb):
        st[f"cnn.cnn.batchnorm{i}.running_mean"] = torch.tensor(rs.normal(0, 0.2, c), dtype=torch.float32)
        st[f"cnn.cnn.batchnorm{i}.running_var"] = torch.tensor(rs.uniform(0.5, 1.5, c), dtype=torch.float32)
    return st


def main_():
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    main, cfg, CRNN = import_reference()
    from oracle import synth, features_np, ref_cpu

    kw = dict(cfg.crnn_kwargs)
    kw0 = dict(kw)
    kw0["dropout"] = 0

    # ---- G1/G2: eval-mode posteriors (+ a few intermediate slices), T=628 and T=864 ----------
    for T in (628, 864):
        m = CRNN(**kw)
        load_params(m, synth.make_params(0), synth_bn(0))
        m.eval()
        x = synth.make_input(T, 2, T)
        inter = {}
        hooks = []
        for name in ["conv0", "batchnorm0", "glu0", "pooling0", "pooling1", "pooling2"]:
            mod = getattr(m.cnn.cnn, name)
            hooks.append(mod.register_forward_hook(lambda _m, _i, o, n=name: inter.__setitem__(n, o.detach())))
        with torch.no_grad():
            strong, weak = m(x)
            gru = m.rnn(m.cnn(x).squeeze(-1).permute(0, 2, 1))
        for h in hooks:
            h.remove()
        np.savez_compressed(
            os.path.join(OUT, f"g1_eval_T{T}.npz"),
            strong=strong.numpy(), weak=weak.numpy(),
            conv0_s=inter["conv0"][:, :, :5, :7].numpy(), bn0_s=inter["batchnorm0"][:, :, :5, :7].numpy(),
            glu0_s=inter["glu0"][:, :, :5, :7].numpy(), pool0_s=inter["pooling0"][:, :, :6, :].numpy(),
            pool1_s=inter["pooling1"][:, :, :8, :].numpy(), pool2=inter["pooling2"].numpy(),
            gru=gru.numpy())

    # ---- G3: train-mode forward, dropout = 0, BN running stats after 2 passes ----------------
    m = CRNN(**kw0)
    load_params(m, synth.make_params(0))
    m.train()
    outs = []
    with torch.no_grad():
        for it in range(2):
            x = synth.make_input(10 + it, 4, 628)
            outs.append(m(x))
    bufs = {k.replace(".", "_"): v.numpy() for k, v in m.named_buffers()}
    np.savez_compressed(os.path.join(OUT, "g3_train_fwd.npz"),
                        strong0=outs[0][0].numpy(), weak0=outs[0][1].numpy(),
                        strong1=outs[1][0].numpy(), weak1=outs[1][1].numpy(), **bufs)

    # ---- G4/G5: three steps of the real main.train -------------------------------------------
    B, T = 8, 628
    student = CRNN(**kw0)
    teacher = CRNN(**kw0)
    load_params(student, synth.make_params(0))
    load_params(teacher, synth.make_params(1))
    for p in teacher.parameters():
        p.detach_()
    student.train()
    teacher.train()

    rec = {"grads": [], "meters": []}

    class RecAdam(torch.optim.Adam):
        def step(self, closure=None):
            rec["grads"].append([p.grad.detach().clone() for g in self.param_groups for p in g["params"]])
            return super().step(closure)

    class RecMeters(main.AverageMeterSet):
        def update(self, name, value, n=1):
            rec["meters"].append((name, float(value)))
            return super().update(name, value, n)

    main.AverageMeterSet = RecMeters
    opt = RecAdam(filter(lambda p: p.requires_grad, student.parameters()), lr=0.001, betas=(0.9, 0.999))
    batches = []
    for it in range(3):
        x = synth.make_input(20 + it, B, T)
        xe = synth.make_input(30 + it, B, T)
        tgt, wm, sm = synth.make_target(it, B, T // 8)
        batches.append((x, xe, tgt))
    main.train(batches, student, opt, 0, ema_model=teacher, weak_mask=wm, strong_mask=sm)

    names = [n for n, _ in student.named_parameters()]
    save = {}
    for it in range(3):
        for n, g in zip(names, rec["grads"][it]):
            key = n.replace(".", "_")
            save[f"s{it}_gnorm_{key}"] = np.array(float(g.double().norm()))
            save[f"s{it}_ghead_{key}"] = g.flatten()[:16].numpy()
    mnames = sorted(set(n for n, _ in rec["meters"]))
    for mn in mnames:
        save["meter_" + mn.replace(" ", "_")] = np.array([v for n, v in rec["meters"] if n == mn])
    for n, p in student.named_parameters():
        save["pS_sum_" + n.replace(".", "_")] = np.array(float(p.detach().double().sum()))
        save["pS_head_" + n.replace(".", "_")] = p.detach().flatten()[:16].numpy()
    for n, p in teacher.named_parameters():
        save["pT_sum_" + n.replace(".", "_")] = np.array(float(p.detach().double().sum()))
        save["pT_head_" + n.replace(".", "_")] = p.detach().flatten()[:16].numpy()
    for n, b in student.named_buffers():
        save["bS_" + n.replace(".", "_")] = b.numpy()
    for n, b in teacher.named_buffers():
        save["bT_" + n.replace(".", "_")] = b.numpy()
    np.savez_compressed(os.path.join(OUT, "g5_train3.npz"), **save)

    # ---- G6: Scaler + transform chain (reference DataLoad/Scaler code, oracle dB formula) ----
    from utils.Scaler import Scaler
    from utils.utils import get_transforms
    rs = np.random.RandomState(77)
    clips = [np.abs(rs.standard_normal((n, 64))).astype(np.float32) * 3.0 for n in (628, 600, 650, 628)]
    labels = [np.zeros((78, 10), dtype=np.float32) for _ in clips]
    tr0 = get_transforms(628)
    pre = [tr0((c, l)) for c, l in zip(clips, labels)]
    sc = Scaler()
    sc.calculate_scaler([(p[0], p[1]) for p in pre])
    tr = get_transforms(628, sc, augment_type="noise")
    np.random.seed(123)
    outs = [tr((c, l)) for c, l in zip(clips, labels)]
    np.random.seed(123)
    noises = [np.abs(np.random.normal(0, 0.5 ** 2, c.shape)) for c in clips]
    trv = get_transforms(628, sc)
    outv = [trv((c, l)) for c, l in zip(clips, labels)]
    sel = np.r_[0:628:9, 590:628]          # subsampled frames + the zero-pad boundary region
    np.savez_compressed(os.path.join(OUT, "g6_transforms.npz"),
                        mean=sc.mean_, mean_of_square=sc.mean_of_square_, std=sc.std_, sel=sel,
                        clean=np.stack([o[0].numpy()[:, sel] for o in outs]),
                        noisy=np.stack([o[1].numpy()[:, sel] for o in outs]),
                        valid=np.stack([o[0].numpy()[:, sel] for o in outv]),
                        noise_head=np.stack([n[:4] for n in noises]))

    # ---- G7: sigmoid_rampup table ------------------------------------------------------------
    from utils import ramps
    cur = np.array([0, 1, 10, 100, 5249, 10499, 10500, 20000], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "g7_rampup.npz"), current=cur,
                        value=np.array([ramps.sigmoid_rampup(c, 10500) for c in cur]),
                        value0=np.array([ramps.sigmoid_rampup(c, 0) for c in cur]))
    print("wrote", sorted(os.listdir(OUT)))


def supervised_():
    """G8: three steps of the REAL baseline/main_simple_CRNN.py train() (config 1 of BASELINE.json: the supervised
    CRNN, weak + strong BCE only, no teacher), B=8, T=628, dropout 0."""
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main, cfg, CRNN = import_reference()
    import main_simple_CRNN as simple
    from oracle import synth
    kw0 = dict(cfg.crnn_kwargs)
    kw0["dropout"] = 0
    B, T = 8, 628
    model = CRNN(**kw0)
    load_params(model, synth.make_params(0))
    model.train()
    rec = {"grads": [], "meters": []}

    class RecAdam(torch.optim.Adam):
        def step(self, closure=None):
            rec["grads"].append([p.grad.detach().clone() for g in self.param_groups for p in g["params"]])
            return super().step(closure)

    class RecMeters(simple.AverageMeterSet):
        def update(self, name, value, n=1):
            rec["meters"].append((name, float(value)))
            return super().update(name, value, n)

    simple.AverageMeterSet = RecMeters
    opt = RecAdam(filter(lambda p: p.requires_grad, model.parameters()), lr=0.001, betas=(0.9, 0.999))
    batches = []
    for it in range(3):
        x = synth.make_input(20 + it, B, T)
        tgt, _, _ = synth.make_target(it, B, T // 8)
        tgt = tgt.clamp(min=0)            # main_simple_CRNN has no unlabeled rows: every clip carries labels
        batches.append((x, tgt))
    wm, sm = slice(B // 2), slice(B // 2, B)      # main_simple_CRNN.py:185-186
    simple.train(batches, model, opt, 0, weak_mask=wm, strong_mask=sm)
    save = {}
    names = [n for n, _ in model.named_parameters()]
    for it in range(3):
        for n, g in zip(names, rec["grads"][it]):
            key = n.replace(".", "_")
            save[f"s{it}_gnorm_{key}"] = np.array(float(g.double().norm()))
            save[f"s{it}_ghead_{key}"] = g.flatten()[:16].numpy()
    for mn in sorted(set(n for n, _ in rec["meters"])):
        save["meter_" + mn.replace(" ", "_")] = np.array([v for n, v in rec["meters"] if n == mn])
    for n, p in model.named_parameters():
        save["p_sum_" + n.replace(".", "_")] = np.array(float(p.detach().double().sum()))
        save["p_head_" + n.replace(".", "_")] = p.detach().flatten()[:16].numpy()
    for n, b in model.named_buffers():
        save["b_" + n.replace(".", "_")] = b.numpy()
    np.savez_compressed(os.path.join(OUT, "g8_supervised3.npz"), **save)
    print("wrote g8_supervised3.npz", sorted(k for k in save if k.startswith("meter_")))


def predictions_():
    """G9: the REAL baseline/evaluation_measures.get_predictions + ManyHotEncoder.decode_strong on 6 synthetic clips
    (eval-mode reference CRNN, T = 628 -> 78 frames): the event table, its TSV text and the filtered decisions."""
    import io
    import pandas as pd
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main, cfg, CRNN = import_reference()
    import evaluation_measures as em
    from utils.utils import ManyHotEncoder
    from oracle import synth, postprocess_np
    if not hasattr(pd.DataFrame, "append"):          # removed in pandas 2.0; the reference still calls it (:221)
        pd.DataFrame.append = lambda self, other: pd.concat([self, other])
    N, T = 6, 628
    post = synth.make_posteriors(0, N, T // 8)
    x = synth.make_input(77, N, T)

    class FixedPosteriors(torch.nn.Module):
        """Stands in for the CRNN: get_predictions only needs model(input[None]) -> (strong, weak).  It returns the
        prescribed posteriors of the clip it is called for (identified by its input), so that the REAL reference loop
        runs its binarization / median filter / decode / DataFrame code on data with real temporal structure."""

        def forward(self, inp):
            i = int(torch.nonzero((x.reshape(N, -1) == inp.reshape(1, -1)).all(dim=1))[0])
            return post[i:i + 1], post[i:i + 1].mean(1)

    m = FixedPosteriors()

    class DS:
        filenames = pd.Series([f"clip_{i}.wav" for i in range(N)])

        def __len__(self):
            return N

        def __getitem__(self, i):
            return x[i], torch.zeros(T // 8, 10)

        def __iter__(self):
            return (self[i] for i in range(N))

    labels = ["Alarm_bell_ringing", "Blender", "Cat", "Dishes", "Dog", "Electric_shaver_toothbrush", "Frying",
              "Running_water", "Speech", "Vacuum_cleaner"]
    enc = ManyHotEncoder(labels, n_frames=T // 8)
    strong = post.numpy()
    margin = float(np.abs(strong - 0.5)[np.abs(strong - 0.5) > 0].min())
    assert margin > 1e-4, margin
    buf = os.path.join("/tmp", "g9_pred.tsv")
    df = em.get_predictions(m, DS(), enc.decode_strong, cfg.pooling_time_ratio, save_predictions=buf)
    tsv = open(buf).read()
    dec = np.stack([postprocess_np.filter_decisions(s) for s in strong]).astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "g9_predictions.npz"), decisions=dec, tsv=np.array(tsv),
                        labels=np.array(labels), onset=df.onset.to_numpy(dtype=np.float64),
                        offset=df.offset.to_numpy(dtype=np.float64), event_label=np.array(df.event_label.tolist()),
                        filename=np.array(df.filename.tolist()), margin=np.array(margin),
                        sample_rate=np.array(cfg.sample_rate), hop_length=np.array(cfg.hop_length),
                        median_window=np.array(cfg.median_window), pooling_time_ratio=np.array(cfg.pooling_time_ratio))
    print("wrote g9_predictions.npz:", len(df), "events, margin", margin)


def wide_():
    """G10 (BASELINE.json configs[4]): the REAL reference CRNN built with nb_filters = [128] * 3, n_RNN_cell = 256
    (baseline/models/CNN.py:35-67 and CRNN.py:12-31 are shape-generic): eval posteriors at T = 628, a train-mode forward
    with its BatchNorm buffers, and two steps of the real main.train (B = 8, T = 216, dropout 0)."""
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main, cfg, CRNN = import_reference()
    from oracle import synth
    C, H = 128, 256
    kw = dict(cfg.crnn_kwargs)
    kw.update(nb_filters=[C] * 3, n_RNN_cell=H)
    kw0 = dict(kw, dropout=0)
    mk = dict(nb_filters=(C,) * 3, n_RNN_cell=H)
    save = {}
    # eval
    m = CRNN(**kw)
    load_params(m, synth.make_params(0, **mk), synth_bn(0, (C,) * 3))
    m.eval()
    with torch.no_grad():
        s, w = m(synth.make_input(628, 2, 628))
    save["eval_strong"], save["eval_weak"] = s.numpy(), w.numpy()
    # train-mode forward, dropout 0, two passes
    m = CRNN(**kw0)
    load_params(m, synth.make_params(0, **mk))
    m.train()
    with torch.no_grad():
        for it in range(2):
            s, w = m(synth.make_input(10 + it, 4, 216))
            save[f"train_strong{it}"], save[f"train_weak{it}"] = s.numpy(), w.numpy()
    for k, v in m.named_buffers():
        save["tb_" + k.replace(".", "_")] = v.numpy()
    # two steps of the real main.train
    B, T = 8, 216
    student, teacher = CRNN(**kw0), CRNN(**kw0)
    load_params(student, synth.make_params(0, **mk))
    load_params(teacher, synth.make_params(1, **mk))
    for p in teacher.parameters():
        p.detach_()
    student.train(); teacher.train()
    rec = {"grads": [], "meters": []}

    class RecAdam(torch.optim.Adam):
        def step(self, closure=None):
            rec["grads"].append([p.grad.detach().clone() for g in self.param_groups for p in g["params"]])
            return super().step(closure)

    class RecMeters(main.AverageMeterSet):
        def update(self, name, value, n=1):
            rec["meters"].append((name, float(value)))
            return super().update(name, value, n)

    main.AverageMeterSet = RecMeters
    opt = RecAdam(filter(lambda p: p.requires_grad, student.parameters()), lr=0.001, betas=(0.9, 0.999))
    batches = []
    for it in range(2):
        tgt, wm, sm = synth.make_target(it, B, T // 8)
        batches.append((synth.make_input(20 + it, B, T), synth.make_input(30 + it, B, T), tgt))
    main.train(batches, student, opt, 0, ema_model=teacher, weak_mask=wm, strong_mask=sm)
    names = [n for n, _ in student.named_parameters()]
    for it in range(2):
        for n, g in zip(names, rec["grads"][it]):
            key = n.replace(".", "_")
            save[f"s{it}_gnorm_{key}"] = np.array(float(g.double().norm()))
            save[f"s{it}_ghead_{key}"] = g.flatten()[:16].numpy()
    for mn in sorted(set(n for n, _ in rec["meters"])):
        save["meter_" + mn.replace(" ", "_")] = np.array([v for n, v in rec["meters"] if n == mn])
    for n, p in student.named_parameters():
        save["pS_sum_" + n.replace(".", "_")] = np.array(float(p.detach().double().sum()))
        save["pS_head_" + n.replace(".", "_")] = p.detach().flatten()[:16].numpy()
    for n, p in teacher.named_parameters():
        save["pT_sum_" + n.replace(".", "_")] = np.array(float(p.detach().double().sum()))
        save["pT_head_" + n.replace(".", "_")] = p.detach().flatten()[:16].numpy()
    np.savez_compressed(os.path.join(OUT, "g10_wide.npz"), **save)
    print("wrote g10_wide.npz", len(save), "arrays")


VARIANTS = {   # G11: constructor variants OUTSIDE the hot path (served with stock torch operators)
    "relu_mean": dict(attention=False, activation="Relu", n_RNN_cell=64, n_layers_RNN=1),          # the CLASS DEFAULTS (CRNN.py:12-13)
    "leaky_att": dict(attention=True, activation="leakyrelu", n_RNN_cell=32, n_layers_RNN=2),
    "cg_att": dict(attention=True, activation="cg", n_RNN_cell=64, n_layers_RNN=1, nb_filters=[32, 48, 64]),
    "glu_mean_pool14": dict(attention=False, activation="glu", n_RNN_cell=64, n_layers_RNN=1, pooling=[(1, 4), (1, 4), (1, 4)]),
}


def variants_():
    """G11: the REAL reference CRNN built with constructor arguments outside the hot path (activation relu / leakyrelu / cg,
    attention = False -> weak = strong.mean(1), other widths / pooling / cell counts): eval-mode posteriors with synthetic running
    statistics, and a train-mode forward (dropout 0) with the gradient norms of a simple loss."""
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main, cfg, CRNN = import_reference()
    from oracle import synth
    out = {}
    for k, (tag, kw) in enumerate(VARIANTS.items()):      # seeds: parameters k, BatchNorm buffers 30 + k, input 300 + k
        full = dict(n_in_channel=1, nclass=10, dropout=0, kernel_size=3 * [3], padding=3 * [1], stride=3 * [1],
                    nb_filters=[64, 64, 64], pooling=list(3 * ((2, 4),)))
        full.update(kw)
        m = CRNN(**full)
        shapes = [(n, tuple(p.shape)) for n, p in m.named_parameters()]
        params = synth.make_params_for(shapes, seed=k)
        bn = synth_bn(30 + k, nb=full["nb_filters"])
        load_params(m, params, bn)
        T = 64 if full["pooling"][0][0] == 2 else 16
        x = synth.make_input(300 + k, 3, T)
        m.eval()
        with torch.no_grad():
            s, w = m(x)
        out[f"{tag}_eval_strong"] = s.numpy(); out[f"{tag}_eval_weak"] = w.numpy()
        m.train()
        s, w = m(x)
        loss = (s * s).mean() + w.sum()
        loss.backward()
        out[f"{tag}_train_strong"] = s.detach().numpy(); out[f"{tag}_train_weak"] = w.detach().numpy()
        out[f"{tag}_grad_norms"] = np.array([float(p.grad.double().norm()) for _, p in m.named_parameters()])
        out[f"{tag}_param_names"] = np.array([n for n, _ in m.named_parameters()])
        out[f"{tag}_bn_mean0"] = dict(m.named_buffers())["cnn.cnn.batchnorm0.running_mean"].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g11_variants.npz"), **out)
    print("g11 written:", sorted(out)[:6], "...")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g11":
        variants_()
    elif len(sys.argv) > 1 and sys.argv[1] == "g10":
        wide_()
    elif len(sys.argv) > 1 and sys.argv[1] == "g9":
        predictions_()
    elif len(sys.argv) > 1 and sys.argv[1] == "g8":
        supervised_()
    else:
        main_()
        supervised_()
        predictions_()
        wide_()
