"""-m gpu tests of the batched inference + post-processing path through the C-ABI:
sed_postprocess against the oracle (scipy median filter + restated dcase_util decode) and the drop-in
get_predictions against the fixture produced by the REAL reference get_predictions (bit-exact: integer work)."""
import io
import os

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import postprocess_np as pp
from oracle import synth
from tests import gpu_util as gu

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


class _DS:
    def __init__(self, x):
        self.x = x
        self.filenames = pd.Series([f"clip_{i}.wav" for i in range(len(x))])

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        return self.x[i], torch.zeros(1)


class _Enc:
    def __init__(self, labels):
        self.labels = labels

    def decode_strong(self, m):
        return pp.decode_strong(m, self.labels)


@pytest.mark.parametrize("N,T,NC,win", [(6, 78, 10, 5), (3, 108, 10, 5), (5, 130, 4, 3), (2, 64, 1, 7), (4, 200, 16, 4),
                                        (1, 1, 3, 5), (2, 5, 2, 9), (257, 78, 10, 5)])
def test_postprocess_kernel_vs_oracle(N, T, NC, win):
    """Bit-exact decisions and (onset, offset) lists; includes columns shorter than the window (reflect wraps more than
    once), an even window, one-frame clips and a batch larger than any single-clip loop would see."""
    from dcase2019_task4_amd.inference import postprocess
    post = synth.make_posteriors(N + T, N, T, NC) if T >= 40 else torch.tensor(
        np.random.RandomState(T).uniform(size=(N, T, NC)), dtype=torch.float32)
    cnt, pairs, binary = postprocess(post.cuda(), 0.5, win, want_binary=True)
    cnt, pairs, binary = cnt.cpu().numpy(), pairs.cpu().numpy(), binary.cpu().numpy()
    for i in range(N):
        want = pp.filter_decisions(post[i].numpy(), 0.5, win)
        np.testing.assert_array_equal(binary[i], want)
        for c in range(NC):
            regions = pp.DecisionEncoder().find_contiguous_regions(want[:, c])
            assert cnt[i, c] == len(regions)
            np.testing.assert_array_equal(pairs[i, c, :len(regions)], regions)


def test_get_predictions_matches_the_reference_event_table_byte_for_byte(tmp_path):
    """G9: same prescribed posteriors, same TSV text as the REAL evaluation_measures.get_predictions wrote."""
    from dcase2019_task4_amd.inference import get_predictions
    g = np.load(os.path.join(HERE, "golden", "g9_predictions.npz"))
    N, T = 6, 628
    post = synth.make_posteriors(0, N, T // 8).cuda()
    x = synth.make_input(77, N, T)

    class Fixed(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.dummy = torch.nn.Parameter(torch.zeros(1))
            self.pos = 0

        def forward(self, inp):
            k = inp.shape[0]
            out = post[self.pos:self.pos + k]
            self.pos += k
            return out, out.mean(1)

    labels = [str(v) for v in g["labels"]]
    for decoder in (_Enc(labels).decode_strong, lambda m: pp.decode_strong(m, labels)):     # device decode / host decoder
        path = str(tmp_path / "pred.tsv")
        df = get_predictions(Fixed().cuda(), _DS(x), decoder, int(g["pooling_time_ratio"]), save_predictions=path, batch_size=4)
        assert open(path).read() == str(g["tsv"])
        np.testing.assert_array_equal(df.onset.to_numpy(dtype=np.float64), g["onset"])
        np.testing.assert_array_equal(df.offset.to_numpy(dtype=np.float64), g["offset"])
        assert df.event_label.tolist() == [str(v) for v in g["event_label"]]


def test_batched_crnn_inference_equals_clip_by_clip():
    """The reference evaluates one clip per forward (evaluation_measures.py:204-207); eval-mode batches of any size
    must give the same posteriors (bit-exact here: per-clip arithmetic does not depend on the batch) and the same
    events as the oracle's post-processing of those posteriors."""
    from dcase2019_task4_amd.inference import get_predictions
    model, _ = gu.make_model(0)
    gu.set_bn(model, {k: v for k, v in __import__("oracle.ref_cpu", fromlist=["x"]).new_bn_state().items()})
    model.eval()
    N, T = 7, 628
    x = synth.make_input(5, N, T)
    with torch.no_grad():
        s_all = model(x.cuda())[0].cpu().numpy()
        s_one = np.concatenate([model(x[i:i + 1].cuda())[0].cpu().numpy() for i in range(N)])
    np.testing.assert_array_equal(s_all, s_one)
    labels = [f"c{i}" for i in range(10)]
    df = get_predictions(model, _DS(x), _Enc(labels).decode_strong, 8, batch_size=3)
    rows = pp.predictions(s_all, [f"clip_{i}.wav" for i in range(N)], labels, 8, 44100, 511)
    assert len(df) == len(rows)
    assert df.event_label.tolist() == [r[0] for r in rows]
    np.testing.assert_array_equal(df.onset.to_numpy(dtype=np.float64), np.array([r[1] for r in rows]))
    np.testing.assert_array_equal(df.offset.to_numpy(dtype=np.float64), np.array([r[2] for r in rows]))
