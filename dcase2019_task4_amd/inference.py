"""Batched on-GPU inference + post-processing: the reference's ``get_predictions``
(baseline/evaluation_measures.py:203-231) with the same signature and the same DataFrame / TSV output.

The reference runs ONE clip per forward and post-processes it on the host (threshold, scipy median filter,
run-length decode) - 1 168 + ~400 forwards per epoch.  Here clips go through the eval-mode CRNN ``batch_size`` at a
time and ``sed_postprocess`` thresholds, median-filters and run-length-decodes the whole batch on the device; the host
only assembles the event table.  There is no CPU fallback: the model and its inputs live on the GPU.
"""
import ctypes as C

import numpy as np
import pandas as pd
import torch

from . import _lib


class _Cfg:
    """The three config.py values get_predictions reads (config.py:17,19,39)."""
    sample_rate = 44100
    hop_length = 511
    median_window = 5


def postprocess(strong, threshold=0.5, median_window=5, want_binary=False):
    """strong [N, T, nclass] cuda float32 -> (ev_count [N, nclass] int32, ev_pairs [N, nclass, max_ev, 2] int32[, binary])."""
    if strong.device.type != "cuda":
        raise _lib.SedError("postprocess needs a GPU tensor (no CPU fallback)")
    strong = strong.contiguous().float()
    N, T, NC = strong.shape
    max_ev = (T + 1) // 2
    cnt = torch.empty(N, NC, dtype=torch.int32, device=strong.device)
    pairs = torch.zeros(N, NC, max_ev, 2, dtype=torch.int32, device=strong.device)
    binary = torch.empty(N, T, NC, dtype=torch.uint8, device=strong.device) if want_binary else None
    _lib.check(_lib.lib().sed_postprocess(_lib.ptr(strong), N, T, NC, float(threshold), int(median_window),
                                          _lib.ptr(binary) if want_binary else None, _lib.ptr(cnt), _lib.ptr(pairs), max_ev,
                                          _lib.stream_ptr()), "sed_postprocess")
    return (cnt, pairs, binary) if want_binary else (cnt, pairs)


def get_predictions(model, valid_dataset, decoder, pooling_time_ratio=1, save_predictions=None, batch_size=64, cfg=None,
                    threshold=0.5):
    """Drop-in for evaluation_measures.get_predictions.

    ``valid_dataset[i]`` yields ``(input [1, T, F], label)`` and has ``.filenames`` (DataLoadDf, DataLoad.py:47-72);
    ``decoder`` is ``many_hot_encoder.decode_strong`` (main.py:326): when it is a bound method of an object with
    ``.labels`` the run-length decode happens on the device, otherwise the callable gets the device-filtered 0/1 matrix
    of each clip, exactly as in the reference.  ``cfg`` supplies sample_rate / hop_length / median_window
    (default: the values of baseline/config.py)."""
    cfg = cfg or _Cfg
    labels = getattr(getattr(decoder, "__self__", None), "labels", None)
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise _lib.SedError("get_predictions needs the model on the GPU (no CPU fallback)")
    was_training = model.training
    model.eval()
    frames, cols, stage = [], [], None
    n = len(valid_dataset)
    filenames = valid_dataset.filenames
    try:
        with torch.no_grad():
            for i0 in range(0, n, batch_size):
                idx = range(i0, min(n, i0 + batch_size))
                items = [torch.as_tensor(valid_dataset[i][0]) for i in idx]
                if stage is None or stage.shape[1:] != items[0].shape or stage.dtype != items[0].dtype:
                    # one reusable pinned staging buffer: a fresh pageable torch.stack per batch cost more than the forward
                    stage = torch.empty((batch_size,) + tuple(items[0].shape), dtype=items[0].dtype).pin_memory()
                for k, it in enumerate(items):
                    stage[k].copy_(it)
                x = stage[:len(items)].to(dev, non_blocking=True).float()
                strong, _ = model(x)
                if labels is not None:
                    cnt, pairs = postprocess(strong, threshold, cfg.median_window)
                    # compact on the device: rows ordered (clip, class, event) = the order of the reference's nested loops
                    keep = torch.arange(pairs.shape[2], device=dev)[None, None, :] < cnt[:, :, None]
                    k_idx, c_idx, _ = torch.nonzero(keep, as_tuple=True)
                    ev = pairs[keep]                                        # [n_events, 2]
                    cols.append((k_idx.cpu().numpy() + i0, c_idx.cpu().numpy(), ev.cpu().numpy()))
                else:
                    _, _, binary = postprocess(strong, threshold, cfg.median_window, want_binary=True)
                    binary = binary.cpu().numpy().astype(int)
                    for k, i in enumerate(idx):
                        pred = pd.DataFrame(decoder(binary[k]), columns=["event_label", "onset", "offset"])
                        pred["filename"] = filenames.iloc[i]
                        frames.append(pred)
    finally:
        model.train(was_training)
    if labels is not None:
        # ONE DataFrame for the whole set, equal to the reference's concatenation of per-clip frames (including its
        # per-clip 0..k-1 index); building 1 168 small DataFrames on the host cost more than all the device work
        clip = np.concatenate([c[0] for c in cols]) if cols else np.zeros(0, dtype=np.int64)
        cls = np.concatenate([c[1] for c in cols]) if cols else np.zeros(0, dtype=np.int64)
        ev = np.concatenate([c[2] for c in cols]) if cols else np.zeros((0, 2), dtype=np.int32)
        starts = np.r_[0, np.flatnonzero(np.diff(clip)) + 1] if len(clip) else np.zeros(0, dtype=np.int64)
        index = np.arange(len(clip)) - np.repeat(starts, np.diff(np.r_[starts, len(clip)])) if len(clip) else clip
        prediction_df = pd.DataFrame({"event_label": np.asarray(labels, dtype=object)[cls],
                                      "onset": ev[:, 0].astype(np.int64), "offset": ev[:, 1].astype(np.int64),
                                      "filename": np.asarray(filenames)[clip]}, index=index,
                                     columns=["event_label", "onset", "offset", "filename"])
    else:
        prediction_df = pd.concat(frames) if frames else pd.DataFrame(columns=["event_label", "onset", "offset", "filename"])
    # In seconds (evaluation_measures.py:225-227)
    prediction_df.onset = prediction_df.onset * pooling_time_ratio / (cfg.sample_rate / cfg.hop_length)
    prediction_df.offset = prediction_df.offset * pooling_time_ratio / (cfg.sample_rate / cfg.hop_length)
    if save_predictions is not None:
        prediction_df.to_csv(save_predictions, index=False, sep="\t")
    return prediction_df
