"""TEST INFRASTRUCTURE ONLY - numpy statement of the reference's strong-posterior post-processing.

Follows baseline/evaluation_measures.py:203-231 (get_predictions) and baseline/utils/utils.py:146-162
(ManyHotEncoder.decode_strong).  Two of its steps live in an un-vendored third-party package that is absent from
this image (dcase_util, unpinned in environment.yml): ``ProbabilityEncoder.binarization`` and
``DecisionEncoder.find_contiguous_regions``.  They are restated here from dcase_util's published source -
**parity unpinned** for exactly these two functions; everything around them (scipy's median filter, the DataFrame
assembly, the frame -> second conversion, the TSV format) is pinned by running the REAL get_predictions with these
two restatements bound into the stub module (oracle/gen_golden.py, fixture g9_predictions.npz).
"""
import numpy as np
import scipy.ndimage


class ProbabilityEncoder:
    """dcase_util.data.ProbabilityEncoder, the one method the reference calls."""

    def binarization(self, probabilities, binarization_type="global_threshold", threshold=0.5, time_axis=1):
        if binarization_type != "global_threshold":
            raise NotImplementedError(binarization_type)
        return np.array(probabilities > threshold, dtype=int)


class DecisionEncoder:
    """dcase_util.data.DecisionEncoder, the one method the reference calls."""

    def find_contiguous_regions(self, activity_array):
        activity_array = np.asarray(activity_array)
        change_indices = np.logical_xor(activity_array[1:], activity_array[:-1]).nonzero()[0]
        change_indices += 1
        if activity_array[0]:
            change_indices = np.r_[0, change_indices]
        if activity_array[-1]:
            change_indices = np.r_[change_indices, activity_array.size]
        return change_indices.reshape((-1, 2))


def filter_decisions(pred_strong, threshold=0.5, median_window=5):
    """[T, nclass] posteriors -> filtered 0/1 decisions (evaluation_measures.py:212-214)."""
    b = ProbabilityEncoder().binarization(pred_strong, binarization_type="global_threshold", threshold=threshold)
    return scipy.ndimage.median_filter(b, (median_window, 1))


def decode_strong(decisions, labels):
    """utils.py:146-162: [[label, onset, offset], ...] class by class, regions in time order."""
    out = []
    for i, col in enumerate(decisions.T):
        for row in DecisionEncoder().find_contiguous_regions(col):
            out.append([labels[i], int(row[0]), int(row[1])])
    return out


def predictions(strong_batch, filenames, labels, pooling_time_ratio, sample_rate, hop_length, threshold=0.5, median_window=5):
    """Rows (event_label, onset_s, offset_s, filename) in the order get_predictions emits them."""
    rows = []
    for pred, fn in zip(strong_batch, filenames):
        for lab, on, off in decode_strong(filter_decisions(pred, threshold, median_window), labels):
            rows.append((lab, on * pooling_time_ratio / (sample_rate / hop_length),
                         off * pooling_time_ratio / (sample_rate / hop_length), fn))
    return rows
