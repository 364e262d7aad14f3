"""Drop-in for Metrics/Biomarker_based_metrics.py (reference :3-38): thickness difference and
vascularity index.  Column sums / totals come from device reductions with numpy's dtype semantics
(the reference sums unsigned masks in uint64, so its thickness difference wraps around whenever the
prediction is thicker than the truth; that is reproduced, not "fixed")."""
import numpy as np

from ._counts import column_absdiff_mean, confusion_sums


def thickness_difference(y_true, y_pred):
    """mean_j | sum_i y_true[i, j...] - sum_i y_pred[i, j...] |  (reference :3-21)"""
    v, npdt = column_absdiff_mean(y_true, y_pred)
    return np.float32(v) if npdt == np.float32 else np.float64(v)


def vascularity_index(y_true, y_pred):
    """| sum(y_true)/size - sum(y_pred)/size |  (reference :23-38)"""
    (_, t, p, _, _, _), n, f32 = confusion_sums(y_true, y_pred)
    if n == 0:
        return np.float64("nan")
    v = abs(t / n - p / n)
    return np.float32(v) if f32 else np.float64(v)
