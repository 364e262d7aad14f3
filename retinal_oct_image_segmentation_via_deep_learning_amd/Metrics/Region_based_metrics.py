"""Drop-in for Metrics/Region_based_metrics.py (reference :3-61): same names, same positional
(y_true, y_pred) signature, same formulas and 1e-7 epsilons -- the three passes plus temporary of
each numpy formula become one fused confusion-count kernel on the MI355X."""
import numpy as np

from ._counts import confusion_sums


def _res(v, f32):
    return np.float32(v) if f32 else np.float64(v)


def dice_coefficient(y_true, y_pred):
    """DSC = 2|X.Y| / (|X| + |Y| + 1e-7)   (reference :3-16; note its "union" is |X|+|Y|)"""
    (tp, t, p, _, _, _), _, f32 = confusion_sums(y_true, y_pred)
    return _res((2.0 * tp) / (t + p + 1e-7), f32)


def iou_score(y_true, y_pred):
    """IoU = |X.Y| / (|X| + |Y| - |X.Y| + 1e-7)   (reference :18-31)"""
    (tp, t, p, _, _, _), _, f32 = confusion_sums(y_true, y_pred)
    return _res(tp / (t + p - tp + 1e-7), f32)


def precision(y_true, y_pred):
    """TP / (sum(y_pred) + 1e-7)   (reference :33-46)"""
    (tp, _, p, _, _, _), _, f32 = confusion_sums(y_true, y_pred)
    return _res(tp / (p + 1e-7), f32)


def recall(y_true, y_pred):
    """TP / (sum(y_true) + 1e-7)   (reference :48-61)"""
    (tp, t, _, _, _, _), _, f32 = confusion_sums(y_true, y_pred)
    return _res(tp / (t + 1e-7), f32)
