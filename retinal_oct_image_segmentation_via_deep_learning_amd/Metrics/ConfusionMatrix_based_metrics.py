"""Drop-in for Metrics/ConfusionMatrix_based_metrics.py (reference :4-84): accuracy, sensitivity,
precision, specificity from ONE confusion-count kernel; `auc_score` stays a host passthrough
(sort-based sklearn call, outside the accelerated path -- SURVEY.md §2 row 18)."""
import numpy as np

from ._counts import confusion_sums


def _res(v, f32):
    return np.float32(v) if f32 else np.float64(v)


def accuracy(y_true, y_pred):
    """(TP + TN) / N, no epsilon (reference :4-18)"""
    (tp, _, _, tn, _, _), n, f32 = confusion_sums(y_true, y_pred)
    if n == 0:
        return _res(float("nan"), f32)
    return _res((tp + tn) / n, f32)


def sensitivity(y_true, y_pred):
    """TP / (TP + FN + 1e-7) (reference :20-33)"""
    (tp, _, _, _, _, fn), _, f32 = confusion_sums(y_true, y_pred)
    return _res(tp / (tp + fn + 1e-7), f32)


def precision(y_true, y_pred):
    """TP / (TP + FP + 1e-7) (reference :35-48)"""
    (tp, _, _, _, fp, _), _, f32 = confusion_sums(y_true, y_pred)
    return _res(tp / (tp + fp + 1e-7), f32)


def specificity(y_true, y_pred):
    """TN / (TN + FP + 1e-7) (reference :50-63)"""
    (_, _, _, tn, fp, _), _, f32 = confusion_sums(y_true, y_pred)
    return _res(tn / (tn + fp + 1e-7), f32)


def auc_score(y_true, y_pred):
    """Area under the ROC curve via sklearn on the host, as in the reference (:65-84)."""
    from sklearn.metrics import roc_auc_score
    yt = np.asarray(y_true.cpu() if hasattr(y_true, "cpu") else y_true).flatten()
    yp = np.asarray(y_pred.cpu() if hasattr(y_pred, "cpu") else y_pred).flatten()
    try:
        return roc_auc_score(yt, yp)
    except ValueError:
        return 0.0
