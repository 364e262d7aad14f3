"""Mirror of the reference's Metrics/ tree: device-side reductions behind the same signatures, plus a
one-pass evaluation entry point the reference lacks.

`evaluate(y_true, y_pred)` -- every Region / ConfusionMatrix metric of a pair of masks from ONE pass over them
(the reference: 3-4 numpy passes and a temporary PER METRIC, Region_based_metrics.py:3-61,
ConfusionMatrix_based_metrics.py:4-63).  `evaluate(y_true, y_pred, classes=C)` takes two integer CLASS MAPS
(e.g. `model.predict(x)` and the label map) and returns, per metric, an array over the C classes: class c is
scored one-vs-rest, exactly what calling the reference's function on `(y_true == c)`, `(y_pred == c)` returns.
"""
import numpy as np

from . import _counts

METRIC_NAMES = ("dice_coefficient", "iou_score", "precision", "recall", "accuracy", "sensitivity", "specificity")


def _formulas(tp, t, p, tn, fp, fn, n):
    """the reference's formulas, verbatim epsilons (Region :14-15,28-30,44-45,59-60; ConfusionMatrix :16-17,32,47,62)"""
    return {
        "dice_coefficient": (2.0 * tp) / (t + p + 1e-7),
        "iou_score": tp / (t + p - tp + 1e-7),
        "precision": tp / (p + 1e-7),
        "recall": tp / (t + 1e-7),
        "accuracy": (tp + tn) / n if n else float("nan"),
        "sensitivity": tp / (tp + fn + 1e-7),
        "cm_precision": tp / (tp + fp + 1e-7),
        "specificity": tn / (tn + fp + 1e-7),
    }


def evaluate(y_true, y_pred, classes=None):
    """dict metric name -> value (binary masks) or -> float64 array [classes] (class maps); one kernel pass.
    Keys: dice_coefficient, iou_score, precision, recall (Region_based_metrics), accuracy, sensitivity,
    cm_precision, specificity (ConfusionMatrix_based_metrics; its `precision` is TP/(TP+FP)), and "counts"."""
    if classes is None:
        (tp, t, p, tn, fp, fn), n, f32 = _counts.confusion_sums(y_true, y_pred)
        res = {k: (np.float32(v) if f32 else np.float64(v)) for k, v in _formulas(tp, t, p, tn, fp, fn, n).items()}
        res["counts"] = {"tp": tp, "t": t, "p": p, "tn": tn, "fp": fp, "fn": fn, "n": n}
        return res
    rows, n = _counts.class_confusion_sums(y_true, y_pred, int(classes))
    per = [_formulas(*r, n) for r in rows]
    res = {k: np.array([d[k] for d in per], dtype=np.float64) for k in per[0]}
    res["counts"] = np.array(rows, dtype=np.int64)
    return res
