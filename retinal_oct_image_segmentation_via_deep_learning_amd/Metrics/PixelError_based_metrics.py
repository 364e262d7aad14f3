"""Drop-in for Metrics/PixelError_based_metrics.py (reference :3-37): MSE / RMSE of two masks.
The reference converts both masks to float64 and averages the squared difference (two temporaries and
two passes); here one streaming kernel accumulates sum (t-p)^2 in fp64 on the MI355X."""
import numpy as np

from ._counts import sqdiff_sum


def mean_squared_error(y_true, y_pred):
    """MSE = 1/n * sum (y_true - y_pred)^2, evaluated in float64 (reference :3-19)"""
    s, n = sqdiff_sum(y_true, y_pred)
    return np.float64(s / n) if n else np.float64("nan")


def root_mean_squared_error(y_true, y_pred):
    """RMSE = sqrt(MSE) (reference :21-37)"""
    return np.sqrt(mean_squared_error(y_true, y_pred))
