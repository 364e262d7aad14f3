"""One-pass confusion sums on the GPU (oct_confusion_counts) + host-side input normalisation.

Every metric of Metrics/{Region,ConfusionMatrix}_based_metrics is a formula over the same six sums, so
evaluating dice, iou, precision, recall ... on ONE pair of masks needs one kernel pass, not one per metric:
  * `Metrics.evaluate(y_true, y_pred)` returns all of them from a single pass (any input kind);
  * the individual functions share a one-entry cache when both masks are device tensors: the counts of the last
    pair are reused while the SAME tensor objects are passed again unmodified (identity via weak references plus
    torch's in-place version counters -- never by address, which a new tensor may inherit from a freed one).
"""
from __future__ import annotations

import weakref

import numpy as np
import torch

from .. import _lib as L

_ELEM = {np.dtype(np.uint8): 0, np.dtype(np.int32): 1, np.dtype(np.int64): 2, np.dtype(np.float32): 3,
         np.dtype(np.float64): 4, np.dtype(np.int8): 5, np.dtype(np.int16): 6, np.dtype(np.uint16): 7}
_TORCH2NP = {torch.bool: np.bool_, torch.uint8: np.uint8, torch.int8: np.int8, torch.int16: np.int16,
             torch.int32: np.int32, torch.int64: np.int64, torch.float32: np.float32, torch.float64: np.float64,
             torch.float16: np.float16, torch.bfloat16: np.float32}


def _canonical(dt: np.dtype) -> np.dtype:
    """dtype the kernel reduces in: numpy's promotion result mapped onto a supported element type."""
    dt = np.dtype(dt)
    if dt == np.bool_:
        return np.dtype(np.uint8)  # bool*bool = and, 1-bool in {0,1}: identical to u8 arithmetic on {0,1}
    if dt in _ELEM:
        return dt
    if dt.kind == "f":
        return np.dtype(np.float32) if dt.itemsize < 4 else np.dtype(np.float64)
    if dt.kind in "iu":
        return np.dtype(np.int64)
    raise TypeError(f"unsupported mask dtype {dt}")


def _prepare(y_true, y_pred, device=None):
    """-> (yt, yp) contiguous device tensors of the kernel dtype, numpy result dtype, kernel dtype"""
    if isinstance(y_true, torch.Tensor) or isinstance(y_pred, torch.Tensor):
        yt = y_true if isinstance(y_true, torch.Tensor) else torch.as_tensor(np.asarray(y_true))
        yp = y_pred if isinstance(y_pred, torch.Tensor) else torch.as_tensor(np.asarray(y_pred))
        dev = device or (yt.device if yt.is_cuda else yp.device if yp.is_cuda else torch.device("cuda"))
        npdt = np.result_type(_TORCH2NP[yt.dtype], _TORCH2NP[yp.dtype])
        if yt.shape != yp.shape:
            yt, yp = torch.broadcast_tensors(yt, yp)
    else:
        a, b = np.asarray(y_true), np.asarray(y_pred)
        npdt = np.result_type(a.dtype, b.dtype)
        if a.shape != b.shape:
            a, b = np.broadcast_arrays(a, b)
        dev = device or torch.device("cuda")
        yt, yp = torch.from_numpy(np.ascontiguousarray(a)), torch.from_numpy(np.ascontiguousarray(b))
    kdt = _canonical(npdt)
    tdt = getattr(torch, kdt.name)
    yt = yt.to(device=dev, dtype=tdt).contiguous()
    yp = yp.to(device=dev, dtype=tdt).contiguous()
    if yt.device.type != "cuda":
        raise L.OctError("Metrics need a GPU: there is no CPU fallback on the product path")
    # the kernels read 16-B vectors: a contiguous VIEW at an odd storage offset (masks[i] of a uint8 batch of 101x101
    # images, flat[1:]) passes .contiguous() unchanged -- the reference's numpy metric computes there, so copy
    if yt.data_ptr() & 15:
        yt = yt.clone()
    if yp.data_ptr() & 15:
        yp = yp.clone()
    return yt, yp, np.dtype(npdt), kdt


def sqdiff_sum(y_true, y_pred):
    """(sum (t-p)^2 in fp64, n) -- PixelError_based_metrics"""
    yt, yp, _, kdt = _prepare(y_true, y_pred)
    out = torch.empty(1, dtype=torch.float64, device=yt.device)
    L.check(L.lib().oct_sqdiff_sum(yt.data_ptr(), yp.data_ptr(), _ELEM[kdt], yt.numel(), out.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream), "oct_sqdiff_sum")
    return float(out.item()), yt.numel()


def column_absdiff_mean(y_true, y_pred):
    """mean over columns of |sum_axis0 t - sum_axis0 p| with numpy's dtype semantics -- Biomarker thickness"""
    yt, yp, npdt, kdt = _prepare(y_true, y_pred)
    if yt.dim() == 0:
        yt, yp = yt.reshape(1), yp.reshape(1)
    rows = yt.shape[0]
    cols = yt.numel() // max(rows, 1) if rows else 0
    out = torch.empty(1, dtype=torch.float64, device=yt.device)
    wrap = 1 if npdt.kind == "u" else 0   # numpy sums unsigned arrays in uint64: the difference wraps
    L.check(L.lib().oct_column_absdiff_sum(yt.data_ptr(), yp.data_ptr(), _ELEM[kdt], wrap, rows, cols, out.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream), "oct_column_absdiff_sum")
    if yt.dim() == 1:   # axis-0 sum of a 1-D mask is a scalar: one "column"
        return float(out.item()), npdt
    return (float(out.item()) / cols if cols else float("nan")), npdt


_last = {"key": None, "val": None}   # one entry: the eval loop pattern is "several metrics on the same pair"
launch_count = [0]                     # confusion-kernel passes issued (tests assert the one-pass property)


def _cache_key(y_true, y_pred, device=None):
    """Identity + torch's version counters + data pointers + the requested device.  Writes that bypass the version counter
    (a HIP-graph replay into a static buffer, a raw-pointer kernel of this library, `.data` / DLPack writes) are NOT seen:
    call `invalidate()` after such a write."""
    if isinstance(y_true, torch.Tensor) and isinstance(y_pred, torch.Tensor) and y_true.is_cuda and y_pred.is_cuda:
        return (weakref.ref(y_true), y_true._version, weakref.ref(y_pred), y_pred._version, y_true.data_ptr(), y_pred.data_ptr(),
                str(device) if device is not None else None, L.param_generation[0])
    return None


def invalidate():
    """Forget the cached counts (out-of-band writes into the cached tensors)."""
    _last["key"], _last["val"] = None, None


def _cache_hit(key):
    k = _last["key"]
    return (k is not None and key is not None and k[0]() is not None and k[0]() is key[0]() and k[2]() is key[2]()
            and k[1] == key[1] and k[3] == key[3] and k[4:] == key[4:])


def confusion_sums(y_true, y_pred, device=None):
    """Returns (sums, n, float32_result): sums = [tp, t, p, tn, fp, fn] as python ints (integer
    masks, exact) or floats (float masks, fp64 accumulation)."""
    key = _cache_key(y_true, y_pred, device)
    if _cache_hit(key):
        return _last["val"]
    yt, yp, npdt, kdt = _prepare(y_true, y_pred, device)
    n = yt.numel()
    out_i = torch.empty(6, dtype=torch.int64, device=yt.device)
    out_f = torch.empty(6, dtype=torch.float64, device=yt.device)
    L.check(L.lib().oct_confusion_counts(yt.data_ptr(), yp.data_ptr(), _ELEM[kdt], n, out_i.data_ptr(),
                                         out_f.data_ptr(), torch.cuda.current_stream().cuda_stream),
            "oct_confusion_counts")
    launch_count[0] += 1
    if kdt.kind == "f":
        sums = [float(v) for v in out_f.tolist()]
    else:
        sums = [int(v) for v in out_i.tolist()]
        if kdt.kind == "u":  # numpy sums unsigned arrays in uint64
            sums = [v & 0xFFFFFFFFFFFFFFFF for v in sums]
    val = (sums, n, np.dtype(npdt) == np.float32 or np.dtype(npdt) == np.float16)
    _last["key"], _last["val"] = key, val
    return val


def class_confusion_sums(y_true, y_pred, classes: int):
    """[classes][6] python ints: the six sums of the one-vs-rest masks (y_true == c), (y_pred == c) for every
    class, from ONE pass over the two class maps (oct_class_confusion_counts)."""
    yt, yp, npdt, kdt = _prepare(y_true, y_pred)
    if kdt.kind == "f":
        raise TypeError("class maps must be integer arrays")
    out = torch.empty((classes, 6), dtype=torch.int64, device=yt.device)
    scratch = torch.empty(48, dtype=torch.int64, device=yt.device)
    L.check(L.lib().oct_class_confusion_counts(yt.data_ptr(), yp.data_ptr(), _ELEM[kdt], yt.numel(), classes,
                                               out.data_ptr(), scratch.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream), "oct_class_confusion_counts")
    launch_count[0] += 1
    return [[int(v) for v in row] for row in out.tolist()], yt.numel()
