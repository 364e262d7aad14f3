"""ctypes binding of liboct_hip.so (C ABI declared in include/oct_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails this
module raises.  Build it with `python __graft_entry__.py` (or `make -C .../csrc`).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OCT_HIP_LIB") or os.path.join(PKG_DIR, "liboct_hip.so")  # override: ablation builds
CSRC_DIR = os.path.join(PKG_DIR, "csrc")

DT_BF16, DT_F32 = 0, 1
XF_NONE, XF_AFFINE_RELU, XF_AFFINE = 0, 1, 2
IN_PLAIN, IN_S2D = 0, 1
IMG_SHIFT_ALL = 2   # OctWgradDesc.in_img_shift: all depth taps in one launch (OCT_IMG_SHIFT_ALL)
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
ACT_PRELU = 3   # host-side tag only: PReLU has its own entry points (oct_affine_prelu_fwd / _bwd)
OUT_PLAIN, OUT_D2S = 0, 1
PACK_CONV_FPROP, PACK_CONV_DGRAD, PACK_DECONV_FPROP, PACK_DECONV_DGRAD, PACK_1X1_DGRAD, PACK_1X1_FPROP = range(6)
PACK_CONV3D_FPROP, PACK_CONV3D_DGRAD, PACK_DECONV3D_FPROP, PACK_DECONV3D_DGRAD = 6, 7, 8, 9
MAX_CLASSES = 16
HEAD_LOSS_SLOTS = 2 + 3 * MAX_CLASSES

c_void_p, c_int, c_size_t, c_float, c_double = C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_double


class PackJob(C.Structure):          # struct OctPackJob
    _fields_ = [("mode", c_int), ("cout", c_int), ("cin", c_int), ("reserved", c_int), ("w", c_void_p),
                ("wpacked", c_void_p)]


class UnpackJob(C.Structure):        # struct OctUnpackJob
    _fields_ = [("mode", c_int), ("cout", c_int), ("cin", c_int), ("accumulate", c_int), ("nparts", c_int),
                ("reserved", c_int), ("dwp", c_void_p), ("grad", c_void_p)]


PACK_BATCH_MAX = 96


class ConvDesc(C.Structure):
    _fields_ = [(k, c_int) for k in (
        "dtype", "n", "h", "w", "c0", "c1", "cout", "taps", "xform0", "xform1", "in_mode", "out_mode",
        "split", "want_stats", "kh", "kw", "depth", "out_img_mul", "out_img_add")]


class ConvArgs(C.Structure):
    _fields_ = [(k, c_void_p) for k in (
        "x0", "x1", "scale0", "shift0", "scale1", "shift1", "wpacked", "bias", "y0", "y1", "stat_partials")]


class WgradDesc(C.Structure):
    _fields_ = [(k, c_int) for k in (
        "dtype", "n", "h", "w", "c0", "c1", "cout", "taps", "xform0", "xform1", "dy_mode", "kh", "kw", "depth",
        "in_img_shift", "dy_img_mul", "dy_img_add", "partials")]


class WgradArgs(C.Structure):
    _fields_ = [(k, c_void_p) for k in ("x0", "x1", "scale0", "shift0", "scale1", "shift1", "dy", "dwp", "dbias",
                                                "dy_y", "dy_coef", "dy_scale", "dy_shift", "dbias_partials")]


class HeadDesc(C.Structure):
    _fields_ = [(k, c_int) for k in ("dtype", "n", "h", "w", "feat", "classes")]


# name -> (restype, argtypes).  Every symbol of include/oct_hip.h is listed here;
# tests/test_abi.py checks the two stay in sync.
SIGNATURES = {
    "oct_version_string": (C.c_char_p, []),
    "oct_version": (c_int, []),
    "oct_get_last_error": (c_int, [C.c_char_p, c_size_t]),
    "oct_device_count": (c_int, []),
    "oct_conv_stat_blocks": (c_int, [C.POINTER(ConvDesc)]),
    "oct_packed_weight_elems": (c_size_t, [c_int, c_int, c_int]),
    "oct_pack_weights": (c_int, [c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "oct_pack_weights3d": (c_int, [c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "oct_unpack_wgrad3d": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_depth_pool_fwd": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p]),
    "oct_depth_pool_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p]),
    "oct_pack_weights_kk": (c_int, [c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_unpack_wgrad_kk": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_affine_prelu_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "oct_affine_prelu_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                     c_int, c_void_p]),
    "oct_maxpool_idx_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_index_scatter": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_size_t, c_int, c_void_p]),
    "oct_index_gather": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_size_t, c_int, c_void_p]),
    "oct_prelu_bn_fused_ok": (c_int, [c_int, c_int]),
    "oct_dact_bn_reduce_prelu": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int, c_int, c_int, c_int, c_void_p]),
    "oct_bn_bwd_apply_prelu_to": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int,
                                          c_void_p]),
    "oct_maxpool_code_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_window_scatter": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_window_gather": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_pack_weights_batch": (c_int, [c_int, c_int, C.POINTER(PackJob), c_void_p]),
    "oct_conv_forward": (c_int, [C.POINTER(ConvDesc), C.POINTER(ConvArgs), c_void_p]),
    "oct_conv_wgrad": (c_int, [C.POINTER(WgradDesc), C.POINTER(WgradArgs), c_void_p]),
    "oct_conv_wgrad_fused_apply_ok": (c_int, [C.POINTER(WgradDesc)]),
    "oct_conv_wgrad_all_depth_taps_ok": (c_int, [C.POINTER(WgradDesc)]),
    "oct_conv_wgrad_partials": (c_int, [C.POINTER(WgradDesc)]),
    "oct_reduce_bias_partials": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "oct_unpack_wgrad": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "oct_unpack_wgrad_batch": (c_int, [c_int, C.POINTER(UnpackJob), c_void_p]),
    "oct_bn_finalize": (c_int, [c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_float, c_float,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "oct_bn_eval_coeffs": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                   c_void_p, c_void_p]),
    "oct_bn_relu_pool_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                     c_void_p]),
    "oct_bn_relu_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "oct_dact_bn_reduce": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_dact_bn_reduce_blocks": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "oct_bn_bwd_apply_pool_ok": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "oct_bn_bwd_apply_pool": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int, c_int, c_int, c_int, c_void_p]),
    "oct_bn_bwd_finalize": (c_int, [c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_int, c_void_p]),
    "oct_bn_bwd_apply": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int,
                                 c_void_p]),
    "oct_bn_bwd_apply_to": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int,
                                    c_void_p]),
    "oct_channel_sum": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "oct_rowdot_ok": (c_int, [c_int, c_int]),
    "oct_rowdot_blocks": (c_int, [c_size_t, c_int]),
    "oct_rowdot_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "oct_rowdot_bwd_data": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "oct_rowdot_bwd_weight": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_void_p]),
    "oct_rowdot_bwd_weight_bias": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int,
                                           c_void_p]),
    "oct_head_blocks": (c_int, [C.POINTER(HeadDesc)]),
    "oct_head_forward": (c_int, [C.POINTER(HeadDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "oct_head_loss_finalize": (c_int, [C.POINTER(HeadDesc), c_void_p, c_int, c_float, c_float, c_float, c_void_p,
                                       c_void_p, c_void_p]),
    "oct_head_dlogits": (c_int, [C.POINTER(HeadDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "oct_head_backward_fused": (c_int, [C.POINTER(HeadDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p]),
    "oct_nchw_to_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_nhwc_to_nchw": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_sgd_step": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float, c_float, c_int,
                             c_void_p]),
    "oct_affine_act_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_int,
                                   c_void_p]),
    "oct_affine_res_act_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t,
                                       c_int, c_void_p]),
    "oct_act_bwd": (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "oct_maxpool_fwd": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_maxpool_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_bilinear_up_fwd": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_bilinear_up_bwd": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_bilinear_resize_fwd": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_bilinear_resize_bwd": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_depth_to_space": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_space_to_depth": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oct_gate_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "oct_gate_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "oct_confusion_counts": (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_void_p, c_void_p, c_void_p]),
    "oct_class_confusion_counts": (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_int, c_void_p, c_void_p, c_void_p]),
    "oct_sqdiff_sum": (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_void_p, c_void_p]),
    "oct_column_absdiff_sum": (c_int, [c_void_p, c_void_p, c_int, c_int, c_size_t, c_size_t, c_void_p, c_void_p]),
}

_lib = None
_lock = threading.Lock()
# bumped whenever parameter memory is modified behind torch's back (FusedSGD, DDP broadcast)
param_generation = [0]


class OctError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile liboct_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, "-j8"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise OctError("building liboct_hip.so failed (see output above)")
    return LIB_PATH


def lib():
    """The loaded library.  Raises OctError when it has not been built -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise OctError(
                    f"{LIB_PATH} is missing: the HIP extension has not been built "
                    "(run `python __graft_entry__.py` or `make -C .../csrc`). There is no CPU fallback.")
            handle = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
                fn.restype = res
                fn.argtypes = args
            _lib = handle
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    lib().oct_get_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise OctError(f"{what or 'liboct_hip'} failed with code {rc}: {last_error()}")


def ptr(t) -> int | None:
    """data_ptr of a torch tensor (None stays NULL)."""
    return None if t is None else t.data_ptr()
