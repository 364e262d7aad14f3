"""Autograd ops over NHWC device tensors, each backed by the C ABI (include/oct_hip.h).

The YNet/BioNet U-Nets run through `engine.UNetEngine`'s fused schedule (raw conv outputs, BN+ReLU
applied on the consumer's load).  The reference's other block families -- MGUNet's UnetConv /
UnetUp / UnetUp4 (MGUNet_2021.py:42-108) and SD_Layer_Net's conv_block / up_conv /
Attention_block (common.py:6-91) -- mix residual sums, bilinear up-sampling, sigmoid gates and
convolutions without BatchNorm, so they are composed from the ops below with materialised
activations; torch.autograd only orders the calls and sums fan-out gradients.  Every tensor
between ops is (N, H, W, C) in the compute dtype (bf16, or fp32 in parity mode).
There is no CPU fallback: `_lib.lib()` raises when the HIP library is missing.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L
from .engine import BN_EPS, BN_MOMENTUM, BNState, Src, UNetEngine, _stream

_engines: dict = {}
# OCT_LAZY=0 (or ops.LAZY[0] = False): every activation between ops is materialised -- the A/B switch of the deferred
# BN + ReLU / bias schedule (LazyAct below); "relu": only the BN + ReLU deferral
DEBUG = None   # a list: ConvAffineAct.backward appends (tag, tensors) -- probes only
LAZY = [{"0": False, "relu": "relu"}.get(os.environ.get("OCT_LAZY", "1"), True)]


def kernels(dtype: str) -> UNetEngine:
    """Launch helpers (weight packing cache, conv / wgrad descriptors) for one compute dtype."""
    e = _engines.get(dtype)
    if e is None:
        e = _engines[dtype] = UNetEngine(1, 1, 4, dtype)
    return e


_PACK_SHAPE = {
    L.PACK_CONV_FPROP: lambda co, ci: (co, 9, ci), L.PACK_CONV_DGRAD: lambda co, ci: (ci, 9, co),
    L.PACK_DECONV_FPROP: lambda co, ci: (4 * co, 1, ci), L.PACK_DECONV_DGRAD: lambda co, ci: (ci, 1, 4 * co),
    L.PACK_1X1_DGRAD: lambda co, ci: (ci, 1, co), L.PACK_1X1_FPROP: lambda co, ci: (co, 1, ci)}


def packed(e: UNetEngine, w: torch.Tensor, mode: int, cout: int, cin: int, cache: bool = True, kk=None) -> torch.Tensor:
    """MFMA-fragment-ordered copy of a weight in the compute dtype.  The copy lives on the
    parameter object itself (so it dies with it and can never be mistaken for another tensor that
    later reuses the address) and is refreshed when torch's version counter or the raw-pointer
    optimizer's generation moves."""
    ver = (w._version, L.param_generation[0], w.data_ptr())
    slot = (mode, e.dtype)
    store = w.__dict__.setdefault("_oct_packed", {}) if cache else {}
    hit = store.get(slot)
    if hit is not None and hit[0] == ver:
        return hit[1]
    rows, taps, kch = _PACK_SHAPE[mode](cout, cin)
    if kk is not None:      # general kernel size (7x3): same modes, taps = kh*kw
        taps = kk[0] * kk[1]
    out = torch.empty(L.lib().oct_packed_weight_elems(rows, taps, kch), dtype=e.tdt, device=w.device)
    if kk is not None:
        L.check(L.lib().oct_pack_weights_kk(mode, e.dt, w.data_ptr(), out.data_ptr(), cout, cin, kk[0], kk[1], _stream()),
                "oct_pack_weights_kk")
    else:
        L.check(L.lib().oct_pack_weights(mode, e.dt, w.data_ptr(), out.data_ptr(), cout, cin, _stream()), "oct_pack_weights")
    store[slot] = (ver, out, cout, cin, kk)
    return out


def prepack(dtype: str, module: torch.nn.Module) -> None:
    """Refresh, in ONE launch, every packed copy an earlier step made of `module`'s weights and the optimizer has since
    outdated (fprop and dgrad order of ~45 convolutions = 87 five-microsecond launches per AttU_Net step otherwise).
    The per-op `packed()` calls of the step then all hit the cache."""
    NBT_PENDING.clear()   # = drop_pending_counters(): every network forward starts here
    e = kernels(dtype)
    jobs, news = [], []
    for w in module.parameters():
        store = w.__dict__.get("_oct_packed")
        if not store:
            continue
        ver = (w._version, L.param_generation[0], w.data_ptr())
        for slot, hit in store.items():
            if slot[1] != e.dtype or hit[0] == ver or hit[4] is not None or slot[0] > L.PACK_1X1_FPROP:
                continue
            jobs.append(L.PackJob(slot[0], hit[2], hit[3], 0, w.data_ptr(), hit[1].data_ptr()))
            news.append((store, slot, (ver,) + tuple(hit[1:])))
    if jobs:
        arr = (L.PackJob * len(jobs))(*jobs)
        L.check(L.lib().oct_pack_weights_batch(e.dt, len(jobs), arr, _stream()), "oct_pack_weights_batch")
        for store, slot, val in news:
            store[slot] = val


NBT_PENDING = []   # BatchNorm step counters of the ops run since the last flush

# --------------------------------------------------------------------------------------------------------------------
# Pixel-pair folding of 16-channel 3x3 convolutions (round 3; MGU-Net with feature_scale 4: 16-channel full-resolution levels).
# The pipelined kernels want channel counts in multiples of 32.  An NHWC tensor (n, h, w, 16k) IS the tensor (n, h, w/2, 32k) --
# the same bytes: two horizontally adjacent pixels' channels side by side -- and a 3x3 convolution of the one is a 3x3 convolution
# of the other with the folded filter
#     W'[(q, co), (p, c), ty, d + 1] = W[co, c, ty, tx]   where tx = 2d + p - q + 1 in {0, 1, 2}   (zero otherwise),
# q / p = parity of the output / input column, d in {-1, 0, 1} the folded column offset: output column 2x' + q reads input column
# 2x' + q + tx - 1 = 2(x' + d) + p.  Half of W' is zero -- twice the FLOPs, on layers that are HBM-bound: the generic kernels ran the
# eight 16-channel convolutions and four weight gradients of MGU-Net's first level in 7.6 of its 18.8 ms per step.  Only host code:
# views of the activations, the folded filter (a gather), the two parities of the BatchNorm partial sums added, the filter gradient
# un-folded (an index_add).  OCT_FOLD16=0 switches it off.
# --------------------------------------------------------------------------------------------------------------------
FOLD16 = [os.environ.get("OCT_FOLD16", "1") != "0"]
_FOLD_TX = {}


def _fold_tx(dev):
    t = _FOLD_TX.get(dev)
    if t is None:
        idx = [[[(2 * d + p - q + 1) if 0 <= 2 * d + p - q + 1 <= 2 else 3 for d in (-1, 0, 1)] for p in (0, 1)] for q in (0, 1)]
        t = _FOLD_TX[dev] = torch.tensor(idx, dtype=torch.long, device=dev)      # [q][p][d + 1] -> tx (3 = the zero column)
    return t


def fold16_ok(e, taps, kk, wd, c0, c1, cout) -> bool:
    return (FOLD16[0] and e.dt == L.DT_BF16 and taps in (1, 9) and kk is None and wd % 2 == 0 and wd >= 4 and c0 % 16 == 0 and c1 % 16 == 0
            and cout % 16 == 0 and bool(c0 % 32 or c1 % 32 or cout % 32))


def fold16_weight(w, c0, c1):
    """(cout, c0 + c1, 3, 3) -> (2 cout, 2 c0 + 2 c1, 3, 3); folded input channels: [source 0: (p, c)] [source 1: (p, c)]"""
    cout = w.shape[0]
    parts = []
    if w.shape[2] == 1:     # 1x1: the two column parities do not mix -- [[W, 0], [0, W]] per source
        wd_ = w.detach()
        for lo, c in ((0, c0), (c0, c1)):
            if c:
                t = torch.zeros((2, cout, 2, c, 1, 1), dtype=w.dtype, device=w.device)
                t[0, :, 0] = wd_[:, lo:lo + c]
                t[1, :, 1] = wd_[:, lo:lo + c]
                parts.append(t.reshape(2 * cout, 2 * c, 1, 1))
        return torch.cat(parts, dim=1).contiguous() if len(parts) > 1 else parts[0].contiguous()
    g = torch.nn.functional.pad(w.detach(), (0, 1))[:, :, :, _fold_tx(w.device)]     # (cout, cin, ty, q, p, d)
    for lo, c in ((0, c0), (c0, c1)):
        if c:
            parts.append(g[:, lo:lo + c].permute(3, 0, 4, 1, 2, 5).reshape(2 * cout, 2 * c, 3, 3))
    return torch.cat(parts, dim=1).contiguous() if len(parts) > 1 else parts[0].contiguous()


def unfold16_wgrad(dwf, cout, c0, c1):
    """gradient of the folded filter (2 cout, 2 c0 + 2 c1, 3, 3) -> gradient of the filter (cout, c0 + c1, 3, 3)"""
    parts, lo = [], 0
    if dwf.shape[2] == 1:
        for c in (c0, c1):
            if c:
                t = dwf[:, lo:lo + 2 * c].reshape(2, cout, 2, c, 1, 1)
                parts.append(t[0, :, 0] + t[1, :, 1])
                lo += 2 * c
        return torch.cat(parts, dim=1).contiguous() if len(parts) > 1 else parts[0].contiguous()
    tx = _fold_tx(dwf.device).reshape(-1)
    for c in (c0, c1):
        if c:
            g = dwf[:, lo:lo + 2 * c].reshape(2, cout, 2, c, 3, 3).permute(1, 3, 4, 0, 2, 5).reshape(cout, c, 3, 12)   # (co, c, ty, (q, p, d))
            parts.append(torch.zeros((cout, c, 3, 4), dtype=dwf.dtype, device=dwf.device).index_add_(3, tx, g)[..., :3])
            lo += 2 * c
    return torch.cat(parts, dim=1).contiguous() if len(parts) > 1 else parts[0].contiguous()


def fold16_deconv_weight(w):
    """ConvTranspose2d(k = 2, s = 2) weight (cin, cout, 2, 2) -> folded (2 cin, 2 cout, 2, 2): input pixel pair (pi, c), output channels
    (dx, co), kernel (dy, pi): out[2y + dy, 2x'' + pi, (dx, co)] = sum_c x[y, x'', (pi, c)] W[c, co, dy, dx] -- the two input parities do
    not mix, the horizontal kernel offset becomes the channel half"""
    cin, cout = w.shape[0], w.shape[1]
    t = torch.zeros((2, cin, 2, cout, 2, 2), dtype=w.dtype, device=w.device)     # (pi', c, dx, co, dy, pi)
    wp_ = w.detach().permute(0, 3, 1, 2)                                           # (c, dx, co, dy)
    t[0, :, :, :, :, 0] = wp_
    t[1, :, :, :, :, 1] = wp_
    return t.reshape(2 * cin, 2 * cout, 2, 2)


def unfold16_deconv_wgrad(dwf, cin, cout):
    t = dwf.reshape(2, cin, 2, cout, 2, 2)
    return (t[0, :, :, :, :, 0] + t[1, :, :, :, :, 1]).permute(0, 2, 3, 1).contiguous()   # (c, dx, co, dy) -> (c, co, dy, dx)


def fold16_deconv_ok(e, k, wd, cin, cout) -> bool:
    return (FOLD16[0] and e.dt == L.DT_BF16 and k == 2 and wd % 2 == 0 and cin % 16 == 0 and cout % 16 == 0 and bool(cin % 32 or cout % 32))


def _fold_src(x0, c0, xf0, x1, c1, xf1):
    n, h, wd, _ = x0.shape

    def bn2(xf):
        return BNState(xf[0].repeat(2), xf[1].repeat(2), relu=xf[2]) if xf else None
    return Src(x0.view(n, h, wd // 2, 2 * c0), 2 * c0, bn2(xf0), None if x1 is None else x1.view(n, h, wd // 2, 2 * c1), 2 * c1, bn2(xf1))


def flush_counters() -> None:
    """num_batches_tracked += 1 for every train-mode BatchNorm op since the last call: ONE multi-tensor launch at the end of
    a block's / network's forward (every public forward ends in to_nchw) instead of one 5-us launch per layer."""
    if NBT_PENDING:
        # a BatchNorm applied twice since the last flush appears twice: ONE entry per tensor with its count (the multi-tensor
        # kernel would otherwise read-modify-write the same address from two slots)
        uniq = {}
        for t in NBT_PENDING:
            uniq.setdefault(id(t), [t, 0])[1] += 1
        NBT_PENDING.clear()
        torch._foreach_add_([t for t, _ in uniq.values()], [c for _, c in uniq.values()])


def drop_pending_counters() -> None:
    """Start of a public forward: counters left behind by a forward that raised before its to_nchw must not be bumped by
    this, unrelated, one."""
    NBT_PENDING.clear()


def _need_cuda(t: torch.Tensor):
    if t.device.type != "cuda":
        raise L.OctError("the HIP path needs a device tensor (there is no CPU fallback)")


class ToNHWC(torch.autograd.Function):
    """(N,C,H,W) float -> (N,H,W,C) compute dtype."""

    @staticmethod
    def forward(ctx, x, dtype):
        _need_cuda(x)
        e = kernels(dtype)
        n, c, h, w = x.shape
        xf = x.detach().to(torch.float32).contiguous()
        out = e._act(n, h, w, c, x.device)
        L.check(L.lib().oct_nchw_to_nhwc(e.dt, xf.data_ptr(), out.data_ptr(), n, c, h, w, _stream()), "oct_nchw_to_nhwc")
        ctx.dtype = dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        return ToNCHW.apply(dout, ctx.dtype), None


class ToNCHW(torch.autograd.Function):
    """(N,H,W,C) compute dtype -> (N,C,H,W) fp32."""

    @staticmethod
    def forward(ctx, a, dtype):
        e = kernels(dtype)
        n, h, w, c = a.shape
        a = a.contiguous()
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=a.device)
        L.check(L.lib().oct_nhwc_to_nchw(e.dt, a.data_ptr(), out.data_ptr(), n, c, h, w, _stream()), "oct_nhwc_to_nchw")
        flush_counters()
        ctx.dtype = dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        return ToNHWC.apply(dout, ctx.dtype), None


class LazyAct:
    """act(scale * y + shift) that is never written: `y` is the raw convolution output and the consumers -- convolutions
    (and the residual input of one) -- apply the per-channel affine while they stage their input: the deferred-activation
    schedule of `engine.UNetEngine`, here between autograd ops.  relu=True: BatchNorm + ReLU (OCT_XF_AFFINE_RELU);
    relu=False: scale = 1, shift = the convolution's bias (OCT_XF_AFFINE).  Values are bit-identical to the materialised
    tensor (same fma, same bf16 rounding)."""
    __slots__ = ("y", "scale", "shift", "relu")

    def __init__(self, y, scale, shift, relu=True):
        self.y, self.scale, self.shift, self.relu = y, scale, shift, relu

    @property
    def shape(self):
        return self.y.shape


class ConvAffineAct(torch.autograd.Function):
    """out = act(affine(conv(cat(x0, x1), w)) (+ res)) with affine = train/eval BatchNorm (bn given) or
    `+ bias` (bn None).  3x3 pad 1, 1x1 or 7x3 pad (3,1) (ReLayNet_2017.py:155-160), chosen by the weight's shape.
    act = ACT_PRELU takes `alpha` (nn.PReLU's single slope) and returns its gradient.

    bn: the nn.BatchNorm2d container (running buffers are updated in train mode, momentum 0.1).
    A conv bias in front of a train-mode BN only moves the running mean; its gradient is zero.

    xf0 / xf1 = (scale, shift, relu): that input is a LazyAct's raw tensor, the affine (+ ReLU) is applied on load
    (forward, weight gradient) and the data gradient returned for it is the gradient w.r.t. the ACTIVATED tensor.
    lazy = "relu": train-mode BN + ReLU without residual -- return (y, scale, shift) instead of the activated tensor;
    backward then takes dA, re-derives the ReLU mask from y (no stored activation, no separate mask pass).
    lazy = "affine": no BN, no activation (a bare convolution with bias) -- return y; the caller pairs it with the bias.
    res_shift: the residual is a LazyAct(relu=False)'s raw tensor; its bias is folded into this op's shift."""

    @staticmethod
    def forward(ctx, dtype, bn, act, x0, x1, w, cbias, gamma, beta, res, alpha=None, xf0=None, xf1=None, lazy=False,
                res_shift=None):
        e = kernels(dtype)
        lib = L.lib()
        x0 = x0.contiguous()
        n, h, wd, c0 = x0.shape
        c1 = 0
        if x1 is not None:
            x1 = x1.contiguous()
            c1 = x1.shape[3]
        cout, cin, kh, kw = w.shape
        if cin != c0 + c1 or (kh, kw) not in ((1, 1), (3, 3), (7, 3)):
            raise RuntimeError(f"conv weight {tuple(w.shape)} does not fit an input with {c0 + c1} channels "
                               f"/ a 1x1, 3x3 or 7x3 kernel")
        taps = kh * kw
        kk = (kh, kw) if taps == 21 else None
        kd = dict(kh=kh, kw=kw) if kk else {}
        if act == L.ACT_PRELU and (alpha is None or res is not None):
            raise RuntimeError("PReLU needs its slope parameter (and takes no residual)")
        dev = x0.device
        src = Src(x0, c0, BNState(xf0[0], xf0[1], relu=xf0[2]) if xf0 else None,
                  x1, c1, BNState(xf1[0], xf1[1], relu=xf1[2]) if xf1 else None)
        if lazy == "relu" and not (bn is not None and bn.training and act == L.ACT_RELU and res is None):
            raise RuntimeError("a lazy BN + ReLU output needs a train-mode BatchNorm + ReLU without residual")
        if lazy == "affine" and not (bn is None and act == L.ACT_NONE and res is None):
            raise RuntimeError("a lazy affine output is a bare convolution (+ bias)")
        # 1-4 output channels (Attention_block's psi, the heads): three streaming kernels instead of GEMMs padded to 32 rows;
        # 5-12 only for bare convolutions (the class heads of ReLayNet / MGU-Net): a narrow Attention_block's W_g / W_x
        # (F_int = 8) keeps the kernel it has in BOTH activation schedules, whose results are compared bit for bit
        rowdot = (taps == 1 and x1 is None and xf0 is None and lib.oct_rowdot_ok(c0, cout) == 1 and not e.rowdot_off
                  and (cout <= 4 or bn is None))
        fold = fold16_ok(e, taps, kk, wd, c0, c1, cout)     # 16-channel 3x3: run as the pixel-pair-folded 32-channel convolution
        if fold:
            wf = fold16_weight(w, c0, c1)
            wp = packed(e, wf, L.PACK_1X1_FPROP if taps == 1 else L.PACK_CONV_FPROP, 2 * cout, 2 * cin, cache=False)
            srcf = _fold_src(x0, c0, xf0, x1, c1, xf1)
        else:
            wp = None if rowdot else packed(e, w, L.PACK_1X1_FPROP if taps == 1 else L.PACK_CONV_FPROP, cout, cin, kk=kk)
        y = e._act(n, h, wd, cout, dev)
        scale = torch.empty(cout, dtype=torch.float32, device=dev)
        shift = torch.empty_like(scale)
        mean = invstd = None
        train_bn = bn is not None and bn.training

        def conv(stats=None):
            if rowdot:
                L.check(lib.oct_rowdot_fwd(e.dt, x0.data_ptr(), w.data_ptr(), y.data_ptr(), L.ptr(stats), n * h * wd, c0, cout,
                                           _stream()), "oct_rowdot_fwd")
            elif fold:
                e._conv(srcf, wp, 2 * cout, taps, n, h, wd // 2, y.view(n, h, wd // 2, 2 * cout), stats=stats)
            else:
                e._conv(src, wp, cout, taps, n, h, wd, y, stats=stats, **kd)

        if train_bn and n * h * wd == 1:
            # torch's batch_norm refuses a single value per channel in training mode (functional.py, _verify_batch_size)
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {torch.Size([n, cout, h, wd])}")
        if train_bn:
            if fold:
                nblk = e._stat_blocks(2 * cout, n, h, wd // 2, srcf, taps)
                pf = torch.empty((nblk, 2, 2 * cout), dtype=torch.float32, device=dev)
                conv(pf)
                partials = pf.view(nblk, 2, 2, cout).sum(2)     # the two column parities of a channel
            else:
                nblk = lib.oct_rowdot_blocks(n * h * wd, c0) if rowdot else e._stat_blocks(cout, n, h, wd, src, taps, **kd)
                partials = torch.empty((nblk, 2, cout), dtype=torch.float32, device=dev)
                conv(partials)
            mean, invstd = torch.empty_like(scale), torch.empty_like(scale)
            L.check(lib.oct_bn_finalize(partials.data_ptr(), nblk, cout, float(n * h * wd), gamma.data_ptr(),
                                        beta.data_ptr(), BN_EPS, BN_MOMENTUM, bn.running_mean.data_ptr(),
                                        bn.running_var.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                        scale.data_ptr(), shift.data_ptr(), L.ptr(cbias), _stream()), "oct_bn_finalize")
            NBT_PENDING.append(bn.num_batches_tracked)   # bumped by one multi-tensor launch (flush_counters)
        else:
            conv()
            if bn is not None:
                L.check(lib.oct_bn_eval_coeffs(cout, gamma.data_ptr(), beta.data_ptr(),
                                               bn.running_mean.data_ptr(), bn.running_var.data_ptr(), BN_EPS,
                                               scale.data_ptr(), shift.data_ptr(), L.ptr(cbias), _stream()),
                        "oct_bn_eval_coeffs")
                if any(ctx.needs_input_grad):
                    # frozen statistics under autograd: backward needs xhat = (y + conv_bias - running_mean) * invstd for d(gamma)
                    invstd = torch.rsqrt(bn.running_var + BN_EPS)
                    mean = bn.running_mean - cbias.detach() if cbias is not None else bn.running_mean.clone()
            else:   # bare convolution (+ bias): constants instead of three tiny fill / copy launches per op
                scale = e._const(1.0, cout, dev)
                shift = cbias.detach() if cbias is not None else e._const(0.0, cout, dev)
        ctx.cfg = (dtype, bn, act, taps, train_bn, res is not None, cbias is not None, kk, lazy)
        ctx.xf = (xf0, xf1)
        ctx.rowdot = rowdot
        ctx.fold = fold
        if lazy == "relu":
            ctx.save_for_backward(x0, x1, w, y, None, mean, invstd, scale, gamma, shift, alpha)
            ctx.mark_non_differentiable(scale, shift)
            return y, scale, shift
        if lazy == "affine":
            ctx.save_for_backward(x0, x1, w, y, None, mean, invstd, scale, gamma, shift, alpha)
            return y
        out = e._act(n, h, wd, cout, dev)
        if res is not None:
            res = res.contiguous()
        if act == L.ACT_PRELU:
            L.check(lib.oct_affine_prelu_fwd(e.dt, y.data_ptr(), scale.data_ptr(), shift.data_ptr(), alpha.data_ptr(),
                                             out.data_ptr(), n * h * wd, cout, _stream()), "oct_affine_prelu_fwd")
        else:
            # res_shift: the residual is a raw tensor whose bias add was deferred (LazyAct, relu=False)
            L.check(lib.oct_affine_res_act_fwd(e.dt, y.data_ptr(), scale.data_ptr(), shift.data_ptr(), L.ptr(res),
                                               L.ptr(res_shift), act, out.data_ptr(), n * h * wd, cout, _stream()),
                    "oct_affine_res_act_fwd")
        ctx.save_for_backward(x0, x1, w, y, out, mean, invstd, scale, gamma, shift, alpha)
        return out

    @staticmethod
    def backward(ctx, dout, *_unused):
        dtype, bn, act, taps, train_bn, has_res, has_bias, kk, lazy = ctx.cfg
        xf0, xf1 = ctx.xf
        x0, x1, w, y, out, mean, invstd, scale, gamma, shift, alpha = ctx.saved_tensors
        kd = dict(kh=kk[0], kw=kk[1]) if kk else {}
        dalpha = None
        fuse_bias = False
        prelu_fused = False
        rowdot_bias = False
        fold_bias = False
        e = kernels(dtype)
        lib = L.lib()
        n, h, wd, c0 = x0.shape
        c1 = x1.shape[3] if x1 is not None else 0
        cout, cin = w.shape[0], w.shape[1]
        npix = n * h * wd
        dev = x0.device
        dout = dout.contiguous()
        if bn is not None and not train_bn and mean is None:
            raise RuntimeError("backward through an eval-mode BatchNorm whose forward ran under no_grad()")
        if lazy == "relu":
            dz = dout       # dA: the mask [scale*y + shift > 0] is applied inside the two BN-backward passes
        elif act == L.ACT_PRELU and bn is not None and lib.oct_prelu_bn_fused_ok(e.dt, cout) == 1:
            # dz = dout * (z > 0 ? 1 : alpha) is re-derived inside the two BatchNorm-backward passes (never written)
            prelu_fused = True
            dz = dout
            dalpha = torch.zeros(1, dtype=torch.float32, device=dev)
        elif act == L.ACT_PRELU:
            dz = torch.empty_like(dout)
            dalpha = torch.zeros(1, dtype=torch.float32, device=dev)
            L.check(lib.oct_affine_prelu_bwd(e.dt, dout.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                             alpha.data_ptr(), dz.data_ptr(), dalpha.data_ptr(), npix, cout, _stream()),
                    "oct_affine_prelu_bwd")
            dalpha = dalpha.reshape(alpha.shape)
        elif act != L.ACT_NONE:
            dz = torch.empty_like(dout)
            L.check(lib.oct_act_bwd(e.dt, dout.data_ptr(), out.data_ptr(), act, dz.data_ptr(), dz.numel(), _stream()),
                    "oct_act_bwd")
        else:
            dz = dout
        dres = dz if has_res else None
        dgamma = dbeta = dcb = None
        if bn is not None:
            # sums of dz and dz*xhat: the reduction kernel of the fused path with its ReLU mask held open
            nblk = lib.oct_dact_bn_reduce_blocks(n, h, wd, cout, 0)
            partials = torch.empty((nblk, 2, cout), dtype=torch.float32, device=dev)
            if prelu_fused:
                L.check(lib.oct_dact_bn_reduce_prelu(e.dt, dz.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                     alpha.data_ptr(), mean.data_ptr(), invstd.data_ptr(), partials.data_ptr(),
                                                     dalpha.data_ptr(), n, h, wd, cout, _stream()), "oct_dact_bn_reduce_prelu")
                dalpha = dalpha.reshape(alpha.shape)
            else:
                if lazy == "relu":
                    msc, msh = scale, shift
                else:       # dz is already masked: hold the reduction kernel's ReLU mask open (0*y + 1 > 0)
                    msc, msh = e._const(0.0, cout, dev), e._const(1.0, cout, dev)
                L.check(lib.oct_dact_bn_reduce(e.dt, dz.data_ptr(), None, y.data_ptr(), msc.data_ptr(), msh.data_ptr(),
                                               mean.data_ptr(), invstd.data_ptr(), None, partials.data_ptr(), n, h, wd, cout,
                                               _stream()), "oct_dact_bn_reduce")
            dgamma, dbeta = torch.empty_like(scale), torch.empty_like(scale)
            coef = torch.empty((3, cout), dtype=torch.float32, device=dev)
            L.check(lib.oct_bn_bwd_finalize(partials.data_ptr(), nblk, cout, float(npix), gamma.data_ptr(),
                                            mean.data_ptr(), invstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                            coef.data_ptr(), 0, _stream()), "oct_bn_bwd_finalize")
            if not train_bn:   # frozen (eval-mode) BatchNorm: a per-channel affine -- dy = k0 * dz, no statistics terms
                coef[1:].zero_()
            # dz stays intact when somebody else still reads it (the residual branch's gradient, autograd's own buffer)
            dy = torch.empty_like(dz) if (has_res or dz is dout) else dz
            if prelu_fused:
                L.check(lib.oct_bn_bwd_apply_prelu_to(e.dt, dy.data_ptr(), dz.data_ptr(), y.data_ptr(), coef.data_ptr(),
                                                      scale.data_ptr(), shift.data_ptr(), alpha.data_ptr(), npix, cout, _stream()),
                        "oct_bn_bwd_apply_prelu_to")
            else:
                L.check(lib.oct_bn_bwd_apply_to(e.dt, dy.data_ptr(), dz.data_ptr(), y.data_ptr(), coef.data_ptr(),
                                                scale.data_ptr() if lazy == "relu" else None,
                                                shift.data_ptr() if lazy == "relu" else None,
                                                npix, cout, _stream()), "oct_bn_bwd_apply_to")
            if has_bias:
                # train mode: a bias in front of BatchNorm cancels in (y - mean); frozen statistics: d(bias) = sum dy = k0 * sum dz
                dcb = torch.zeros(cout, dtype=torch.float32, device=dev) if train_bn else coef[0] * partials[:, 0, :].sum(0)
        else:
            dy = dz
            if has_bias and cin % 32 == 0 and not ctx.rowdot and not ctx.fold:
                # bias gradient = sum over pixels of dY: one extra MFMA against a ones fragment inside the weight-gradient
                # kernel instead of a separate pass over dY (the direct first-layer kernel has no such path: cin = 1, 3 ...)
                dcb = torch.zeros(cout, dtype=torch.float32, device=dev)
                fuse_bias = True
            elif has_bias and ctx.fold:
                fold_bias = True        # rides on the folded weight-gradient kernel (both column parities, added below)
            elif has_bias and ctx.rowdot:
                rowdot_bias = True      # rides on the streaming weight-gradient pass below
                dcb = torch.empty(cout, dtype=torch.float32, device=dev)
            elif has_bias:
                dcb = torch.empty(cout, dtype=torch.float32, device=dev)
                L.check(lib.oct_channel_sum(e.dt, dy.data_ptr(), dcb.data_ptr(), npix, cout, 0, _stream()),
                        "oct_channel_sum")
        if ctx.rowdot:
            dw = torch.empty_like(w)
            scratch = torch.empty((lib.oct_rowdot_blocks(npix, c0), cout * c0 + cout), dtype=torch.float32, device=dev)
            if rowdot_bias:
                L.check(lib.oct_rowdot_bwd_weight_bias(e.dt, dy.data_ptr(), x0.data_ptr(), dw.data_ptr(), dcb.data_ptr(),
                                                       scratch.data_ptr(), npix, c0, cout, 0, _stream()), "oct_rowdot_bwd_weight_bias")
            else:
                L.check(lib.oct_rowdot_bwd_weight(e.dt, dy.data_ptr(), x0.data_ptr(), dw.data_ptr(), scratch.data_ptr(), npix, c0,
                                                  cout, 0, _stream()), "oct_rowdot_bwd_weight")
            d0 = None
            if ctx.needs_input_grad[3]:
                d0 = e._act(n, h, wd, c0, dev)
                L.check(lib.oct_rowdot_bwd_data(e.dt, dy.data_ptr(), w.data_ptr(), d0.data_ptr(), npix, c0, cout, _stream()),
                        "oct_rowdot_bwd_data")
            return None, None, None, d0, None, dw, dcb, dgamma, dbeta, dres, dalpha, None, None, None, None
        if DEBUG is not None:
            DEBUG.append((f"{cout}x{cin}x{taps}@{h}x{wd} lazy={lazy} xf={bool(xf0)},{bool(xf1)} res={has_res}",
                          dict(dout=dout.float().clone(), dy=dy.float().clone(), y=y.float().clone(), x0=x0.float().clone(),
                               mean=None if mean is None else mean.clone(), invstd=None if invstd is None else invstd.clone(),
                               scale=scale.clone(), shift=shift.clone(),
                               dbeta=None if dbeta is None else dbeta.clone())))
        if ctx.fold:    # the pixel-pair-folded convolution (see fold16_ok): same tensors viewed (n, h, w/2, 2c), folded filter
            srcf = _fold_src(x0, c0, xf0, x1, c1, xf1)
            dyf = dy.view(n, h, wd // 2, 2 * cout)
            db2 = torch.zeros(2 * cout, dtype=torch.float32, device=dev) if fold_bias else None
            dwpf = e._wgrad(srcf, dyf, 2 * cout, taps, n, h, wd // 2, dbias=db2)
            if fold_bias:
                dcb = db2.view(2, cout).sum(0)
            dwf = torch.empty((2 * cout, 2 * cin) + tuple(w.shape[2:]), dtype=w.dtype, device=dev)
            e._unpack(L.PACK_CONV_FPROP if taps == 9 else L.PACK_1X1_FPROP, dwpf, dwf, 2 * cout, 2 * cin, False)
            dw = unfold16_wgrad(dwf, cout, c0, c1)
            d0 = d1 = None
            if ctx.needs_input_grad[3] or (x1 is not None and ctx.needs_input_grad[4]):
                wpd = packed(e, fold16_weight(w, c0, c1), L.PACK_CONV_DGRAD if taps == 9 else L.PACK_1X1_DGRAD, 2 * cout, 2 * cin, cache=False)
                d0 = e._act(n, h, wd, c0, dev)
                d1 = e._act(n, h, wd, c1, dev) if c1 else None
                e._conv(Src(dyf, 2 * cout), wpd, 2 * cin, taps, n, h, wd // 2, d0.view(n, h, wd // 2, 2 * c0),
                        y1=None if d1 is None else d1.view(n, h, wd // 2, 2 * c1), split=2 * c0 if c1 else 0)
            return None, None, None, d0, d1, dw, dcb, dgamma, dbeta, dres, dalpha, None, None, None, None
        src = Src(x0, c0, BNState(xf0[0], xf0[1], relu=xf0[2]) if xf0 else None,
                  x1, c1, BNState(xf1[0], xf1[1], relu=xf1[2]) if xf1 else None)
        dwp = e._wgrad(src, dy, cout, taps, n, h, wd, dbias=dcb if fuse_bias else None, partials_ok=not kk, **kd)
        dw = torch.empty_like(w)
        if kk:
            L.check(lib.oct_unpack_wgrad_kk(dwp.data_ptr(), dw.data_ptr(), cout, cin, kk[0], kk[1], 0, _stream()),
                    "oct_unpack_wgrad_kk")
        else:
            e._unpack(L.PACK_CONV_FPROP if taps == 9 else L.PACK_1X1_FPROP, dwp, dw, cout, cin, False)
        d0 = d1 = None
        if ctx.needs_input_grad[3] or (x1 is not None and ctx.needs_input_grad[4]):  # x0 / x1
            wp = packed(e, w, L.PACK_1X1_DGRAD if taps == 1 else L.PACK_CONV_DGRAD, cout, cin, kk=kk)
            d0 = e._act(n, h, wd, c0, dev)
            d1 = e._act(n, h, wd, c1, dev) if c1 else None
            e._conv(Src(dy, cout), wp, cin, taps, n, h, wd, d0, y1=d1, split=c0 if c1 else 0, **kd)
        return None, None, None, d0, d1, dw, dcb, dgamma, dbeta, dres, dalpha, None, None, None, None


class MaxPool(torch.autograd.Function):
    """nn.MaxPool2d(k) (stride k, torch's floor mode: trailing rows / columns that no whole window covers are ignored)."""

    @staticmethod
    def forward(ctx, dtype, k, a):
        e = kernels(dtype)
        a = a.contiguous()
        n, h, w, c = a.shape
        if h < k or w < k:
            raise RuntimeError(f"max-pool window {k} is larger than the {h}x{w} input")
        out = e._act(n, h // k, w // k, c, a.device)
        L.check(L.lib().oct_maxpool_fwd(e.dt, a.data_ptr(), out.data_ptr(), n, h, w, c, k, _stream()), "oct_maxpool_fwd")
        ctx.cfg = (dtype, k)
        ctx.save_for_backward(a)
        return out

    @staticmethod
    def backward(ctx, dout):
        dtype, k = ctx.cfg
        (a,) = ctx.saved_tensors
        e = kernels(dtype)
        n, h, w, c = a.shape
        dout = dout.contiguous()
        da = torch.empty_like(a)
        L.check(L.lib().oct_maxpool_bwd(e.dt, a.data_ptr(), dout.data_ptr(), da.data_ptr(), n, h, w, c, k, _stream()),
                "oct_maxpool_bwd")
        return None, None, da


class MaxPoolIdx(torch.autograd.Function):
    """nn.MaxPool2d(k, k, return_indices=True) on NHWC: (pooled, idx) with torch's per-plane index iy*W + ix.
    Backward = scatter of the pooled gradient to the recorded winners."""

    @staticmethod
    def forward(ctx, dtype, k, a):
        e = kernels(dtype)
        a = a.contiguous()
        n, h, w, c = a.shape
        if h % k or w % k:
            raise RuntimeError(f"max-pool window {k} does not divide {h}x{w}")
        out = e._act(n, h // k, w // k, c, a.device)
        idx = torch.empty((n, h // k, w // k, c), dtype=torch.int64, device=a.device)
        L.check(L.lib().oct_maxpool_idx_fwd(e.dt, a.data_ptr(), out.data_ptr(), idx.data_ptr(), n, h, w, c, k, _stream()),
                "oct_maxpool_idx_fwd")
        ctx.cfg = (dtype, (n, h, w, c))
        ctx.save_for_backward(idx)
        ctx.mark_non_differentiable(idx)
        return out, idx

    @staticmethod
    def backward(ctx, dout, _didx):
        dtype, (n, h, w, c) = ctx.cfg
        (idx,) = ctx.saved_tensors
        e = kernels(dtype)
        dout = dout.contiguous()
        da = torch.zeros((n, h, w, c), dtype=e.tdt, device=dout.device)
        L.check(L.lib().oct_index_scatter(e.dt, dout.data_ptr(), idx.data_ptr(), da.data_ptr(), n, idx.shape[1] * idx.shape[2],
                                          h * w, c, _stream()), "oct_index_scatter")
        return None, None, da


class MaxUnpool(torch.autograd.Function):
    """nn.MaxUnpool2d(k, k) on NHWC: v (n,hp,wp,c) scattered to (n, hp*k, wp*k, c) at idx; backward = gather."""

    @staticmethod
    def forward(ctx, dtype, k, v, idx):
        e = kernels(dtype)
        v, idx = v.contiguous(), idx.contiguous()
        n, hp, wp, c = v.shape
        if idx.shape != v.shape or idx.dtype != torch.int64:
            raise RuntimeError(f"indices must be int64 of the pooled shape {tuple(v.shape)}, got {idx.dtype} {tuple(idx.shape)}")
        out = torch.zeros((n, hp * k, wp * k, c), dtype=e.tdt, device=v.device)
        L.check(L.lib().oct_index_scatter(e.dt, v.data_ptr(), idx.data_ptr(), out.data_ptr(), n, hp * wp, hp * k * wp * k, c,
                                          _stream()), "oct_index_scatter")
        ctx.cfg = (dtype, k)
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, dout):
        dtype, k = ctx.cfg
        (idx,) = ctx.saved_tensors
        e = kernels(dtype)
        dout = dout.contiguous()
        n, hp, wp, c = idx.shape
        dv = e._act(n, hp, wp, c, dout.device)
        L.check(L.lib().oct_index_gather(e.dt, dout.data_ptr(), idx.data_ptr(), dv.data_ptr(), n, hp * wp, hp * k * wp * k, c,
                                         _stream()), "oct_index_gather")
        return None, None, dv, None


class MaxPoolCode(torch.autograd.Function):
    """MaxPoolIdx with the winner kept as a one-byte WINDOW CODE (dy*k + dx) instead of torch's int64 plane index: for the
    encoder -> decoder path INSIDE a network, where the indices never reach the caller (oct_hip.h, oct_maxpool_code_fwd).
    Backward = dense window scatter (no zero fill)."""

    @staticmethod
    def forward(ctx, dtype, k, a):
        e = kernels(dtype)
        a = a.contiguous()
        n, h, w, c = a.shape
        if h % k or w % k:
            raise RuntimeError(f"max-pool window {k} does not divide {h}x{w}")
        out = e._act(n, h // k, w // k, c, a.device)
        code = torch.empty((n, h // k, w // k, c), dtype=torch.uint8, device=a.device)
        L.check(L.lib().oct_maxpool_code_fwd(e.dt, a.data_ptr(), out.data_ptr(), code.data_ptr(), n, h, w, c, k, _stream()),
                "oct_maxpool_code_fwd")
        ctx.cfg = (dtype, k, (n, h, w, c))
        ctx.save_for_backward(code)
        ctx.mark_non_differentiable(code)
        return out, code

    @staticmethod
    def backward(ctx, dout, _dcode):
        dtype, k, (n, h, w, c) = ctx.cfg
        (code,) = ctx.saved_tensors
        e = kernels(dtype)
        dout = dout.contiguous()
        da = torch.empty((n, h, w, c), dtype=e.tdt, device=dout.device)
        L.check(L.lib().oct_window_scatter(e.dt, dout.data_ptr(), code.data_ptr(), da.data_ptr(), n, h // k, w // k, c, k, _stream()),
                "oct_window_scatter")
        return None, None, da


class MaxUnpoolCode(torch.autograd.Function):
    """MaxUnpool2d(k, k) from window codes: the whole output is written (value at the code, zeros elsewhere); backward = gather."""

    @staticmethod
    def forward(ctx, dtype, k, v, code):
        e = kernels(dtype)
        v, code = v.contiguous(), code.contiguous()
        n, hp, wp, c = v.shape
        if code.shape != v.shape or code.dtype != torch.uint8:
            raise RuntimeError(f"window codes must be uint8 of the pooled shape {tuple(v.shape)}, got {code.dtype} {tuple(code.shape)}")
        out = torch.empty((n, hp * k, wp * k, c), dtype=e.tdt, device=v.device)
        L.check(L.lib().oct_window_scatter(e.dt, v.data_ptr(), code.data_ptr(), out.data_ptr(), n, hp, wp, c, k, _stream()),
                "oct_window_scatter")
        ctx.cfg = (dtype, k)
        ctx.save_for_backward(code)
        return out

    @staticmethod
    def backward(ctx, dout):
        dtype, k = ctx.cfg
        (code,) = ctx.saved_tensors
        e = kernels(dtype)
        dout = dout.contiguous()
        n, hp, wp, c = code.shape
        dv = e._act(n, hp, wp, c, dout.device)
        L.check(L.lib().oct_window_gather(e.dt, dout.data_ptr(), code.data_ptr(), dv.data_ptr(), n, hp, wp, c, k, _stream()),
                "oct_window_gather")
        return None, None, dv, None


class BilinearUp(torch.autograd.Function):
    """x`factor` bilinear up-sampling with align_corners=True."""

    @staticmethod
    def forward(ctx, dtype, factor, x):
        e = kernels(dtype)
        x = x.contiguous()
        n, h, w, c = x.shape
        out = e._act(n, h * factor, w * factor, c, x.device)
        L.check(L.lib().oct_bilinear_up_fwd(e.dt, x.data_ptr(), out.data_ptr(), n, h, w, c, factor, _stream()),
                "oct_bilinear_up_fwd")
        ctx.cfg = (dtype, factor, (n, h, w, c))
        return out

    @staticmethod
    def backward(ctx, dout):
        dtype, factor, (n, h, w, c) = ctx.cfg
        e = kernels(dtype)
        dout = dout.contiguous()
        dx = e._act(n, h, w, c, dout.device)
        L.check(L.lib().oct_bilinear_up_bwd(e.dt, dout.data_ptr(), dx.data_ptr(), n, h, w, c, factor, _stream()),
                "oct_bilinear_up_bwd")
        return None, None, dx


class BilinearResize(torch.autograd.Function):
    """F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=True) on NHWC (MGUNet_2021.py:180,184,188)."""

    @staticmethod
    def forward(ctx, dtype, size, x):
        e = kernels(dtype)
        x = x.contiguous()
        n, h, w, c = x.shape
        ho, wo = int(size[0]), int(size[1])
        out = e._act(n, ho, wo, c, x.device)
        L.check(L.lib().oct_bilinear_resize_fwd(e.dt, x.data_ptr(), out.data_ptr(), n, h, w, c, ho, wo, _stream()),
                "oct_bilinear_resize_fwd")
        ctx.cfg = (dtype, (ho, wo), (n, h, w, c))
        return out

    @staticmethod
    def backward(ctx, dout):
        dtype, (ho, wo), (n, h, w, c) = ctx.cfg
        e = kernels(dtype)
        dout = dout.contiguous()
        dx = e._act(n, h, w, c, dout.device)
        L.check(L.lib().oct_bilinear_resize_bwd(e.dt, dout.data_ptr(), dx.data_ptr(), n, h, w, c, ho, wo, _stream()),
                "oct_bilinear_resize_bwd")
        return None, None, dx


class Deconv(torch.autograd.Function):
    """nn.ConvTranspose2d(cin, cout, kernel_size=k, stride=k) with bias, k in {2, 4}.  k=2 stores
    straight from the GEMM epilogue (depth-to-space fused); k=4 runs the (16*cout x cin) GEMM as a
    1x1 convolution followed by oct_depth_to_space."""

    @staticmethod
    def forward(ctx, dtype, x, w, bias):
        e = kernels(dtype)
        x = x.contiguous()
        n, h, wd, cin = x.shape
        if w.shape[0] != cin or w.shape[2] != w.shape[3] or w.shape[2] not in (2, 4):
            raise RuntimeError(f"transposed-conv weight {tuple(w.shape)} does not fit {cin} input channels / k in (2, 4)")
        cout, k = w.shape[1], w.shape[2]
        out = e._act(n, h * k, wd * k, cout, x.device)
        fold = fold16_deconv_ok(e, k, wd, cin, cout)
        if fold:    # 16-channel transposed convolution, pixel-pair folded (see fold16_ok): 2 cin -> 2 cout on (n, h, w/2)
            wp = packed(e, fold16_deconv_weight(w), L.PACK_DECONV_FPROP, 2 * cout, 2 * cin, cache=False)
            e._conv(Src(x.view(n, h, wd // 2, 2 * cin), 2 * cin), wp, 8 * cout, 1, n, h, wd // 2, out.view(n, 2 * h, wd, 2 * cout),
                    out_mode=L.OUT_D2S, bias=bias.detach().repeat(2).contiguous())
            w1 = None
        elif k == 2:
            wp = packed(e, w, L.PACK_DECONV_FPROP, cout, cin)
            e._conv(Src(x, cin), wp, 4 * cout, 1, n, h, wd, out, out_mode=L.OUT_D2S, bias=bias)
            w1 = None
        else:
            w1 = w.detach().permute(2, 3, 1, 0).reshape(k * k * cout, cin, 1, 1).contiguous()
            wp = packed(e, w1, L.PACK_1X1_FPROP, k * k * cout, cin, cache=False)   # w1 is a temporary
            y = e._act(n, h, wd, k * k * cout, x.device)
            e._conv(Src(x, cin), wp, k * k * cout, 1, n, h, wd, y)
            L.check(L.lib().oct_depth_to_space(e.dt, y.data_ptr(), bias.data_ptr(), out.data_ptr(), n, h, wd, cout, k,
                                               _stream()), "oct_depth_to_space")
        ctx.cfg = (dtype, k)
        ctx.fold = fold
        ctx.save_for_backward(x, w, w1)
        return out

    @staticmethod
    def backward(ctx, dout):
        dtype, k = ctx.cfg
        x, w, w1 = ctx.saved_tensors
        e = kernels(dtype)
        lib = L.lib()
        n, h, wd, cin = x.shape
        cout = w.shape[1]
        dev = x.device
        dout = dout.contiguous()
        dw = torch.empty_like(w)
        db = torch.zeros(cout, dtype=torch.float32, device=dev)
        dx = None
        if ctx.fold:
            xf = x.view(n, h, wd // 2, 2 * cin)
            df = dout.view(n, 2 * h, wd, 2 * cout)
            db2 = torch.zeros(2 * cout, dtype=torch.float32, device=dev)
            dwp = e._wgrad(Src(xf, 2 * cin), df, 8 * cout, 1, n, h, wd // 2, dy_mode=L.IN_S2D, dbias=db2)
            dwf = torch.empty((2 * cin, 2 * cout, 2, 2), dtype=w.dtype, device=dev)
            e._unpack(L.PACK_DECONV_FPROP, dwp, dwf, 2 * cout, 2 * cin, False)
            dw = unfold16_deconv_wgrad(dwf, cin, cout)
            db = db2.view(2, cout).sum(0)
            if ctx.needs_input_grad[1]:
                wp = packed(e, fold16_deconv_weight(w), L.PACK_DECONV_DGRAD, 2 * cout, 2 * cin, cache=False)
                dx = e._act(n, h, wd, cin, dev)
                e._conv(Src(df, 2 * cout), wp, 2 * cin, 1, n, h, wd // 2, dx.view(n, h, wd // 2, 2 * cin), in_mode=L.IN_S2D)
        elif k == 2:
            dwp = e._wgrad(Src(x, cin), dout, 4 * cout, 1, n, h, wd, dy_mode=L.IN_S2D, dbias=db)
            e._unpack(L.PACK_DECONV_FPROP, dwp, dw, cout, cin, False)
            if ctx.needs_input_grad[1]:
                wp = packed(e, w, L.PACK_DECONV_DGRAD, cout, cin)
                dx = e._act(n, h, wd, cin, dev)
                e._conv(Src(dout, cout), wp, cin, 1, n, h, wd, dx, in_mode=L.IN_S2D)
        else:
            rows = k * k * cout
            L.check(lib.oct_channel_sum(e.dt, dout.data_ptr(), db.data_ptr(), n * h * k * wd * k, cout, 0, _stream()),
                    "oct_channel_sum")
            dyd = e._act(n, h, wd, rows, dev)
            L.check(lib.oct_space_to_depth(e.dt, dout.data_ptr(), dyd.data_ptr(), n, h, wd, cout, k, _stream()),
                    "oct_space_to_depth")
            dwp = e._wgrad(Src(x, cin), dyd, rows, 1, n, h, wd)
            g1 = torch.empty_like(w1)
            e._unpack(L.PACK_1X1_FPROP, dwp, g1, rows, cin, False)
            dw = g1.reshape(k, k, cout, cin).permute(3, 2, 0, 1).contiguous()
            if ctx.needs_input_grad[1]:
                wp = packed(e, w1, L.PACK_1X1_DGRAD, rows, cin, cache=False)
                dx = e._act(n, h, wd, cin, dev)
                e._conv(Src(dyd, rows), wp, cin, 1, n, h, wd, dx)
        return None, dx, dw, db


class Gate(torch.autograd.Function):
    """out[n,h,w,c] = x[n,h,w,c] * p[n,h,w,0]  (Attention_block's `x * psi`, common.py:91)."""

    @staticmethod
    def forward(ctx, dtype, x, p):
        e = kernels(dtype)
        x, p = x.contiguous(), p.contiguous()
        n, h, w, c = x.shape
        out = torch.empty_like(x)
        L.check(L.lib().oct_gate_fwd(e.dt, x.data_ptr(), p.data_ptr(), out.data_ptr(), n * h * w, c, _stream()),
                "oct_gate_fwd")
        ctx.dtype = dtype
        ctx.save_for_backward(x, p)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, p = ctx.saved_tensors
        e = kernels(ctx.dtype)
        n, h, w, c = x.shape
        dout = dout.contiguous()
        dx, dp = torch.empty_like(x), torch.empty_like(p)
        L.check(L.lib().oct_gate_bwd(e.dt, dout.data_ptr(), x.data_ptr(), p.data_ptr(), dx.data_ptr(), dp.data_ptr(),
                                     n * h * w, c, _stream()), "oct_gate_bwd")
        return None, dx, dp


# ---- functional spellings -----------------------------------------------------------------------
def conv_bn_act(dtype, x0, conv, bn=None, act=L.ACT_NONE, x1=None, res=None, prelu=None, lazy=False):
    """conv: nn.Conv2d container (3x3 pad 1, 1x1, or 7x3 pad (3,1)); bn: nn.BatchNorm2d container or None;
    prelu: nn.PReLU container (num_parameters 1) -> act becomes PReLU.
    x0 / x1 may be LazyAct (BN + ReLU applied on load); lazy=True returns one (only for a consumer that is a
    convolution; falls back to the materialised tensor when BN is in eval mode or the 7x3 / fp32 kernels would run)."""
    if prelu is not None:
        if prelu.weight.numel() != 1:
            raise NotImplementedError("per-channel PReLU is not on the HIP path (the reference uses nn.PReLU())")
        act = L.ACT_PRELU
    xf0 = xf1 = res_shift = None
    if isinstance(x0, LazyAct):
        x0, xf0 = x0.y, (x0.scale, x0.shift, x0.relu)
    if isinstance(x1, LazyAct):
        x1, xf1 = x1.y, (x1.scale, x1.shift, x1.relu)
    if isinstance(res, LazyAct):
        if res.relu:
            raise NotImplementedError("a deferred BN + ReLU tensor cannot be a residual input: materialise() it")
        res, res_shift = res.y, res.shift
    kind = False
    if lazy and LAZY[0] and tuple(conv.weight.shape[2:]) in ((1, 1), (3, 3)) and res is None and prelu is None:
        if bn is not None and bn.training and act == L.ACT_RELU:
            kind = "relu"
        elif bn is None and act == L.ACT_NONE and LAZY[0] is True:
            kind = "affine"
    r = ConvAffineAct.apply(dtype, bn, act, x0, x1, conv.weight, conv.bias,
                            bn.weight if bn is not None else None, bn.bias if bn is not None else None, res,
                            prelu.weight if prelu is not None else None, xf0, xf1, kind, res_shift)
    if kind == "relu":
        return LazyAct(*r)
    if kind == "affine":
        e = kernels(dtype)
        cout = conv.weight.shape[0]
        shift = conv.bias.detach() if conv.bias is not None else e._const(0.0, cout, r.device)
        return LazyAct(r, e._const(1.0, cout, r.device), shift, relu=False)
    return r


# Dropout2d (SD_Layer_Net/common.py:13,17,34: between BatchNorm and the activation): a keep flag per (image, channel), scaled
# by 1 / (1 - p).  DROPOUT_MASK_HOOK[0] (tests): callable(n, c, p) -> [n, c] tensor of 0 / 1 keep flags, asked once per
# Dropout2d application in forward order; None draws from torch's generator of the tensor's device.
DROPOUT_MASK_HOOK = [None]


def dropout2d(a, p: float, training: bool):
    """a: materialised NHWC tensor.  Identity in eval mode and for p = 0 (the reference default)."""
    if not training or p <= 0.0:
        return a
    n, c = a.shape[0], a.shape[-1]
    if p >= 1.0:
        return a * 0.0
    hook = DROPOUT_MASK_HOOK[0]
    keep = hook(n, c, p) if hook is not None else torch.bernoulli(torch.full((n, c), 1.0 - p, device=a.device))
    keep = keep.to(device=a.device, dtype=torch.float32)
    if tuple(keep.shape) != (n, c):
        raise RuntimeError(f"dropout mask must be [{n}, {c}], got {tuple(keep.shape)}")
    return a * (keep / (1.0 - p)).to(a.dtype).view(n, 1, 1, c)


def materialise(dtype, a):
    """LazyAct -> the activated NHWC tensor (for a consumer that is not a convolution); tensors pass through."""
    if not isinstance(a, LazyAct):
        return a
    return _Materialise.apply(dtype, a.y, a.scale, a.shift, a.relu)


class _Materialise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dtype, y, scale, shift, relu):
        e = kernels(dtype)
        n, h, w, c = y.shape
        out = e._act(n, h, w, c, y.device)
        L.check(L.lib().oct_affine_act_fwd(e.dt, y.data_ptr(), scale.data_ptr(), shift.data_ptr(), None,
                                           L.ACT_RELU if relu else L.ACT_NONE, out.data_ptr(), n * h * w, c, _stream()),
                "oct_affine_act_fwd")
        return out

    @staticmethod
    def backward(ctx, dout):
        return None, dout, None, None, None      # the producer's backward takes dA and masks it itself


def to_nhwc(x, dtype):
    return ToNHWC.apply(x, dtype)


def to_nchw(a, dtype):
    return ToNCHW.apply(a, dtype)
