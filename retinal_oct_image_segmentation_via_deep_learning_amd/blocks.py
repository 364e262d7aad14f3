"""Host-side mirrors of the reference's block families, executed by the HIP ops layer (ops.py).

  MGUNet_2021.py:42-108,314-352   UnetConv, UnetUp, UnetUp4, init_weights (+ weights_init_*)
  SD_Layer_Net/common.py:6-41,64-91  conv_block, up_conv, Attention_block
  SD_Layer_Net/unet.py:8-150      U_Net, AttU_Net

Same constructor arguments, sub-module names (hence state_dict keys and seeded default init) and
forward semantics as the reference classes.  The torch.nn members are parameter containers: no
torch.nn forward is ever called.  Every block accepts NCHW float tensors like the reference; inside
a network the blocks hand NHWC compute-dtype tensors to each other through `.nhwc(...)`.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn import init

from . import _lib as L
from . import ops


class HipModule(nn.Module):
    """compute_dtype: 'bf16' (production) or 'f32' (parity mode: fp32 storage, exact fp32 MFMA)."""
    compute_dtype = "bf16"

    def set_compute_dtype(self, dtype: str):
        if dtype not in ("bf16", "f32"):
            raise ValueError("dtype must be 'bf16' or 'f32'")
        for m in self.modules():
            if isinstance(m, HipModule):
                m.compute_dtype = dtype
        return self

    def _in(self, x):
        if x.dim() != 4:
            raise RuntimeError(f"expected a 4-D (B,C,H,W) input, got {tuple(x.shape)}")
        return ops.to_nhwc(x, self.compute_dtype)

    def _out(self, a):
        return ops.to_nchw(a, self.compute_dtype)


def _drops(mod: nn.Module) -> bool:
    """True when a Dropout2d of the block is active (p > 0, training): the block then runs its general schedule."""
    return any(isinstance(m, nn.Dropout2d) and m.p > 0 and m.training for m in mod.modules())


# ------------------------------------------------------------------------------------------------
# MGUNet_2021.py
# ------------------------------------------------------------------------------------------------
class UnetConv(HipModule):
    """MGUNet_2021.py:42-70: two 3x3 convolutions (bias) each followed by [BatchNorm] + ReLU."""

    def __init__(self, in_channels, out_channels, is_batchnorm=True, compute_dtype="bf16"):
        super().__init__()
        self.compute_dtype = compute_dtype
        if is_batchnorm:
            self.conv1 = nn.Sequential(nn.Conv2d(in_channels, out_channels, 3, 1, 1), nn.BatchNorm2d(out_channels),
                                       nn.ReLU(inplace=True))
            self.conv2 = nn.Sequential(nn.Conv2d(out_channels, out_channels, 3, 1, 1), nn.BatchNorm2d(out_channels),
                                       nn.ReLU(inplace=True))
        else:
            self.conv1 = nn.Sequential(nn.Conv2d(in_channels, out_channels, 3, 1, 1), nn.ReLU(inplace=True))
            self.conv2 = nn.Sequential(nn.Conv2d(out_channels, out_channels, 3, 1, 1), nn.ReLU(inplace=True))
        self._bn = bool(is_batchnorm)

    def nhwc(self, a, a1=None):
        dt = self.compute_dtype
        a = ops.conv_bn_act(dt, a, self.conv1[0], self.conv1[1] if self._bn else None, L.ACT_RELU, x1=a1)
        return ops.conv_bn_act(dt, a, self.conv2[0], self.conv2[1] if self._bn else None, L.ACT_RELU)

    def forward(self, x):
        return self._out(self.nhwc(self._in(x)))


class _UnetUpBase(HipModule):
    _k = 2

    def __init__(self, in_channels, out_channels, is_deconv=True, compute_dtype="bf16"):
        super().__init__()
        self.compute_dtype = compute_dtype
        k = self._k
        if is_deconv:
            self.up = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=k, stride=k)
        else:
            self.up = nn.Sequential(nn.UpsamplingBilinear2d(scale_factor=k), nn.Conv2d(in_channels, out_channels, 1))
        self.conv = UnetConv(in_channels, out_channels, True, compute_dtype)
        self._deconv = bool(is_deconv)

    def nhwc(self, a1, a2):
        dt = self.compute_dtype
        if self._deconv:
            u = ops.Deconv.apply(dt, a1, self.up.weight, self.up.bias)
        else:
            u = ops.conv_bn_act(dt, ops.BilinearUp.apply(dt, self._k, a1), self.up[1])
        if u.shape[1:3] != a2.shape[1:3] or u.shape[0] != a2.shape[0]:
            # torch.cat([x2, x1], dim=1), MGUNet_2021.py:87,106
            raise RuntimeError(f"Sizes of tensors must match except in dimension 1. Expected size {a2.shape[1]}x"
                               f"{a2.shape[2]} but got size {u.shape[1]}x{u.shape[2]}")
        return self.conv.nhwc(a2, u)        # skip first, up-sampled second

    def forward(self, x1, x2):
        return self._out(self.nhwc(self._in(x1), self._in(x2)))


class UnetUp(_UnetUpBase):
    """MGUNet_2021.py:72-89: ConvTranspose2d k2s2 (or bilinear x2 + 1x1 conv), cat([x2, up(x1)]), UnetConv."""
    _k = 2


class UnetUp4(_UnetUpBase):
    """MGUNet_2021.py:91-108: the x4 variant (ConvTranspose2d k4s4 or bilinear x4 + 1x1 conv)."""
    _k = 4


_CONV_INIT = {
    "normal": lambda w: init.normal_(w, 0.0, 0.02),
    "xavier": lambda w: init.xavier_normal_(w, gain=1),
    "kaiming": lambda w: init.kaiming_normal_(w, a=0, mode="fan_in"),
}


def _initializer(kind: str):
    """The reference selects by class-NAME substring (MGUNet_2021.py:314-342): anything whose name contains
    'Conv' or 'Linear' gets the weight rule of `kind` (so does ConvTranspose2d -- and a container class
    such as UnetConv raises AttributeError on `.weight`, there as here); 'BatchNorm' gets N(1, 0.02) / 0."""
    def fn(m):
        name = type(m).__name__
        if "Conv" in name or "Linear" in name:
            _CONV_INIT[kind](m.weight.data)
        elif "BatchNorm" in name:
            init.normal_(m.weight.data, 1.0, 0.02)
            init.constant_(m.bias.data, 0.0)
    fn.__name__ = f"weights_init_{kind}"
    return fn


weights_init_normal = _initializer("normal")
weights_init_xavier = _initializer("xavier")
weights_init_kaiming = _initializer("kaiming")


def init_weights(net, init_type="normal"):
    """MGUNet_2021.py:344-352: `net.apply` of the rule; unknown names raise NotImplementedError."""
    if init_type not in _CONV_INIT:
        raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
    net.apply({"normal": weights_init_normal, "xavier": weights_init_xavier, "kaiming": weights_init_kaiming}[init_type])


# ------------------------------------------------------------------------------------------------
# SD_Layer_Net/common.py
# ------------------------------------------------------------------------------------------------
# act != nn.ReLU / Dropout2d(p > 0) (common.py:7,13,17,29,34): the GENERAL schedule -- convolutions and BatchNorm on the HIP
# kernels with nothing deferred, Dropout2d (ops.dropout2d) and the activation module the constructor made (`act()`, as the
# reference does) applied to the materialised NHWC tensor.  The activation must therefore be elementwise; the ones that look
# at a dimension are refused.  nn.ReLU without dropout keeps the fused / deferred schedule.
_DIM_ACTS = (nn.Softmax, nn.Softmax2d, nn.LogSoftmax, nn.Softmin, nn.GLU, nn.Threshold, nn.MultiheadAttention)


def _check_act(act):
    if not (isinstance(act, type) and issubclass(act, nn.Module)):
        raise TypeError("act must be an nn.Module class constructed without arguments (the reference calls act())")
    if issubclass(act, _DIM_ACTS):
        raise NotImplementedError(f"act={act.__name__} is not elementwise: not on the HIP path")


class conv_block(HipModule):
    """common.py:6-25: init_conv (3x3, bias) -> [conv-BN-Dropout2d-act-conv-BN-Dropout2d] + init_conv -> act."""

    def __init__(self, ch_in, ch_out, act=nn.ReLU, drop_rate=0.0, kernel_size=3, compute_dtype="bf16"):
        super().__init__()
        self.compute_dtype = compute_dtype
        _check_act(act)
        self._relu = act is nn.ReLU
        if kernel_size != 3:
            raise NotImplementedError("only kernel_size=3 (the reference default) is on the HIP path")
        self.init_conv = nn.Conv2d(ch_in, ch_out, kernel_size=3, stride=1, padding=1, bias=True)
        self.conv = nn.Sequential(
            nn.Conv2d(ch_out, ch_out, kernel_size=3, stride=1, padding=1, bias=True), nn.BatchNorm2d(ch_out),
            nn.Dropout2d(drop_rate, inplace=True), act(),
            nn.Conv2d(ch_out, ch_out, kernel_size=3, stride=1, padding=1, bias=True), nn.BatchNorm2d(ch_out),
            nn.Dropout2d(drop_rate, inplace=True))
        self.activation = act()

    def nhwc(self, a, a1=None):
        """a / a1: NHWC tensors or ops.LazyAct (BN + ReLU of the producer applied on load)."""
        dt = self.compute_dtype
        if not self._relu or _drops(self):   # general schedule (see _check_act)
            i = ops.conv_bn_act(dt, a, self.init_conv, x1=a1)
            t = ops.conv_bn_act(dt, i, self.conv[0], self.conv[1])
            t = self.conv[3](ops.dropout2d(t, self.conv[2].p, self.conv[2].training))
            u = ops.conv_bn_act(dt, t, self.conv[4], self.conv[5])
            return self.activation(ops.dropout2d(u, self.conv[6].p, self.conv[6].training) + i)
        # init_conv's bias add is deferred too: the next convolution and the residual sum apply it (OCT_XF_AFFINE)
        i = ops.conv_bn_act(dt, a, self.init_conv, x1=a1, lazy=True)
        # relu(bn(conv0(i))) has one consumer, a convolution: it is never written (deferred activation)
        t = ops.conv_bn_act(dt, i, self.conv[0], self.conv[1], L.ACT_RELU, lazy=True)
        return ops.conv_bn_act(dt, t, self.conv[4], self.conv[5], L.ACT_RELU, res=i)

    def forward(self, x):
        return self._out(self.nhwc(self._in(x)))


class up_conv(HipModule):
    """common.py:28-41: bilinear x`scale_factor` (align_corners=True) -> conv3x3(bias) -> BN -> Dropout2d -> act."""

    def __init__(self, ch_in, ch_out, act=nn.ReLU, drop_rate=0.0, scale_factor=2, compute_dtype="bf16"):
        super().__init__()
        self.compute_dtype = compute_dtype
        _check_act(act)
        self._relu = act is nn.ReLU
        if int(scale_factor) != scale_factor or scale_factor < 1:
            raise NotImplementedError("only integer scale factors are on the HIP path")
        self._factor = int(scale_factor)
        self.up = nn.Sequential(
            nn.Upsample(scale_factor=scale_factor, mode="bilinear", align_corners=True),
            nn.Conv2d(ch_in, ch_out, kernel_size=3, stride=1, padding=1, bias=True), nn.BatchNorm2d(ch_out),
            nn.Dropout2d(drop_rate, inplace=True), act())

    def nhwc(self, a, lazy=False):
        """lazy=True: the caller feeds the result to convolutions only (U_Net / AttU_Net: the gate's W_g and the
        concatenating conv_block) and gets an ops.LazyAct."""
        dt = self.compute_dtype
        if not self._relu or _drops(self):   # general schedule (see _check_act): a materialised tensor, also for lazy=True
            t = ops.conv_bn_act(dt, ops.BilinearUp.apply(dt, self._factor, a), self.up[1], self.up[2])
            return self.up[4](ops.dropout2d(t, self.up[3].p, self.up[3].training))
        return ops.conv_bn_act(dt, ops.BilinearUp.apply(dt, self._factor, a), self.up[1], self.up[2], L.ACT_RELU, lazy=lazy)

    def forward(self, x):
        return self._out(self.nhwc(self._in(x)))


class Attention_block(HipModule):
    """common.py:64-91: psi = sigmoid(BN(conv1x1(relu(BN(W_g g) + BN(W_x x))))); returns x * psi.
    The reference's own callers pass F_g / F_l (unet.py:92 -- a TypeError there); both spellings work here."""

    def __init__(self, channels_g=None, channels_x=None, F_int=None, F_g=None, F_l=None, compute_dtype="bf16"):
        super().__init__()
        self.compute_dtype = compute_dtype
        channels_g = F_g if channels_g is None else channels_g
        channels_x = F_l if channels_x is None else channels_x
        if channels_g is None or channels_x is None or F_int is None:
            raise TypeError("Attention_block needs channels_g, channels_x and F_int")
        self.W_g = nn.Sequential(nn.Conv2d(channels_g, F_int, kernel_size=1, stride=1, padding=0, bias=True),
                                 nn.BatchNorm2d(F_int))
        self.W_x = nn.Sequential(nn.Conv2d(channels_x, F_int, kernel_size=1, stride=1, padding=0, bias=True),
                                 nn.BatchNorm2d(F_int))
        self.psi = nn.Sequential(nn.Conv2d(F_int, 1, kernel_size=1, stride=1, padding=0, bias=True),
                                 nn.BatchNorm2d(1), nn.Sigmoid())
        self.relu = nn.ReLU(inplace=True)

    def nhwc(self, g, x):
        dt = self.compute_dtype
        g1 = ops.conv_bn_act(dt, g, self.W_g[0], self.W_g[1])
        s = ops.conv_bn_act(dt, x, self.W_x[0], self.W_x[1], L.ACT_RELU, res=g1)
        p = ops.conv_bn_act(dt, s, self.psi[0], self.psi[1], L.ACT_SIGMOID)
        return ops.Gate.apply(dt, x, p)

    def forward(self, g, x):
        return self._out(self.nhwc(self._in(g), self._in(x)))


# ------------------------------------------------------------------------------------------------
# SD_Layer_Net/unet.py
# ------------------------------------------------------------------------------------------------
class _SDUNetBase(HipModule):
    _levels = 5

    def _encode(self, a):
        dt = self.compute_dtype
        feats = []
        for i in range(1, self._levels + 1):
            if i > 1:
                a = ops.MaxPool.apply(dt, 2, a)
            a = getattr(self, f"Conv{i}").nhwc(a)
            feats.append(a)
        return feats

    def forward(self, x):
        if x.dim() != 4:
            raise RuntimeError(f"expected a 4-D (B,C,H,W) input, got {tuple(x.shape)}")
        div = 1 << (self._levels - 1)
        if x.shape[2] % div or x.shape[3] % div:
            # the reference fails at torch.cat((x4, d5), dim=1) (unet.py:55,130) for such sizes
            raise RuntimeError(f"Sizes of tensors must match except in dimension 1. Input {x.shape[2]}x{x.shape[3]} "
                               f"is not divisible by {div}")
        ops.prepack(self.compute_dtype, self)
        feats = self._encode(self._in(x))
        d = feats[-1]
        for i in range(self._levels, 1, -1):
            skip = feats[i - 2]
            d = getattr(self, f"Up{i}").nhwc(d, lazy=True)
            att = getattr(self, f"Att{i}", None)
            if att is not None:
                skip = att.nhwc(d, skip)
            d = getattr(self, f"Up_conv{i}").nhwc(skip, d)      # torch.cat((skip, d), dim=1)
        return self._out(ops.conv_bn_act(self.compute_dtype, d, self.Conv_1x1))


class U_Net(_SDUNetBase):
    """SD_Layer_Net/unet.py:8-74.  As in the reference the head is Conv2d(64, output_ch): channels[0] must be 64."""

    def __init__(self, img_ch: int = 3, output_ch: int = 1, channels=[64, 128, 256, 512, 1024], act_func=None,
                 drop_rate=0.0, compute_dtype="bf16"):
        super().__init__()
        self.compute_dtype = compute_dtype
        if act_func is None:
            act_func = nn.ReLU
        kw = dict(act=act_func, drop_rate=drop_rate, compute_dtype=compute_dtype)
        self.Maxpool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.Conv1 = conv_block(ch_in=img_ch, ch_out=channels[0], **kw)
        for i in range(1, 5):
            setattr(self, f"Conv{i + 1}", conv_block(ch_in=channels[i - 1], ch_out=channels[i], **kw))
        for i in (5, 4, 3, 2):
            setattr(self, f"Up{i}", up_conv(ch_in=channels[i - 1], ch_out=channels[i - 2], **kw))
            setattr(self, f"Up_conv{i}", conv_block(ch_in=channels[i - 1], ch_out=channels[i - 2], **kw))
        self.Conv_1x1 = nn.Conv2d(64, output_ch, kernel_size=1, stride=1, padding=0)


class AttU_Net(_SDUNetBase):
    """SD_Layer_Net/unet.py:76-150 (BASELINE cfg4): U_Net with an additive attention gate on every skip."""

    def __init__(self, img_ch=1, output_ch=1, channels=[64, 128, 256, 512, 1024], act=None, drop_rate=0.0,
                 compute_dtype="bf16"):
        super().__init__()
        self.compute_dtype = compute_dtype
        if act is None:
            act = nn.ReLU
        kw = dict(act=act, drop_rate=drop_rate, compute_dtype=compute_dtype)
        self.Maxpool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.Conv1 = conv_block(ch_in=img_ch, ch_out=channels[0], **kw)
        for i in range(1, 5):
            setattr(self, f"Conv{i + 1}", conv_block(ch_in=channels[i - 1], ch_out=channels[i], **kw))
        for i in (5, 4, 3, 2):
            setattr(self, f"Up{i}", up_conv(ch_in=channels[i - 1], ch_out=channels[i - 2], **kw))
            setattr(self, f"Att{i}", Attention_block(F_g=channels[i - 2], F_l=channels[i - 2],
                                                     F_int=channels[i - 2] // 2, compute_dtype=compute_dtype))
            setattr(self, f"Up_conv{i}", conv_block(ch_in=channels[i - 1], ch_out=channels[i - 2], **kw))
        self.Conv_1x1 = nn.Conv2d(channels[0], output_ch, kernel_size=1, stride=1, padding=0)


class AttU_Net4(_SDUNetBase):
    """SD_Layer_Net/unet.py:153-214: the four-level attention-gated U-Net (three poolings, channels 64..512)."""
    _levels = 4

    def __init__(self, img_ch=1, output_ch=1, channels=[64, 128, 256, 512], act=None, drop_rate=0.0, compute_dtype="bf16"):
        super().__init__()
        self.compute_dtype = compute_dtype
        if act is None:
            act = nn.ReLU
        kw = dict(act=act, drop_rate=drop_rate, compute_dtype=compute_dtype)
        self.Maxpool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.Conv1 = conv_block(ch_in=img_ch, ch_out=channels[0], **kw)
        for i in range(1, 4):
            setattr(self, f"Conv{i + 1}", conv_block(ch_in=channels[i - 1], ch_out=channels[i], **kw))
        for i in (4, 3, 2):
            setattr(self, f"Up{i}", up_conv(ch_in=channels[i - 1], ch_out=channels[i - 2], **kw))
            setattr(self, f"Att{i}", Attention_block(F_g=channels[i - 2], F_l=channels[i - 2],
                                                     F_int=channels[i - 2] // 2, compute_dtype=compute_dtype))
            setattr(self, f"Up_conv{i}", conv_block(ch_in=channels[i - 1], ch_out=channels[i - 2], **kw))
        self.Conv_1x1 = nn.Conv2d(channels[0], output_ch, kernel_size=1, stride=1, padding=0)
