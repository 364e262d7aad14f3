"""MGU-Net (SOTAS/Layers_Segment/MGUNet_2021.py:29-39,110-309): `Basconv`, `GloRe_Unit`, `MGR_Module`, `MGUNet`, `MGUNet_2`.

Same constructor arguments, sub-module names (state_dict keys) and seeded initialisation as the reference classes; the
torch.nn members are parameter containers.  Everything conv-shaped runs on the HIP ops layer (ops.py): 3x3 / 1x1 convolutions
with BatchNorm + ReLU, the 2x2 / 3x3 / 5x5 poolings of the multi-scale module (torch's floor mode), the bilinear resize back to
the bottleneck resolution (`oct_bilinear_resize_*`), transposed convolutions and the U-Net blocks of blocks.py.  The graph
reasoning step of `GloRe_Unit` -- two M x M matrix products per image with a row softmax between them (:135-140) -- is a
plain batched GEMM and goes to rocBLAS through `torch.bmm` in fp32 (M <= 256 nodes over <= 1/64 of the pixels: 0.1 % of the
network's FLOPs), and `torch.cat` of the four scale branches (:190) is a device copy.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .blocks import HipModule, UnetConv, UnetUp, UnetUp4, init_weights


class Basconv(HipModule):
    """MGUNet_2021.py:29-39: Conv2d(k, padding, bias) + BatchNorm2d + ReLU; k = 3 (pad 1) or 1 (pad 0)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, padding=1, compute_dtype="bf16"):
        super().__init__()
        self.compute_dtype = compute_dtype
        if (kernel_size, padding) not in ((3, 1), (1, 0)):
            raise NotImplementedError("Basconv: only kernel_size=3/padding=1 and kernel_size=1/padding=0 are on the HIP path")
        self.conv = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, padding=padding),
                                  nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))

    def nhwc(self, a):
        return ops.conv_bn_act(self.compute_dtype, a, self.conv[0], self.conv[1], L.ACT_RELU)

    def forward(self, x):
        return self._out(self.nhwc(self._in(x)))


class GloRe_Unit(HipModule):
    """MGUNet_2021.py:110-148: out = x + conv_extend(softmax(S P^T / sqrt(hw)) P) with S = conv_state(x), P = conv_proj(x)."""

    def __init__(self, in_channels, out_channels, kernel=1, compute_dtype="bf16"):
        super().__init__()
        self.compute_dtype = compute_dtype
        self.N = in_channels
        self.M = out_channels
        self.conv_state = nn.Conv2d(self.N, self.M, kernel_size=1)
        self.conv_proj = nn.Conv2d(self.N, self.M, kernel_size=1)
        self.conv_extend = nn.Conv2d(self.M, self.N, kernel_size=1)

    def nhwc(self, a):
        dt = self.compute_dtype
        n, h, w, _ = a.shape
        hw = h * w
        s = ops.conv_bn_act(dt, a, self.conv_state).reshape(n, hw, self.M)     # x_state, pixel-major
        p = ops.conv_bn_act(dt, a, self.conv_proj).reshape(n, hw, self.M)      # x_proj
        pf = p.float()
        adj = torch.bmm(s.float().transpose(1, 2), pf) / (hw ** 0.5)           # [n, M, M]  (:135-136)
        adj = torch.softmax(adj, dim=2)
        r = torch.bmm(pf, adj.transpose(1, 2)).to(a.dtype).reshape(n, h, w, self.M)   # x_rstate (:140), pixel-major
        return ops.conv_bn_act(dt, r, self.conv_extend, res=a)                 # x + conv_extend(.)  (:146)

    def forward(self, x):
        return self._out(self.nhwc(self._in(x)))


class MGR_Module(HipModule):
    """MGUNet_2021.py:150-194: four scale branches (full, /2, /3, /5 resolution), graph reasoning on each, bilinear resize
    back, concatenation, 1x1 fusion."""

    def __init__(self, in_channels, out_channels, compute_dtype="bf16"):
        super().__init__()
        self.compute_dtype = compute_dtype
        kw = dict(compute_dtype=compute_dtype)

        def glore(m):
            return nn.Sequential(OrderedDict([("GCN%02d" % i, GloRe_Unit(out_channels, m, kernel=1, **kw)) for i in range(1)]))

        self.conv0_1 = Basconv(in_channels=in_channels, out_channels=out_channels, kernel_size=3, padding=1, **kw)
        self.glou0 = glore(out_channels)
        self.conv1_1 = Basconv(in_channels=in_channels, out_channels=out_channels, kernel_size=3, padding=1, **kw)
        self.pool1 = nn.MaxPool2d(kernel_size=[2, 2], stride=2)
        self.conv1_2 = Basconv(in_channels=out_channels, out_channels=out_channels, kernel_size=3, padding=1, **kw)
        self.glou1 = glore(out_channels)
        self.conv2_1 = Basconv(in_channels=in_channels, out_channels=out_channels, kernel_size=3, padding=1, **kw)
        self.pool2 = nn.MaxPool2d(kernel_size=[3, 3], stride=3)
        self.conv2_2 = Basconv(in_channels=out_channels, out_channels=out_channels, kernel_size=3, padding=1, **kw)
        self.glou2 = glore(int(out_channels / 2))
        self.conv3_1 = Basconv(in_channels=in_channels, out_channels=out_channels, kernel_size=3, padding=1, **kw)
        self.pool3 = nn.MaxPool2d(kernel_size=[5, 5], stride=5)
        self.conv3_2 = Basconv(in_channels=out_channels, out_channels=out_channels, kernel_size=3, padding=1, **kw)
        self.glou3 = glore(int(out_channels / 2))
        self.f1 = Basconv(in_channels=4 * out_channels, out_channels=in_channels, kernel_size=1, padding=0, **kw)

    def nhwc(self, a):
        dt = self.compute_dtype
        n, h, w, _ = a.shape
        if h < 5 or w < 5:
            # the reference fails in the 5x5 pooling (:168) with torch's "Output size is too small"
            raise RuntimeError(f"Given input size: ({a.shape[3]}x{h}x{w}). Calculated output size: "
                               f"({a.shape[3]}x{h // 5}x{w // 5}). Output size is too small")
        branches = [self.glou0[0].nhwc(self.conv0_1.nhwc(a))]
        for k, c1, c2, g in ((2, self.conv1_1, self.conv1_2, self.glou1), (3, self.conv2_1, self.conv2_2, self.glou2),
                             (5, self.conv3_1, self.conv3_2, self.glou3)):
            t = g[0].nhwc(c2.nhwc(ops.MaxPool.apply(dt, k, c1.nhwc(a))))
            branches.append(ops.BilinearResize.apply(dt, (h, w), t))
        return self.f1.nhwc(torch.cat(branches, dim=3))

    def forward(self, x):
        return self._out(self.nhwc(self._in(x)))


class _MGUNetBase(HipModule):
    _pools = (2, 2, 2)

    def __init__(self, in_channels=1, num_classes=11, feature_scale=4, is_deconv=True, is_batchnorm=True, compute_dtype="bf16"):
        super().__init__()
        self.compute_dtype = compute_dtype
        self.is_deconv = is_deconv
        self.in_channels = in_channels
        self.is_batchnorm = is_batchnorm
        self.feature_scale = feature_scale
        kw = dict(compute_dtype=compute_dtype)
        filters = [int(x / self.feature_scale) for x in [64, 128, 256, 512, 1024]]
        p1, p2, p3 = self._pools
        ups = [UnetUp4 if p == 4 else UnetUp for p in (p3, p2, p1)]
        self.conv1 = UnetConv(self.in_channels, filters[0], self.is_batchnorm, **kw)
        self.maxpool1 = nn.MaxPool2d(kernel_size=p1)
        self.conv2 = UnetConv(filters[0], filters[1], self.is_batchnorm, **kw)
        self.maxpool2 = nn.MaxPool2d(kernel_size=p2)
        self.conv3 = UnetConv(filters[1], filters[2], self.is_batchnorm, **kw)
        self.maxpool3 = nn.MaxPool2d(kernel_size=p3)
        self.mgb = MGR_Module(filters[2], filters[3], **kw)
        self.center = UnetConv(filters[2], filters[3], self.is_batchnorm, **kw)
        self.up_concat3 = ups[0](filters[3], filters[2], self.is_deconv, **kw)
        self.up_concat2 = ups[1](filters[2], filters[1], self.is_deconv, **kw)
        self.up_concat1 = ups[2](filters[1], filters[0], self.is_deconv, **kw)
        self.final_1 = nn.Conv2d(filters[0], num_classes, 1)
        # MGUNet_2021.py:232-236 / :284-288: kaiming-normal convolutions, N(1, 0.02) BatchNorm weights -- nn.Conv2d and
        # nn.BatchNorm2d instances only (the transposed convolutions keep torch's default)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                init_weights(m, init_type="kaiming")
            elif isinstance(m, nn.BatchNorm2d):
                init_weights(m, init_type="kaiming")

    def forward(self, inputs):
        dt = self.compute_dtype
        ops.prepack(dt, self)
        p1, p2, p3 = self._pools
        conv1 = self.conv1.nhwc(self._in(inputs))
        conv2 = self.conv2.nhwc(ops.MaxPool.apply(dt, p1, conv1))
        conv3 = self.conv3.nhwc(ops.MaxPool.apply(dt, p2, conv2))
        feat_sum = self.mgb.nhwc(ops.MaxPool.apply(dt, p3, conv3))
        center = self.center.nhwc(feat_sum)
        up3 = self.up_concat3.nhwc(center, conv3)
        up2 = self.up_concat2.nhwc(up3, conv2)
        up1 = self.up_concat1.nhwc(up2, conv1)
        return self._out(ops.conv_bn_act(dt, up1, self.final_1))


class MGUNet(_MGUNetBase):
    """MGUNet_2021.py:197-252: poolings 2 / 4 / 4, up-sampling x4 / x4 / x2 (`UnetUp4`, `UnetUp`), logits out."""
    _pools = (2, 4, 4)


class MGUNet_2(_MGUNetBase):
    """MGUNet_2021.py:255-309: poolings 2 / 2 / 2, up-sampling x2 three times, logits out."""
    _pools = (2, 2, 2)
