"""ReLayNet block family on the HIP ops layer -- mirror of SOTAS/Lesions_Segment/ReLayNet_2017.py:21-201.

  BasicBlock      :133-168  Conv2d(kernel (7,3), padding (3,1), bias) -> BatchNorm2d -> nn.PReLU() (one slope)
  EncoderBlock    :171-179  BasicBlock + MaxPool2d(2, 2, return_indices=True): returns (pooled, out_block, indices)
  DecoderBlock    :182-191  MaxUnpool2d(2, 2)(input, indices), cat((out_block, unpool), 1), BasicBlock
  ClassifierBlock :194-203  Conv2d(1x1, bias); the reference's Softmax2d is constructed but NOT applied -> logits
  ReLayNet        :21-126   3 encoders, bottleneck, 3 decoders, classifier

Same constructor arguments (the `params` dictionaries included), sub-module names (hence state_dict keys and
seeded default init) and forward signatures / return values as the reference.  The torch.nn members are
parameter containers; the arithmetic is liboct_hip.so: the 7x3 convolution runs on the generalised (kh, kw)
implicit-GEMM kernel (K = 21*Cin), PReLU / pool-with-indices / unpool on their own kernels (blocks.hip).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .blocks import HipModule


def _nchw_idx(idx_nhwc):
    return idx_nhwc.permute(0, 3, 1, 2).contiguous()


def _nhwc_idx(idx_nchw):
    if idx_nchw.dtype != torch.int64:
        raise RuntimeError(f"max_unpool2d: indices must be int64, got {idx_nchw.dtype}")
    return idx_nchw.permute(0, 2, 3, 1).contiguous()


class BasicBlock(HipModule):
    def __init__(self, params, compute_dtype="bf16"):
        super().__init__()
        self.compute_dtype = compute_dtype
        kh, kw = params["kernel_h"], params["kernel_w"]
        if (kh, kw) not in ((7, 3), (3, 3), (1, 1)) or params["stride_conv"] != 1:
            raise NotImplementedError("the HIP path runs the reference's 7x3 (and 3x3 / 1x1) stride-1 convolutions")
        self.conv = nn.Conv2d(in_channels=params["num_channels"], out_channels=params["num_filters"],
                              kernel_size=(kh, kw), padding=(int((kh - 1) / 2), int((kw - 1) / 2)),
                              stride=params["stride_conv"])
        self.batchnorm = nn.BatchNorm2d(num_features=params["num_filters"])
        self.prelu = nn.PReLU()

    def nhwc(self, a, a1=None):
        return ops.conv_bn_act(self.compute_dtype, a, self.conv, self.batchnorm, x1=a1, prelu=self.prelu)

    def forward(self, input):
        return self._out(self.nhwc(self._in(input)))


# OCT_POOL_CODES=0: the network's internal pool -> un-pool path on torch's int64 indices (the public block API always uses them)
import os as _os
_WINDOW_CODES = [_os.environ.get("OCT_POOL_CODES", "1") != "0"]


class EncoderBlock(BasicBlock):
    def __init__(self, params, compute_dtype="bf16"):
        super().__init__(params, compute_dtype)
        if params["pool"] != params["stride_pool"]:
            raise NotImplementedError("only non-overlapping pooling (stride == kernel, the reference's 2/2) is on the HIP path")
        self._k = params["pool"]
        self.maxpool = nn.MaxPool2d(kernel_size=params["pool"], stride=params["stride_pool"], return_indices=True)

    def nhwc(self, a, codes=False):
        """codes=True (inside ReLayNet.forward, where the indices only travel to the matching DecoderBlock): one-byte window codes
        instead of torch's int64 plane indices (ops.MaxPoolCode) -- the same winners, an eighth of the index traffic, and a dense
        un-pooling that needs no zero fill."""
        out_block = BasicBlock.nhwc(self, a)
        if codes:
            pooled, idx = ops.MaxPoolCode.apply(self.compute_dtype, self._k, out_block)
        else:
            pooled, idx = ops.MaxPoolIdx.apply(self.compute_dtype, self._k, out_block)
        return pooled, out_block, idx

    def forward(self, input):
        pooled, out_block, idx = self.nhwc(self._in(input))
        return self._out(pooled), self._out(out_block), _nchw_idx(idx)


class DecoderBlock(BasicBlock):
    def __init__(self, params, compute_dtype="bf16"):
        super().__init__(params, compute_dtype)
        if params["pool"] != params["stride_pool"]:
            raise NotImplementedError("only non-overlapping unpooling (stride == kernel) is on the HIP path")
        self._k = params["pool"]
        self.unpool = nn.MaxUnpool2d(kernel_size=params["pool"], stride=params["stride_pool"])

    def nhwc(self, a, out_block, idx):
        if idx.dtype == torch.uint8:     # window codes of an EncoderBlock.nhwc(codes=True)
            unpool = ops.MaxUnpoolCode.apply(self.compute_dtype, self._k, a, idx)
        else:
            unpool = ops.MaxUnpool.apply(self.compute_dtype, self._k, a, idx)
        if unpool.shape[:3] != out_block.shape[:3]:
            # torch.cat((out_block, unpool), dim=1), ReLayNet_2017.py:187
            raise RuntimeError(f"Sizes of tensors must match except in dimension 1. Expected size {out_block.shape[1]}x"
                               f"{out_block.shape[2]} but got size {unpool.shape[1]}x{unpool.shape[2]}")
        return BasicBlock.nhwc(self, out_block, unpool)        # skip first, unpooled second

    def forward(self, input, out_block, indices):
        return self._out(self.nhwc(self._in(input), self._in(out_block), _nhwc_idx(indices)))


class ClassifierBlock(HipModule):
    def __init__(self, params, compute_dtype="bf16"):
        super().__init__()
        self.compute_dtype = compute_dtype
        if params["kernel_c"] != 1 or params["stride_conv"] != 1:
            raise NotImplementedError("the classifier is the reference's 1x1 stride-1 convolution")
        self.conv = nn.Conv2d(params["num_channels"], params["num_class"], params["kernel_c"], params["stride_conv"])
        self.softmax = nn.Softmax2d()          # constructed and never applied, as in the reference (:201-203)

    def nhwc(self, a):
        return ops.conv_bn_act(self.compute_dtype, a, self.conv)

    def forward(self, input):
        return self._out(self.nhwc(self._in(input)))


class ReLayNet(HipModule):
    def __init__(self, in_channels=1, num_classes=10, num_filters=64, kernel_h=7, kernel_w=3, stride_conv=1, pool=2,
                 stride_pool=2, compute_dtype="bf16"):
        super().__init__()
        self.compute_dtype = compute_dtype
        base = {"num_channels": in_channels, "num_filters": num_filters, "kernel_h": kernel_h, "kernel_w": kernel_w,
                "stride_conv": stride_conv, "pool": pool, "stride_pool": stride_pool, "kernel_c": 1}
        wide = dict(base, num_channels=num_filters)
        cat = dict(base, num_channels=num_filters * 2, num_filters=num_filters)
        self.encode1 = EncoderBlock(base, compute_dtype)
        self.encode2 = EncoderBlock(wide, compute_dtype)
        self.encode3 = EncoderBlock(wide, compute_dtype)
        self.bottleneck = BasicBlock(wide, compute_dtype)
        self.decode1 = DecoderBlock(cat, compute_dtype)
        self.decode2 = DecoderBlock(cat, compute_dtype)
        self.decode3 = DecoderBlock(cat, compute_dtype)
        self.classifier = ClassifierBlock(dict(wide, num_class=num_classes), compute_dtype)
        self._div = pool ** 3

    def forward(self, input):
        if input.dim() != 4:
            raise RuntimeError(f"expected a 4-D (B,C,H,W) input, got {tuple(input.shape)}")
        if input.shape[2] % self._div or input.shape[3] % self._div:
            # the reference fails at torch.cat((out_block, unpool), dim=1) (ReLayNet_2017.py:187) for such sizes
            raise RuntimeError(f"Sizes of tensors must match except in dimension 1. Input {input.shape[2]}x"
                               f"{input.shape[3]} is not divisible by {self._div} (three {self._div ** (1 / 3):.0f}x poolings "
                               f"followed by as many unpoolings)")
        a = self._in(input)
        codes = _WINDOW_CODES[0]
        e1, out1, ind1 = self.encode1.nhwc(a, codes)
        e2, out2, ind2 = self.encode2.nhwc(e1, codes)
        e3, out3, ind3 = self.encode3.nhwc(e2, codes)
        bn = self.bottleneck.nhwc(e3)
        d3 = self.decode1.nhwc(bn, out3, ind3)
        d2 = self.decode2.nhwc(d3, out2, ind2)
        d1 = self.decode3.nhwc(d2, out1, ind1)
        return self._out(self.classifier.nhwc(d1))

    @property
    def is_cuda(self):
        return next(self.parameters()).is_cuda

    def save(self, path):
        print("Saving model... %s" % path)
        torch.save(self, path)
