"""U-Net with the reference's constructor API and state_dict, executed by the HIP engine.

Mirrors `UNet` / `get_model` of SOTAS/Lesions_Segment/YNet_2022.py:496-602 (the class in
SOTAS/Layers_Segment/YNet_2022:33-139 is byte-identical): same argument names and defaults, same
118 state_dict keys and shapes, same default initialisation (the torch.nn modules below are built
in the reference's order, so a seeded construction yields the same weights), same output
(softmax probabilities, NCHW), same failure for inputs not divisible by 16 and for unknown model
names.  The torch.nn modules are parameter containers only: `forward` never calls them -- it runs
the hand-written gfx950 kernels through `engine.UNetEngine`.  There is no CPU fallback.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib as L
from .engine import UNetEngine, bionet_unet_spec


def _block(cin: int, cout: int, name: str) -> nn.Sequential:
    layers = OrderedDict()
    for i, ci in ((1, cin), (2, cout)):
        layers[f"{name}conv{i}"] = nn.Conv2d(ci, cout, kernel_size=3, padding=1, bias=False)
        layers[f"{name}norm{i}"] = nn.BatchNorm2d(cout)
        layers[f"{name}relu{i}"] = nn.ReLU(inplace=True)
    return nn.Sequential(layers)


class _UNetFn(torch.autograd.Function):
    """One autograd node for the whole network: forward saves the raw conv outputs, backward runs
    the complete HIP backward schedule from d(loss)/d(probabilities)."""

    @staticmethod
    def forward(ctx, model, x, *params):
        # eval() mode under autograd: BatchNorm on its running statistics, differentiated as the per-channel affine it then is
        # (frozen-BN fine-tuning, as the reference nn.Module allows -- YNet_2022.py:548-569)
        out = model._run(x, train=model.training, keep_ctx=ctx, frozen_bwd=not model.training)
        ctx.model = model
        return out

    @staticmethod
    def backward(ctx, dout):
        model = ctx.model
        if ctx.ectx is None:
            raise RuntimeError("Trying to backward through the HIP U-Net a second time: the saved conv outputs are "
                               "freed by the first backward (retain_graph is not supported; run forward again)")
        names = [n for n, _ in model.named_parameters()]
        P = model._tensors()
        G = {n: torch.empty_like(P[n]) for n in names}
        if model._engine.spec.softmax_out:
            model._engine.backward(P, ctx.ectx, G, dprobs=dout)
        else:
            model._engine.backward(P, ctx.ectx, G, dlogits=dout)
        ctx.ectx = None
        return (None, None) + tuple(G[n] for n in names)


class _EngineNet(nn.Module):
    """Shared host logic of the engine-backed networks; subclasses build the parameter containers
    (in the reference's construction order) and set `self._engine`."""

    def _run(self, x, train: bool, keep_ctx=None, frozen_bwd=False):
        """The reference forward's return value: probabilities (YNet_2022 UNet) or logits (BioNet UNet)."""
        soft = self._engine.spec.softmax_out
        kw = {"frozen_bwd": True} if frozen_bwd else {}
        ectx, probs, _, lg = self._engine.forward(self._tensors(), x, train=train, want_probs=soft,
                                                  want_logits=not soft, **kw)
        if keep_ctx is not None:
            keep_ctx.ectx = ectx
        return probs if soft else lg

    # ---- configuration --------------------------------------------------------------------------
    def set_compute_dtype(self, dtype: str):
        """'bf16' (production) or 'f32' (parity mode: same kernels, fp32 storage, exact fp32 MFMA)."""
        self._engine.set_dtype(dtype)
        return self

    @property
    def compute_dtype(self) -> str:
        return self._engine.dtype

    def _tensors(self) -> dict:
        t = dict(self.named_parameters())  # Parameters themselves: data_ptr + live _version
        t.update({n: b for n, b in self.named_buffers()})
        for n, v in t.items():
            if v.is_floating_point() and (v.dtype != torch.float32 or not v.is_contiguous()):
                raise L.OctError(f"parameter {n} must be contiguous fp32 (got {v.dtype})")
        return t

    # ---- reference API ---------------------------------------------------------------------------
    def forward(self, x):
        grad_on = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if torch.is_grad_enabled() and x.requires_grad:
            # the reference nn.Module returns d(out)/d(x); the engine stops at the first layer's weights
            raise NotImplementedError("gradients with respect to the network INPUT are not implemented on the HIP "
                                      "path (x.requires_grad=True); detach the input")
        if grad_on and (self.training or self._engine.supports_frozen_bwd):
            return _UNetFn.apply(self, x, *self.parameters())
        return self._run(x, train=self.training)

    # ---- fused extras (not in the reference) --------------------------------------
    @torch.no_grad()
    def forward_backward(self, x, target, w_ce=1.0, w_dice=0.0, dice_eps=1e-7, want_probs=False, stage_hook=None):
        """Training step without the optimizer: forward, fused CE(+Dice) loss head, full backward.
        Writes `.grad` of every parameter and returns the device tensor [loss, ce, dice]
        (plus the probabilities when want_probs).  With w_dice == 0 (and the fused head: 32 head
        features, <= 8 classes, no probabilities requested) the Dice sums are not accumulated and the
        third entry is 0: the cross-entropy then comes out of the backward head pass and the forward
        one is skipped.  stage_hook: see UNetEngine.backward (data-parallel gradient buckets).
        A target outside [0, classes) -- where torch's nll_loss raises -- yields a NaN loss (device-side
        flag, no synchronisation)."""
        if not self.training:
            raise RuntimeError("forward_backward needs train() mode (batch statistics)")
        P = self._tensors()
        ectx, probs, _, _ = self._engine.forward(P, x, train=True, target=target,
                                                 loss_cfg=(w_ce, w_dice, dice_eps), want_probs=want_probs,
                                                 defer_loss=True)
        G = {}
        for n, p in self.named_parameters():
            if p.grad is None:
                p.grad = torch.empty_like(p.data)
            G[n] = p.grad
        self._engine.backward(P, ectx, G, stage_hook=stage_hook)
        return (ectx.loss, probs) if want_probs else ectx.loss

    @torch.no_grad()
    def loss(self, x, target, w_ce=1.0, w_dice=0.0, dice_eps=1e-7):
        """[loss, ce, dice] of the current mode's forward pass (no gradients)."""
        ectx, _, _, _ = self._engine.forward(self._tensors(), x, train=self.training, target=target,
                                             loss_cfg=(w_ce, w_dice, dice_eps), want_probs=False)
        return ectx.loss

    @torch.no_grad()
    def predict(self, x):
        """Class map argmax_c p (int64, B x H x W); first maximum wins, like torch.argmax."""
        _, _, amax, _ = self._engine.forward(self._tensors(), x, train=self.training, want_probs=False,
                                             want_argmax=True)
        return amax

    @torch.no_grad()
    def logits(self, x):
        _, _, _, lg = self._engine.forward(self._tensors(), x, train=self.training, want_probs=False,
                                           want_logits=True)
        return lg


class UNet(_EngineNet):
    """SOTAS/{Lesions,Layers}_Segment/YNet_2022 `UNet` (reference :509-602)."""

    def __init__(self, in_channels=3, out_channels=1, init_features=32, compute_dtype="bf16"):
        super().__init__()
        f = init_features
        self.encoder1 = _block(in_channels, f, "enc1")
        self.pool1 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.encoder2 = _block(f, f * 2, "enc2")
        self.pool2 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.encoder3 = _block(f * 2, f * 4, "enc3")
        self.pool3 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.encoder4 = _block(f * 4, f * 8, "enc4")
        self.pool4 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.bottleneck = _block(f * 8, f * 16, "bottleneck")
        for k, mult in ((4, 8), (3, 4), (2, 2), (1, 1)):
            setattr(self, f"upconv{k}", nn.ConvTranspose2d(f * mult * 2, f * mult, kernel_size=2, stride=2))
            setattr(self, f"decoder{k}", _block(f * mult * 2, f * mult, f"dec{k}"))
        self.conv = nn.Conv2d(f, out_channels, kernel_size=1)
        self.softmax = nn.Softmax2d()
        self._engine = UNetEngine(in_channels, out_channels, f, compute_dtype)


def _bias_block(cin: int, cout: int) -> nn.Sequential:
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
                         nn.Conv2d(cout, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class BioUNet(_EngineNet):
    """`UNet` of SOTAS/Layers_Segment/BioNet_2020.py:24-75 (BASELINE cfg1 is `UNet(1, 2)`): widths
    64..512, three poolings, bias convolutions, `cat([enc, dec])`, raw logits out.  Same
    constructor arguments, construction order (seeded init matches) and 106 state_dict keys."""

    def __init__(self, in_channels, out_channels, compute_dtype="bf16"):
        super().__init__()
        self.enc1 = _bias_block(in_channels, 64)
        self.enc2 = _bias_block(64, 128)
        self.enc3 = _bias_block(128, 256)
        self.enc4 = _bias_block(256, 512)
        self.up4 = nn.ConvTranspose2d(512, 256, kernel_size=2, stride=2)
        self.dec4 = _bias_block(512, 256)
        self.up3 = nn.ConvTranspose2d(256, 128, kernel_size=2, stride=2)
        self.dec3 = _bias_block(256, 128)
        self.up2 = nn.ConvTranspose2d(128, 64, kernel_size=2, stride=2)
        self.dec2 = _bias_block(128, 64)
        self.final = nn.Conv2d(64, out_channels, kernel_size=1)
        self.maxpool = nn.MaxPool2d(2)
        self._engine = UNetEngine(in_channels, out_channels, dtype=compute_dtype,
                                  spec=bionet_unet_spec(in_channels, out_channels))

    def conv_block(self, in_ch, out_ch):
        return _bias_block(in_ch, out_ch)


def get_model(model_name, in_channels=1, num_classes=9, ratio=0.5):
    """Reference factory (YNet_2022.py:496-507).  Only "unet" is on the accelerated path; the
    Fourier Y-Net variants are out of scope and say so."""
    if model_name == "unet":
        return UNet(in_channels, num_classes)
    if model_name in ("y_net_gen", "y_net_gen_ffc"):
        raise NotImplementedError(f"{model_name}: the FFC Y-Net is outside the MI355X hot path (SURVEY.md §8)")
    print("Model name not found")
    assert False
