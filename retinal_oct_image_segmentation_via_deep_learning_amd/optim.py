"""Fused SGD over ONE flat fp32 parameter buffer (torch.optim.SGD semantics: momentum,
dampening 0, no nesterov).  Parameters and their .grad become views into two flat buffers, so the
optimizer step is a single HIP kernel and the data-parallel exchange a few contiguous slices.

`FlatParams` is the layout alone (host logic, any device -- the gloo tests build it on CPU);
`FusedSGD.step` is the HIP kernel and refuses CPU tensors: there is no CPU fallback."""
from __future__ import annotations

import torch

from . import _lib as L

_ALIGN = 64  # floats


class FlatParams:
    """Re-homes parameters (and their gradients) as views into two flat fp32 buffers, in the order
    given -- `model.named_parameters()` order, i.e. the reference's construction order."""

    def __init__(self, named_params):
        named = [(n, p) for n, p in named_params]
        if not named:
            raise ValueError("no parameters")
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        dev = self.params[0].device
        self.offsets, total = [], 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise L.OctError("flat parameter buffer needs fp32 parameters")
            self.offsets.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.total = total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + n].view(p.shape)
            p.grad = self.flat_g[o:o + n].view(p.shape)

    def span(self, name: str):
        """[lo, hi) of one parameter inside the flat buffers (hi includes the alignment padding)."""
        i = self.names.index(name)
        hi = self.offsets[i + 1] if i + 1 < len(self.offsets) else self.total
        return self.offsets[i], hi


class FusedSGD:
    def __init__(self, params, lr: float, momentum: float = 0.0, weight_decay: float = 0.0):
        params = list(params)
        if params and isinstance(params[0], tuple):
            named = params
        else:
            named = [(f"p{i}", p) for i, p in enumerate(params)]
        if not named:
            raise ValueError("no parameters")
        if named[0][1].device.type != "cuda":
            raise L.OctError("FusedSGD needs device parameters (no CPU fallback)")
        self.layout = FlatParams(named)
        self.params = self.layout.params
        self.flat_p, self.flat_g = self.layout.flat_p, self.layout.flat_g
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        self.buf = torch.zeros_like(self.flat_p) if momentum != 0.0 else None
        self.steps = 0

    def zero_grad(self, set_to_none: bool = False):
        """Gradients are overwritten (not accumulated) by UNet.forward_backward; kept for API parity."""
        self.flat_g.zero_()

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        L.check(L.lib().oct_sgd_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), L.ptr(self.buf),
                                     self.flat_p.numel(), self.lr, self.momentum, self.weight_decay, grad_scale,
                                     1 if self.steps == 0 else 0, torch.cuda.current_stream().cuda_stream),
                "oct_sgd_step")
        self.steps += 1
        L.param_generation[0] += 1  # packed-weight caches must notice the raw-pointer update
