"""Fused SGD over ONE flat fp32 parameter buffer (torch.optim.SGD semantics: momentum,
dampening 0, no nesterov).  Parameters and their .grad become views into two flat buffers, so the
optimizer step is a single HIP kernel and the data-parallel all-reduce a single RCCL call."""
from __future__ import annotations

import torch

from . import _lib as L

_ALIGN = 64  # floats


class FusedSGD:
    def __init__(self, params, lr: float, momentum: float = 0.0, weight_decay: float = 0.0):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise L.OctError("FusedSGD needs device parameters (no CPU fallback)")
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.buf = torch.zeros(total, dtype=torch.float32, device=dev) if momentum != 0.0 else None
        for p, o in zip(self.params, offs):
            n = p.numel()
            self.flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + n].view(p.shape)
            p.grad = self.flat_g[o:o + n].view(p.shape)
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
