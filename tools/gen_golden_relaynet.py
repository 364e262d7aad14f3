#!/usr/bin/env python3
"""Golden vectors for the ReLayNet block family, made by IMPORTING
  /root/reference/SOTAS/Lesions_Segment/ReLayNet_2017.py   ReLayNet (:21-126), BasicBlock / EncoderBlock /
                                                            DecoderBlock / ClassifierBlock (:133-203)
in the build container.  Nothing of its source is copied; the fixtures hold inputs and outputs.

Every module is run in float64 on fp32-representable weights / inputs (arithmetic of the reference's own torch
ops).  Seeds are advanced until no BatchNorm output sits within 2e-5 of the PReLU kink and no pooling window has
its two largest entries closer than 1e-5 (a tie there is rounding noise that re-routes a gradient).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_relaynet.py
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, "/root/reference/SOTAS/Lesions_Segment")
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import ReLayNet_2017 as ref  # noqa: E402

BASE = {"num_channels": 3, "num_filters": 8, "kernel_h": 7, "kernel_w": 3, "stride_conv": 1, "pool": 2,
        "stride_pool": 2, "kernel_c": 1}


def margins(m, *inputs):
    """(min |PReLU input|, min top-2 gap over all pooling windows)"""
    zmin, gap = [1e9], [1e9]

    def pre_prelu(_m, i):
        zmin[0] = min(zmin[0], float(i[0].abs().min()))

    def pre_pool(_m, i):
        x = i[0]
        b, c, h, w = x.shape
        v = x.reshape(b, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(b, c, h // 2, w // 2, 4)
        s = v.sort(-1).values
        gap[0] = min(gap[0], float((s[..., 3] - s[..., 2]).min()))
    hooks = [mod.register_forward_pre_hook(pre_prelu) for mod in m.modules() if isinstance(mod, nn.PReLU)]
    hooks += [mod.register_forward_pre_hook(pre_pool) for mod in m.modules() if isinstance(mod, nn.MaxPool2d)]
    with torch.no_grad():
        m(*inputs)
    for h in hooks:
        h.remove()
    return zmin[0], gap[0]


def perturb(m, g):
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if p.dim() == 1 and p.numel() > 1:
                p.add_(0.2 * torch.randn(p.shape, generator=g))
            elif p.numel() == 1:                       # PReLU slope off its 0.25 default
                p.add_(0.1 * torch.randn(p.shape, generator=g))


def block_case(name, make, in_shapes, seed, int_inputs=()):
    while True:
        torch.manual_seed(seed)
        m = make().train()
        g = torch.Generator().manual_seed(seed + 1000)
        perturb(m, g)
        xs = [torch.randn(s, generator=g) for s in in_shapes]
        state0 = {k: v.clone() for k, v in m.state_dict().items()}
        extra = []
        if int_inputs:      # DecoderBlock: indices come from a real pooling of a random tensor of the skip's shape
            src = torch.randn(in_shapes[1], generator=g)
            _, idx = F.max_pool2d(src, 2, 2, return_indices=True)
            extra = [idx]
        z, gp = margins(m, *xs, *extra)
        if z > 2e-5 and gp > 1e-5:
            break
        seed += 1
    m.load_state_dict(state0)
    md = m.double()
    xd = [x.double().requires_grad_(True) for x in xs]
    out = md(*xd, *extra)
    outs = out if isinstance(out, tuple) else (out,)
    rec = {"seed": np.array(seed), "n_out": np.array(len(outs))}
    loss = 0
    for i, o in enumerate(outs):
        rec[f"out{i}"] = o.detach().numpy()
        if o.dtype.is_floating_point:
            r = torch.randn(o.shape, generator=g)
            rec[f"r{i}"] = r.numpy()
            loss = loss + (o * r.double()).sum()
    loss.backward()
    for i, x in enumerate(xs):
        rec[f"x{i}"] = x.numpy()
        rec[f"gx{i}"] = xd[i].grad.numpy()
    for i, e in enumerate(extra):
        rec[f"idx{i}"] = e.numpy()
    for k, v in state0.items():
        rec["w0/" + k] = v.numpy()
    for k, p in md.named_parameters():
        rec["g/" + k] = p.grad.numpy()
    for k, v in md.state_dict().items():
        if "running" in k or "num_batches" in k:
            rec["b1/" + k] = v.numpy()
    md.eval()
    with torch.no_grad():
        oe = md(*[x.double() for x in xs], *extra)
        rec["out_eval"] = (oe[0] if isinstance(oe, tuple) else oe).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: seed {seed} min|prelu in| {z:.2e} pool gap {gp:.2e} outs {[tuple(o.shape) for o in outs]}")


def net_case(name, seed, n, cin, ncls, nf, h, w):
    while True:
        torch.manual_seed(seed)
        m = ref.ReLayNet(in_channels=cin, num_classes=ncls, num_filters=nf).train()
        g = torch.Generator().manual_seed(seed + 1000)
        perturb(m, g)
        x = torch.randn(n, cin, h, w, generator=g)
        t = torch.randint(0, ncls, (n, h, w), generator=g)
        state0 = {k: v.clone() for k, v in m.state_dict().items()}
        z, gp = margins(m, x)
        with torch.no_grad():
            lg = m(x)
        top2 = lg.sort(1).values[:, -2:]
        margin = float((top2[:, 1] - top2[:, 0]).min())
        if z > 1e-5 and gp > 1e-5 and margin > 2e-5:
            break
        seed += 1
    m.load_state_dict(state0)
    rec = {"meta": np.array([seed, n, cin, ncls, nf, h, w]), "x": x.numpy(), "target": t.numpy(),
           "keys": np.array(list(state0.keys()))}
    for k, v in state0.items():
        rec["w0/" + k] = v.numpy()
    m = m.double()
    logits = m(x.double())
    loss = F.cross_entropy(logits, t)
    loss.backward()
    rec["logits"] = logits.detach().numpy()
    rec["argmax"] = logits.detach().argmax(1).numpy()
    rec["loss"] = np.array([loss.item()])
    for k, p in m.named_parameters():
        rec["g/" + k] = p.grad.numpy()
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            rec["b1/" + k] = v.numpy()
    m.eval()
    with torch.no_grad():
        rec["logits_eval"] = m(x.double()).numpy()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: seed {seed} loss {loss.item():.6f} min|prelu in| {z:.2e} pool gap {gp:.2e} margin {margin:.2e} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    torch.set_num_threads(8)
    block_case("relay_basic", lambda: ref.BasicBlock(dict(BASE)), [(2, 3, 16, 24)], 600)
    block_case("relay_encoder", lambda: ref.EncoderBlock(dict(BASE)), [(2, 3, 16, 24)], 610)
    block_case("relay_decoder", lambda: ref.DecoderBlock(dict(BASE, num_channels=16)), [(2, 8, 8, 12), (2, 8, 16, 24)], 620,
               int_inputs=(2,))
    block_case("relay_classifier", lambda: ref.ClassifierBlock(dict(BASE, num_channels=8, num_class=5)), [(2, 8, 16, 24)], 630)
    net_case("relaynet_c4_f8_2x32x48", 700, 2, 1, 4, 8, 32, 48)
    net_case("relaynet_in3_c9_f16_1x16x40", 720, 1, 3, 9, 16, 16, 40)
    # API facts: default construction, parameter count, the reference's behaviour for a size not divisible by 8
    m = ref.ReLayNet()
    rec = {"default_params": np.array(sum(p.numel() for p in m.parameters())), "keys": np.array(list(m.state_dict().keys()))}
    try:
        ref.ReLayNet(1, 4, num_filters=4)(torch.zeros(1, 1, 20, 24))
        rec["negative_msg"] = np.array("")
    except RuntimeError as e:
        rec["negative_msg"] = np.array(str(e))
    np.savez_compressed(os.path.join(OUT, "relaynet_api.npz"), **rec)
    print("relaynet_api:", int(rec["default_params"]), "params;", str(rec["negative_msg"])[:90])


if __name__ == "__main__":
    main()
