#!/usr/bin/env python3
"""Golden vectors for the `UNet` of /root/reference/SOTAS/Layers_Segment/BioNet_2020.py:24-75,
made by IMPORTING that file in the build container.

The file's first lines import torchvision (used only by BioRegularization/BioNet, :77-130, which
are outside the hot path and are never instantiated here).  torchvision is not installed, so an
EMPTY module object is registered under that name for the duration of the import: it supplies no
function, class or arithmetic -- the `UNet` class under test is plain torch.nn.

The network has 7.7 M parameters (31 MB), too large for a fixture, so the fixture holds
  * the seeded recipe (oracle/cases.bio_case) + float64 checksums of every state tensor the
    reference class ended up with, so a test can prove its rebuilt weights are the same,
  * input, labels, train-mode logits, loss terms, eval-mode logits after the step,
  * every gradient: in full when it has <= 4096 elements, otherwise its L2 norm, sum and a
    strided sample (every 211th element),
  * BatchNorm running statistics after the step.
The reference module is run in float64 (`.double()` on fp32-representable weights) so the
vectors carry no fp32 rounding of their own.
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle.cases import bio_case  # noqa: E402

_tv = types.ModuleType("torchvision")
_tv.models = types.ModuleType("torchvision.models")
sys.modules["torchvision"], sys.modules["torchvision.models"] = _tv, _tv.models
sys.path.insert(0, os.path.join(REF, "SOTAS", "Layers_Segment"))
import BioNet_2020 as ref_bio  # noqa: E402

DICE_EPS = 1e-7
STRIDE = 211
FULL = 4096


def loss_fn(logits, target, ncls, w_ce, w_dice):
    ce = F.cross_entropy(logits, target)
    p = torch.softmax(logits, 1)
    oh = F.one_hot(target, ncls).permute(0, 3, 1, 2).to(p.dtype)
    dice = 1.0 - ((2 * (p * oh).sum((0, 2, 3)) + DICE_EPS) / (p.sum((0, 2, 3)) + oh.sum((0, 2, 3)) + DICE_EPS)).mean()
    return w_ce * ce + w_dice * dice, ce, dice


def case(name, seed, n, cin, ncls, h, w, w_ce, w_dice):
    while True:
        m, x, t = bio_case(ref_bio.UNet, seed, n, cin, ncls, h, w)
        zmin = [1e9]
        hooks = [mod.register_forward_hook(lambda _m, _i, o: zmin.__setitem__(0, min(zmin[0], float(o.abs().min()))))
                 for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm2d)]
        with torch.no_grad():
            lg = m(x)
        for hk in hooks:
            hk.remove()
        top2 = lg.sort(1).values[:, -2:]
        margin = float((top2[:, 1] - top2[:, 0]).min())
        if zmin[0] > 2e-5 and margin > 2e-5:
            break
        print(f"{name}: seed {seed} ill-conditioned (min|z|={zmin[0]:.2e}, margin={margin:.2e}); next")
        seed += 1
    m, x, t = bio_case(ref_bio.UNet, seed, n, cin, ncls, h, w)   # fresh: the probe advanced the BN buffers
    out = {"meta": np.array([seed, n, cin, ncls, h, w]), "hyper": np.array([w_ce, w_dice, DICE_EPS]),
           "x": x.numpy(), "target": t.numpy(), "keys": np.array(list(m.state_dict().keys()))}
    for k, v in m.state_dict().items():
        v = v.double()
        out["wsum/" + k] = np.array([float(v.sum()), float(v.abs().sum())])
    m = m.double()
    logits = m(x.double())
    loss, ce, dice = loss_fn(logits, t, ncls, w_ce, w_dice)
    loss.backward()
    out["logits"] = logits.detach().numpy()
    out["loss"] = np.array([loss.item(), ce.item(), dice.item()])
    for k, p in m.named_parameters():
        gr = p.grad.detach().numpy()
        if gr.size <= FULL:
            out["g/" + k] = gr
        else:
            out["gs/" + k] = gr.reshape(-1)[::STRIDE].copy()
            out["gn/" + k] = np.array([np.sqrt((gr ** 2).sum()), gr.sum()])
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            out["b1/" + k] = v.numpy()
    m.eval()
    with torch.no_grad():
        out["logits_eval"] = m(x.double()).numpy()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: seed {seed} loss {loss.item():.6f} min|z| {zmin[0]:.2e} margin {margin:.2e} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    torch.set_num_threads(8)
    case("bionet_unet_c2_2x16x24", 21, 2, 1, 2, 16, 24, 1.0, 0.5)
    case("bionet_unet_in3_c4_1x32x16", 22, 1, 3, 4, 32, 16, 1.0, 0.0)
    # negative: 3 poolings need H, W divisible by 8 -> torch.cat raises (BioNet_2020.py:64)
    try:
        ref_bio.UNet(1, 2)(torch.zeros(1, 1, 20, 16))
        msg = ""
    except RuntimeError as e:
        msg = str(e)
    print("negative 20x16:", msg[:80])
    np.savez_compressed(os.path.join(OUT, "bionet_api.npz"), negative_msg=np.array(msg),
                        n_params=np.array(sum(p.numel() for p in ref_bio.UNet(1, 2).parameters())))


if __name__ == "__main__":
    main()
