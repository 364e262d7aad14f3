#!/usr/bin/env python3
"""Golden vectors for the reference's block families, made by IMPORTING
  /root/reference/SOTAS/Layers_Segment/MGUNet_2021.py          UnetConv/UnetUp/UnetUp4/init_weights (:42-108,314-352)
  /root/reference/SOTAS/Layers_Segment/SD_Layer_Net/{common,unet}.py   conv_block/up_conv/Attention_block, U_Net, AttU_Net
in the build container.  Nothing of their source is copied; the fixtures hold inputs and outputs.

AttU_Net cannot be constructed as shipped: unet.py:92 passes F_g/F_l to Attention_block, whose
parameters are called channels_g/channels_x (common.py:65) -> TypeError.  As SURVEY.md §8c
prescribes, this script (and only this script) wraps the reference's Attention_block so that both
spellings reach the reference's own constructor; no arithmetic is replaced.

Every module is run in float64 on fp32-representable weights/inputs.  Block fixtures store all
weights; the two networks store the seeded recipe + checksums (oracle/cases.py) like the BioNet
fixture, and for U_Net (34.5 M parameters) gradients as norm + sum + strided sample.
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle.cases import bio_case  # noqa: E402

sys.path.insert(0, os.path.join(REF, "SOTAS", "Layers_Segment"))
import MGUNet_2021 as ref_mg  # noqa: E402
import SD_Layer_Net.common as ref_common  # noqa: E402
import SD_Layer_Net.unet as ref_sd  # noqa: E402

_RefAtt = ref_common.Attention_block


def _att_both_spellings(channels_g=None, channels_x=None, F_int=None, F_g=None, F_l=None):
    return _RefAtt(channels_g if channels_g is not None else F_g, channels_x if channels_x is not None else F_l, F_int)


ref_sd.Attention_block = _att_both_spellings

STRIDE, FULL = 211, 4096

# Dropout2d(p > 0) under a FIXED mask: while a *_drop case runs, torch.nn.functional.dropout2d (what the reference's
# nn.Dropout2d modules call) is replaced by a function that applies the keep flags recorded in the fixture, in call order.
_MASKS = {"list": None, "i": 0}
_real_dropout2d = F.dropout2d


def _fixed_dropout2d(input, p=0.5, training=True, inplace=False):
    if _MASKS["list"] is None or not training or p == 0:
        return _real_dropout2d(input, p, training, inplace)
    m = _MASKS["list"][_MASKS["i"] % len(_MASKS["list"])]
    _MASKS["i"] += 1
    return input * (m.to(input.dtype) / (1.0 - p))[:, :, None, None]


def _rewind():
    _MASKS["i"] = 0


ACTS = (nn.ReLU, nn.LeakyReLU)   # activations whose kink at 0 the margin search keeps inputs away from


def _min_nonzero(t):
    """smallest non-zero magnitude (a channel dropped by Dropout2d reaches its activation as exact zeros: no rounding there)"""
    a = t.abs()
    a = a[a > 0]
    return float(a.min()) if a.numel() else 1e9


def relu_margin(m, *inputs):
    """smallest |pre-activation| over every ReLU of the module (a sign flip there is fp32 noise)"""
    zmin = [1e9]
    hooks = [mod.register_forward_pre_hook(lambda _m, i: zmin.__setitem__(0, min(zmin[0], _min_nonzero(i[0]))))
             for mod in m.modules() if isinstance(mod, ACTS)]
    _rewind()
    with torch.no_grad():
        m(*inputs)
    for h in hooks:
        h.remove()
    return zmin[0]


def block_case(name, make, in_shapes, seed, init=None, thresh=2e-5, masks=None):
    """make() -> module; inputs N(0,1); cotangent r ~ N(0,1); loss = sum(out * r).
    masks: [(n, c), ...] shapes of the Dropout2d keep masks of one forward, drawn from the case's seed with keep rate 0.75."""
    if masks:
        gm = torch.Generator().manual_seed(seed + 5000)
        _MASKS["list"] = [(torch.rand(s, generator=gm) < 0.75).float() for s in masks]
        F.dropout2d = _fixed_dropout2d
    try:
        _block_case(name, make, in_shapes, seed, init, thresh)
    finally:
        F.dropout2d = _real_dropout2d
        _MASKS["list"] = None


def _block_case(name, make, in_shapes, seed, init, thresh):
    while True:
        torch.manual_seed(seed)
        m = make().train()
        if init:
            ref_mg.init_weights(m, init) if not isinstance(m, (ref_mg.UnetConv, ref_mg.UnetUp, ref_mg.UnetUp4)) else \
                [ref_mg.init_weights(c, init) for c in m.modules() if isinstance(c, (nn.Conv2d, nn.BatchNorm2d, nn.ConvTranspose2d))]
        g = torch.Generator().manual_seed(seed + 1000)
        with torch.no_grad():
            for p in m.parameters():
                if p.dim() == 1:
                    p.add_(0.2 * torch.randn(p.shape, generator=g))
        xs = [torch.randn(s, generator=g) for s in in_shapes]
        state0 = {k: v.clone() for k, v in m.state_dict().items()}
        z = relu_margin(m, *xs)
        if z > thresh:
            break
        seed += 1
    m.load_state_dict(state0)          # the probe advanced the BN buffers
    md = m.double()
    xd = [x.double().requires_grad_(True) for x in xs]
    _rewind()
    out = md(*xd)
    r = torch.randn(out.shape, generator=g)
    (out * r.double()).sum().backward()
    rec = {"seed": np.array(seed), "r": r.numpy(), "out": out.detach().numpy()}
    if _MASKS["list"] is not None:
        assert _MASKS["i"] == len(_MASKS["list"]), "one mask per Dropout2d application"
        for j, mk in enumerate(_MASKS["list"]):
            rec[f"mask{j}"] = mk.numpy()
    for i, x in enumerate(xs):
        rec[f"x{i}"] = x.numpy()
        rec[f"gx{i}"] = xd[i].grad.numpy()
    for k, v in state0.items():
        rec["w0/" + k] = v.numpy()
    for k, p in md.named_parameters():
        rec["g/" + k] = p.grad.numpy()
    for k, v in md.state_dict().items():
        if "running" in k or "num_batches" in k:
            rec["b1/" + k] = v.numpy()
    md.eval()
    with torch.no_grad():
        rec["out_eval"] = md(*[x.double() for x in xs]).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: seed {seed} min|relu in| {z:.2e} out {tuple(out.shape)}")


def net_case(name, cls, seed, n, cin, ncls, h, w, thresh, full_weights, compact=False):
    """compact: logits as sum / abs-sum / every 97th element / arg-max map instead of the full float64 tensors."""
    while True:
        m, x, t = bio_case(cls, seed, n, cin, ncls, h, w)
        z = relu_margin(m, x)
        with torch.no_grad():
            lg = m(x)
        top2 = lg.sort(1).values[:, -2:]
        margin = float((top2[:, 1] - top2[:, 0]).min()) if ncls > 1 else 1.0
        if z > thresh and margin > 2e-5:
            break
        seed += 1
    m, x, t = bio_case(cls, seed, n, cin, ncls, h, w)
    rec = {"meta": np.array([seed, n, cin, ncls, h, w]), "x": x.numpy(), "target": t.numpy(),
           "keys": np.array(list(m.state_dict().keys()))}
    for k, v in m.state_dict().items():
        rec["wsum/" + k] = np.array([float(v.double().sum()), float(v.double().abs().sum())])
        if full_weights:
            rec["w0/" + k] = v.numpy()
    m = m.double()
    logits = m(x.double())
    loss = F.cross_entropy(logits, t)
    loss.backward()
    def put(key, v):
        if compact:
            rec[key + "_sample"] = v.reshape(-1)[::97].copy()
            rec[key + "_sums"] = np.array([v.sum(), np.abs(v).sum()])
            rec[key + "_argmax"] = v.argmax(1).astype(np.uint8)
        else:
            rec[key] = v
    put("logits", logits.detach().numpy())
    rec["loss"] = np.array([loss.item()])
    for k, p in m.named_parameters():
        gr = p.grad.numpy()
        if gr.size <= FULL:
            rec["g/" + k] = gr
        else:
            rec["gs/" + k] = gr.reshape(-1)[::STRIDE].copy()
            rec["gn/" + k] = np.array([np.sqrt((gr ** 2).sum()), gr.sum()])
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            rec["b1/" + k] = v.numpy()
    m.eval()
    with torch.no_grad():
        put("logits_eval", m(x.double()).numpy())
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: seed {seed} loss {loss.item():.6f} min|relu in| {z:.2e} margin {margin:.2e} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def init_case():
    """Seeded init_weights: checksums of every tensor after each init_type (MGUNet_2021.py:314-352)."""
    rec = {}
    for kind in ("normal", "xavier", "kaiming"):
        torch.manual_seed(77)
        m = ref_mg.UnetUp(16, 8, True)
        for c in m.modules():           # the reference applies it leaf by leaf (MGUNet_2021.py:232-236)
            if isinstance(c, (nn.Conv2d, nn.BatchNorm2d)):
                ref_mg.init_weights(c, init_type=kind)
        for k, v in m.state_dict().items():
            rec[f"{kind}/{k}"] = np.array([float(v.double().sum()), float(v.double().abs().sum())])
    try:
        ref_mg.init_weights(nn.Conv2d(1, 1, 1), "nope")
        rec["bad_type_msg"] = np.array("")
    except NotImplementedError as e:
        rec["bad_type_msg"] = np.array(str(e))
    np.savez_compressed(os.path.join(OUT, "mgunet_init.npz"), **rec)
    print("mgunet_init:", len(rec), "entries;", rec["bad_type_msg"])


def widened_cases():
    """act != nn.ReLU and Dropout2d(drop_rate > 0) of conv_block / up_conv (common.py:7,13,17,29,34)"""
    block_case("blk_conv_block_drop", lambda: ref_common.conv_block(3, 8, drop_rate=0.2), [(3, 3, 16, 24)], 600,
               masks=[(3, 8), (3, 8)])
    block_case("blk_up_conv_drop", lambda: ref_common.up_conv(8, 4, drop_rate=0.2), [(3, 8, 8, 12)], 610, masks=[(3, 4)])
    block_case("blk_conv_block_leaky", lambda: ref_common.conv_block(3, 8, act=nn.LeakyReLU), [(2, 3, 16, 24)], 620)
    block_case("blk_up_conv_tanh_drop", lambda: ref_common.up_conv(8, 4, act=nn.Tanh, drop_rate=0.2), [(2, 8, 8, 12)], 630,
               masks=[(2, 4)])


def main():
    torch.set_num_threads(8)
    if "--widened" in sys.argv:      # only the cases added in round 3 (the other fixtures stay byte-identical)
        return widened_cases()
    block_case("blk_unetconv_bn", lambda: ref_mg.UnetConv(3, 8, True), [(2, 3, 16, 24)], 300)
    block_case("blk_unetconv_nobn", lambda: ref_mg.UnetConv(1, 8, False), [(2, 1, 16, 24)], 310)
    block_case("blk_unetup_deconv", lambda: ref_mg.UnetUp(16, 8, True), [(2, 16, 8, 12), (2, 8, 16, 24)], 320)
    block_case("blk_unetup_bilinear", lambda: ref_mg.UnetUp(16, 8, False), [(2, 16, 8, 12), (2, 8, 16, 24)], 330)
    block_case("blk_unetup4_deconv", lambda: ref_mg.UnetUp4(16, 8, True), [(1, 16, 4, 6), (1, 8, 16, 24)], 340)
    block_case("blk_unetup4_bilinear", lambda: ref_mg.UnetUp4(16, 8, False), [(1, 16, 4, 6), (1, 8, 16, 24)], 350)
    block_case("blk_conv_block", lambda: ref_common.conv_block(3, 8), [(2, 3, 16, 24)], 360)
    block_case("blk_up_conv", lambda: ref_common.up_conv(8, 4), [(2, 8, 8, 12)], 370)
    block_case("blk_attention", lambda: ref_common.Attention_block(8, 8, 4), [(2, 8, 16, 24), (2, 8, 16, 24)], 380)
    widened_cases()
    init_case()
    net_case("attunet_c3_2x32x48", lambda ci, nc: ref_sd.AttU_Net(ci, nc, channels=[4, 8, 16, 32, 64]), 400,
             2, 1, 3, 32, 48, thresh=1e-5, full_weights=True)
    net_case("attunet4_c3_2x24x40", lambda ci, nc: ref_sd.AttU_Net4(ci, nc, channels=[4, 8, 16, 32]), 450,
             2, 1, 3, 24, 40, thresh=1e-5, full_weights=True)
    net_case("sd_unet_c2_1x32x32", lambda ci, nc: ref_sd.U_Net(ci, nc), 500, 1, 1, 2, 32, 32, thresh=5e-6,
             full_weights=False)
    # negative: AttU_Net as shipped raises TypeError at construction (unet.py:92)
    ref_sd.Attention_block = _RefAtt
    try:
        ref_sd.AttU_Net(1, 3, channels=[4, 8, 16, 32, 64])
        msg = ""
    except TypeError as e:
        msg = str(e)
    ref_sd.Attention_block = _att_both_spellings
    try:
        ref_sd.AttU_Net(1, 3, channels=[4, 8, 16, 32, 64])(torch.zeros(1, 1, 24, 32))
        neg = ""
    except RuntimeError as e:
        neg = str(e)
    np.savez_compressed(os.path.join(OUT, "sd_api.npz"), attunet_ctor_msg=np.array(msg), negative_msg=np.array(neg),
                        attunet_default_params=np.array(sum(p.numel() for p in ref_sd.AttU_Net(1, 3).parameters())),
                        unet_default_params=np.array(sum(p.numel() for p in ref_sd.U_Net(1, 2).parameters())))
    print("sd_api:", msg[:70], "|", neg[:70])


if __name__ == "__main__":
    main()
