#!/usr/bin/env python3
"""Generate golden input/output vectors by IMPORTING the reference (CPU, fp32).

Runs only in the build container, where /root/reference is mounted read-only.
Nothing of the reference's source is copied: this script imports
  /root/reference/SOTAS/Lesions_Segment/YNet_2022.py   (UNet :509-602, get_model :496-507)
  /root/reference/Metrics/Region_based_metrics.py       (:3-61)
  /root/reference/Metrics/ConfusionMatrix_based_metrics.py (:4-63)
  /root/reference/Metrics/PixelError_based_metrics.py (:3-37), Biomarker_based_metrics.py (:3-38)
feeds them seeded inputs and writes inputs + outputs as small .npz fixtures under
tests/golden/.  The loss head, optimizer and DDP do not exist in the reference
(SURVEY.md §0); for those the fixture records what stock torch (the reference's own
arithmetic provider) computes on the reference module: nll_loss(log(p)) (+ soft Dice),
autograd gradients and torch.optim.SGD steps.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

sys.path.insert(0, os.path.join(REF, "SOTAS", "Lesions_Segment"))
sys.path.insert(0, os.path.join(REF, "Metrics"))
import YNet_2022 as ref_ynet  # noqa: E402
import Region_based_metrics as ref_region  # noqa: E402
import ConfusionMatrix_based_metrics as ref_cm  # noqa: E402
import PixelError_based_metrics as ref_px  # noqa: E402
import Biomarker_based_metrics as ref_bio  # noqa: E402

DICE_EPS = 1e-7


def loss_fn(probs, target, num_classes, w_ce, w_dice):
    """Loss head definition used by the build, in stock torch."""
    logp = torch.log(probs)
    ce = F.nll_loss(logp, target)
    onehot = F.one_hot(target, num_classes).permute(0, 3, 1, 2).to(probs.dtype)
    inter = (probs * onehot).sum((0, 2, 3))
    psum = probs.sum((0, 2, 3))
    ysum = onehot.sum((0, 2, 3))
    dice = 1.0 - ((2.0 * inter + DICE_EPS) / (psum + ysum + DICE_EPS)).mean()
    return w_ce * ce + w_dice * dice, ce, dice


def well_conditioned(in_ch, n_cls, feat, shape, seed):
    """A fixture is only a fair parity target if no ReLU pre-activation, pooling pair or
    arg-max margin sits within fp32 rounding of a tie: a sign flip there is legitimate fp32
    noise but changes gradients by O(1%).  Returns (min |BN output|, min top-2 margin)."""
    torch.manual_seed(seed)
    model = ref_ynet.UNet(in_ch, n_cls, init_features=feat)
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.copy_(1.0 + 0.3 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.2 * torch.randn(mod.bias.shape, generator=g))
    B, H, W = shape
    x = torch.randn(B, in_ch, H, W, generator=g)
    mins = []
    hooks = [m.register_forward_hook(lambda m, i, o: mins.append(o.detach().abs().min().item()))
             for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    model.train()
    with torch.no_grad():
        p = model(x)
    for h in hooks:
        h.remove()
    top2 = p.topk(min(2, n_cls), dim=1).values
    return min(mins), (top2[:, 0] - top2[:, -1]).min().item()


def unet_case(name, in_ch, n_cls, feat, shape, seed, w_ce=1.0, w_dice=0.0, steps=3,
              lr=0.05, momentum=0.9, light=False):
    while True:
        mz, mm = well_conditioned(in_ch, n_cls, feat, shape, seed)
        if mz > 2e-5 and mm > 2e-5:
            break
        print(f"  {name}: seed {seed} rejected (min|z|={mz:.2e}, margin={mm:.2e})")
        seed += 100
    print(f"  {name}: seed {seed} accepted (min|z|={mz:.2e}, margin={mm:.2e})")
    torch.manual_seed(seed)
    model = ref_ynet.UNet(in_ch, n_cls, init_features=feat)
    # non-trivial BN affine parameters so that gamma/beta paths are exercised
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.copy_(1.0 + 0.3 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.2 * torch.randn(mod.bias.shape, generator=g))
    B, H, W = shape
    x = torch.randn(B, in_ch, H, W, generator=g)
    target = torch.randint(0, n_cls, (B, H, W), generator=g)
    out = {"x": x.numpy(), "target": target.numpy(),
           "meta": np.array([in_ch, n_cls, feat, B, H, W, steps], dtype=np.int64),
           "hyper": np.array([w_ce, w_dice, lr, momentum, DICE_EPS], dtype=np.float64),
           "seed": np.array(seed), "min_abs_preact": np.array(mz)}
    for k, v in model.state_dict().items():
        out["w0/" + k] = v.detach().numpy().copy()

    # ---- train-mode forward, logits via a hook on the 1x1 head ---------------------------
    logits_box = {}
    hook = model.conv.register_forward_hook(lambda m, i, o: logits_box.__setitem__("v", o.detach()))
    model.train()
    probs = model(x)
    loss, ce, dice = loss_fn(probs, target, n_cls, w_ce, w_dice)
    loss.backward()
    hook.remove()
    logits = logits_box["v"]
    out["logits"] = logits.numpy()
    out["probs"] = probs.detach().numpy()
    out["argmax"] = probs.detach().argmax(1).numpy()
    top2 = probs.detach().topk(min(2, n_cls), dim=1).values
    out["min_margin"] = np.array((top2[:, 0] - top2[:, -1]).min().item())
    out["loss"] = np.array([loss.item(), ce.item(), dice.item()], dtype=np.float64)
    for k, p in model.named_parameters():
        if not light or "conv1.weight" in k or k.startswith("conv.") or "upconv4" in k:
            out["g0/" + k] = p.grad.detach().numpy().copy()
    for k, v in model.state_dict().items():
        if "running" in k or "num_batches" in k:
            out["b1/" + k] = v.detach().numpy().copy()

    # ---- eval-mode forward with the buffers after that one training forward -------------
    model.eval()
    with torch.no_grad():
        pe = model(x)
    out["probs_eval"] = pe.numpy()
    out["argmax_eval"] = pe.argmax(1).numpy()

    # ---- SGD trajectory: `steps` optimizer steps on the same batch ----------------------
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=momentum)
    # the backward above already produced the grads of step 0
    losses = [loss.item()]
    opt.step()
    for _ in range(steps - 1):
        opt.zero_grad()
        l2, _, _ = loss_fn(model(x), target, n_cls, w_ce, w_dice)
        l2.backward()
        losses.append(l2.item())
        opt.step()
    out["traj_loss"] = np.array(losses, dtype=np.float64)
    for k, v in model.state_dict().items():
        if not light or "norm" in k or k.startswith("conv."):
            out["wN/" + k] = v.detach().numpy().copy()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: loss={losses} min_margin={float(out['min_margin']):.3e} -> {path} "
          f"({os.path.getsize(path)/1024:.0f} KiB)")


def negative_case():
    """62x96 input is not divisible by 16 -> torch.cat raises (YNet_2022.py:557)."""
    model = ref_ynet.UNet(1, 2, init_features=4)
    try:
        model(torch.zeros(1, 1, 62, 96))
        msg = ""
    except RuntimeError as e:  # noqa: PERF203
        msg = str(e)
    assert "Sizes of tensors must match" in msg
    return msg


def get_model_case():
    m = ref_ynet.get_model("unet", in_channels=1, num_classes=9)
    n_params = sum(p.numel() for p in m.parameters())
    keys = list(m.state_dict().keys())
    shapes = [list(v.shape) for v in m.state_dict().values()]
    try:
        ref_ynet.get_model("nope")
        raised = False
    except AssertionError:
        raised = True
    return n_params, keys, shapes, raised


def metrics_cases():
    out = {}
    fns = {
        "region.dice_coefficient": ref_region.dice_coefficient,
        "region.iou_score": ref_region.iou_score,
        "region.precision": ref_region.precision,
        "region.recall": ref_region.recall,
        "cm.accuracy": ref_cm.accuracy,
        "cm.sensitivity": ref_cm.sensitivity,
        "cm.precision": ref_cm.precision,
        "cm.specificity": ref_cm.specificity,
        "pixel.mean_squared_error": ref_px.mean_squared_error,          # PixelError_based_metrics.py:3-19
        "pixel.root_mean_squared_error": ref_px.root_mean_squared_error,  # :21-37
        "bio.thickness_difference": ref_bio.thickness_difference,      # Biomarker_based_metrics.py:3-21
        "bio.vascularity_index": ref_bio.vascularity_index,            # :23-38
    }
    cases = {}
    cases["tiny_i64"] = (np.array([[1, 1, 0, 0], [1, 0, 0, 0]], dtype=np.int64),
                         np.array([[1, 0, 1, 0], [1, 0, 0, 1]], dtype=np.int64))
    cases["zeros_4x4"] = (np.zeros((4, 4), dtype=np.int64), np.zeros((4, 4), dtype=np.int64))
    cases["ones_3x5"] = (np.ones((3, 5), dtype=np.uint8), np.ones((3, 5), dtype=np.uint8))
    rng = np.random.default_rng(7)
    a = (rng.random((3, 37, 53)) < 0.4)
    b = (rng.random((3, 37, 53)) < 0.6)
    cases["ragged_bool"] = (a, b)
    cases["ragged_u8"] = (a.astype(np.uint8), b.astype(np.uint8))
    cases["ragged_i32"] = (a.astype(np.int32), b.astype(np.int32))
    cases["ragged_f32"] = (a.astype(np.float32), b.astype(np.float32))
    cases["ragged_f64"] = (a.astype(np.float64), b.astype(np.float64))
    cases["single_px"] = (np.array([1], dtype=np.uint8), np.array([0], dtype=np.uint8))
    for cname, (yt, yp) in cases.items():
        out[f"in/{cname}/y_true"] = yt
        out[f"in/{cname}/y_pred"] = yp
        for fname, fn in fns.items():
            out[f"out/{cname}/{fname}"] = np.array(fn(yt, yp), dtype=np.float64)
    # the seeded full-size case of SURVEY App. B: only the scalar answers are stored
    rng = np.random.default_rng(1234)
    a = (rng.random((32, 512, 1024)) < 0.3).astype(np.uint8)
    b = (rng.random((32, 512, 1024)) < 0.3).astype(np.uint8)
    for fname, fn in fns.items():
        out[f"out/seeded_32x512x1024/{fname}"] = np.array(fn(a, b), dtype=np.float64)
    out["seeded_counts"] = np.array(
        [np.sum(a.astype(np.int64) * b), np.sum(a, dtype=np.int64), np.sum(b, dtype=np.int64), a.size],
        dtype=np.int64)
    path = os.path.join(OUT, "metrics.npz")
    np.savez_compressed(path, **out)
    print("metrics ->", path, f"({os.path.getsize(path)/1024:.0f} KiB)")
    for k in sorted(out):
        if k.startswith("out/tiny_i64") or k.startswith("out/seeded"):
            print("  ", k, repr(float(out[k])))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    unet_case("unet_c8_f4_2x32x32", 1, 8, 4, (2, 32, 32), seed=0)
    unet_case("unet_c2_f4_1x48x64_dice", 1, 2, 4, (1, 48, 64), seed=1, w_ce=1.0, w_dice=0.5)
    unet_case("unet_in3_c3_f4_2x32x48", 3, 3, 4, (2, 32, 48), seed=2, w_ce=0.7, w_dice=0.3)
    unet_case("unet_c8_f8_1x32x64_light", 1, 8, 8, (1, 32, 64), seed=5, light=True)
    msg = negative_case()
    n_params, keys, shapes, raised = get_model_case()
    np.savez_compressed(os.path.join(OUT, "api.npz"),
                        negative_msg=np.array(msg), n_params=np.array(n_params),
                        keys=np.array(keys), shapes=np.array([str(s) for s in shapes]),
                        unknown_raises_assert=np.array(raised))
    print("api: n_params", n_params, "keys", len(keys), "unknown->AssertionError", raised)
    metrics_cases()


if __name__ == "__main__":
    main()
