"""which gradients differ between OCT_LAZY settings (SD U_Net fixture, f32 and bf16), in module order"""
import os, sys
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.cases import bio_case
from retinal_oct_image_segmentation_via_deep_learning_amd import ops
from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.SD_Layer_Net import unet as U
z = np.load(os.path.join(ROOT, "tests", "golden", "sd_unet_c2_1x32x32.npz"))
seed, n, cin, ncls, h, w = (int(v) for v in z["meta"])
G = {}
D = {}
for setting in (True, "relu"):
    ops.LAZY[0] = setting
    ops.DEBUG = D[setting] = []
    m, x, t = bio_case(lambda ci, nc: U.U_Net(ci, nc, compute_dtype="f32"), seed, n, cin, ncls, h, w)
    m.cuda()
    xin = x.cuda().requires_grad_(True)
    loss = F.cross_entropy(m(xin), t.cuda()); loss.backward()
    G[setting] = {k: p.grad.double().cpu() for k, p in m.named_parameters()}
    G[setting]["x"] = xin.grad.double().cpu()
for k in G[True]:
    a, b = G[True][k], G["relu"][k]
    d = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)
    if d > 1e-4:
        print(f"{k:34s} rel diff {d:.2e}  max|ref| {float(b.abs().max()):.3e}")

for (ta, a), (tb, b) in zip(D[True], D["relu"]):
    line = []
    for k in a:
        if a[k] is None or b[k] is None:
            continue
        if a[k].shape != b[k].shape:
            line.append(f"{k}: shape"); continue
        d = float((a[k] - b[k]).abs().max()) / max(float(b[k].abs().max()), 1e-12)
        if d > 1e-5:
            line.append(f"{k}: {d:.1e}")
    print(ta, "|", tb.split(" ", 1)[1], "->", ", ".join(line) or "same")
print("---- dbeta of the lazy BN+ReLU layers against a torch evaluation of the captured tensors ----")
for setting in (True, "relu"):
    for tag, a in D[setting]:
        if "lazy=relu" not in tag:
            continue
        mask = (a["y"].double() * a["scale"].double() + a["shift"].double()) > 0
        ref = (a["dout"].double() * mask).sum((0, 1, 2))
        err = float((a["dbeta"].double() - ref).abs().max()) / max(float(ref.abs().max()), 1e-12)
        print(setting, tag, f"dbeta err {err:.2e}")
print("---- detail of the first differing lazy layer ----")
for (ta, a), (tb, b) in zip(D[True], D["relu"]):
    if "128x128x9@16x16 lazy=relu" in ta:
        for k in ("y", "dout", "scale", "shift", "mean", "invstd", "dy", "dbeta"):
            da_ = (a[k].double() - b[k].double()).abs()
            print(k, "max abs diff", float(da_.max()), "n differing", int((da_ > 0).sum()), "of", da_.numel(), "max|b|", float(b[k].abs().max()))
        ma = (a["y"].double() * a["scale"].double() + a["shift"].double()) > 0
        mb = (b["y"].double() * b["scale"].double() + b["shift"].double()) > 0
        print("mask flips", int((ma != mb).sum()))
        break
