"""(diagnostic, not collected by pytest) A/B of the deferred-activation schedule on the SD_Layer_Net fixtures (f32 parity mode): per-tensor gradient errors against
the reference fixture with OCT_LAZY-style settings.  Usage (GPU box): python tools/lazy_probe.py"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.cases import bio_case  # noqa: E402
from retinal_oct_image_segmentation_via_deep_learning_amd import ops  # noqa: E402
from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.SD_Layer_Net import unet as U  # noqa: E402

for name, cls, kw in [("sd_unet_c2_1x32x32", "U_Net", {}), ("attunet_c3_2x32x48", "AttU_Net", dict(channels=[4, 8, 16, 32, 64]))]:
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    seed, n, cin, ncls, h, w = (int(v) for v in z["meta"])
    for setting in (True, "relu", False):
        ops.LAZY[0] = setting
        m, x, t = bio_case(lambda ci, nc: getattr(U, cls)(ci, nc, compute_dtype="f32", **kw), seed, n, cin, ncls, h, w)
        m.cuda()
        out = m(x.cuda())
        loss = F.cross_entropy(out, t.cuda())
        loss.backward()
        worst = []
        grads = {k: p.grad.cpu().numpy().astype(np.float64) for k, p in m.named_parameters()}
        for key in z.files:
            kind, _, k = key.partition("/")
            if kind not in ("g", "gs", "gn") or k.endswith((".0.bias", ".3.bias")):
                continue
            g = grads[k]
            ref, got = (z[key], g) if kind == "g" else ((z[key], g.reshape(-1)[::211]) if kind == "gs" else
                                                         (z[key][:1], np.array([np.sqrt((g ** 2).sum())])))
            worst.append((float(np.abs(got - ref).max()) / max(float(np.abs(ref).max()), 1e-4), key))
        worst.sort(reverse=True)
        print(name, "lazy =", setting, "loss", float(loss), "ref", float(z["loss"][0]), "worst rel-of-max:",
              [(f"{e:.2e}", k) for e, k in worst[:4]], "median", f"{np.median([e for e, _ in worst]):.2e}")
