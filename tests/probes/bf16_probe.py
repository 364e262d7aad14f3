"""Layer-by-layer differences between the bf16 HIP path and the rounding-aware oracle (diagnostic, GPU box):
for every conv layer the fraction of stored y / g / dy elements that differ and the largest difference in
bf16 ulps, against the oracle accumulating in float64 and in float32."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_cpu as R  # noqa: E402
from retinal_oct_image_segmentation_via_deep_learning_amd import UNet  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "unet_c2_f4_1x48x64_dice"
z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
in_ch, n_cls, feat = (int(v) for v in z["meta"][:3])
w0 = {k[3:]: z[k] for k in z.files if k.startswith("w0/")}
w_ce, w_dice, lr, mom, eps = (float(v) for v in z["hyper"])
model = UNet(in_ch, n_cls, init_features=feat, compute_dtype="bf16")
model.load_state_dict({k: torch.from_numpy(v) for k, v in w0.items()})
model.cuda().train()
model._engine.debug = {}
model.forward_backward(torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["target"]).cuda(), w_ce, w_dice, eps)
torch.cuda.synchronize()
dbg = {k: v.cpu().numpy() for k, v in model._engine.debug.items()}
hg = {k: p.grad.cpu().numpy() for k, p in model.named_parameters()}
for dt in (np.float64, np.float32):
    net = R.OracleUNet(w0, dtype=dt, storage="bf16")
    net.trace = {}
    _, _, g = net.loss_and_grads(z["x"], z["target"], w_ce, w_dice, eps)
    print(f"--- oracle accumulating in {np.dtype(dt).name}")
    for k in sorted(net.trace, key=lambda s: list(net.trace).index(s)):
        if k not in dbg:
            continue
        a = dbg[k].transpose(0, 3, 1, 2).astype(np.float64)      # NHWC -> NCHW
        b = np.asarray(net.trace[k], np.float64)
        d = a != b
        ulp = np.abs(a - b) / np.maximum(np.abs(b), 1e-30) * 256
        print(f"{k:46s} differ {d.mean():9.2e}  max ulp {ulp[d].max() if d.any() else 0:8.2f}  n={d.size}")
    worst = sorted(((np.linalg.norm(hg[k] - g[k]) / (np.linalg.norm(g[k]) + 1e-30)), k) for k in g)[::-1]
    print("worst grads:", [(round(float(a), 4), k) for a, k in worst[:5]])
