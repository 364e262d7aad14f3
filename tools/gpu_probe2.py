import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from retinal_oct_image_segmentation_via_deep_learning_amd import UNet

def run(dtype, B, H, W, f, ncls):
    torch.manual_seed(0)
    m = UNet(1, ncls, init_features=f, compute_dtype=dtype).cuda().train()
    m._engine.debug = {}
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 1, H, W, generator=g).cuda(); t = torch.randint(0, ncls, (B, H, W), generator=g).cuda()
    m.forward_backward(x, t)
    torch.cuda.synchronize()
    return m._engine.debug

def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))

B, H, W, f = 4, 128, 128, 16
d32 = run("f32", B, H, W, f, 8)
d16 = run("bf16", B, H, W, f, 8)
for k in d32:
    if k.startswith("y:"):
        w = k[2:]
        print(f"{w:40s} y {rel(d16[k], d32[k]):.4f}  g {rel(d16['g:'+w], d32['g:'+w]):.4f}  dy {rel(d16['dy:'+w], d32['dy:'+w]):.4f}  coef {rel(d16['coef:'+w], d32['coef:'+w]):.4f}")
