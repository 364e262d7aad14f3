"""Conv / weight-gradient shapes of one MGUNet_2(1, 11) training step at 2 x 496 x 768 (which ones are not multiples of 32 channels).  usage (GPU box): python tools/probe_conv_shapes.py"""
import sys, collections
sys.path.insert(0, '/root/repo')
import torch
from retinal_oct_image_segmentation_via_deep_learning_amd import _lib as L, engine as E
from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.MGUNet_2021 import MGUNet_2
seen = collections.Counter()
oc, ow = E.UNetEngine._conv, E.UNetEngine._wgrad
def conv(self, src, wp, cout, taps, n, h, w, *a, **k):
    seen[("conv", taps, src.c0, src.c1, cout, h, w, k.get("in_mode", 0), k.get("out_mode", 0))] += 1
    return oc(self, src, wp, cout, taps, n, h, w, *a, **k)
def wgrad(self, src, dy, cout, taps, n, h, w, *a, **k):
    seen[("wgrad", taps, src.c0, src.c1, cout, h, w, k.get("dy_mode", 0), 0)] += 1
    return ow(self, src, dy, cout, taps, n, h, w, *a, **k)
E.UNetEngine._conv, E.UNetEngine._wgrad = conv, wgrad
m = MGUNet_2(1, 11).cuda().train()
x = torch.randn(2, 1, 496, 768, device="cuda"); t = torch.randint(0, 11, (2, 496, 768), device="cuda")
out = m(x); out = out[0] if isinstance(out, (tuple, list)) else out
torch.nn.functional.cross_entropy(out, t).backward()
for k, v in sorted(seen.items()):
    irregular = (k[2] % 32 or k[3] % 32 or (k[4] % 32 and not (k[8] == 1))) 
    print(v, k, "IRREGULAR" if irregular else "")
