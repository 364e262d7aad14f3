"""Timing of the transposed-convolution weight gradients of the cfg2 step (wgrad2 taps = 1, dY gathered
space-to-depth).  usage: deconv_wgrad_probe.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from retinal_oct_image_segmentation_via_deep_learning_amd import _lib as L, engine as E
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
eng = E.UNetEngine(1, 8, 32, "bf16")
bf = torch.bfloat16
tot = 0.0
for (h, w, cin, cout) in [(32, 64, 512, 256), (64, 128, 256, 128), (128, 256, 128, 64), (256, 512, 64, 32)]:
    x = torch.randn(B, h, w, cin, device="cuda").to(bf)
    bn = E.BNState(torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1)
    du = torch.randn(B, 2 * h, 2 * w, cout, device="cuda").to(bf)
    db = torch.zeros(cout, device="cuda")
    run = lambda: eng._wgrad(E.Src(x, cin, bn), du, 4 * cout, 1, B, h, w, dy_mode=L.IN_S2D, dbias=db)
    for _ in range(2):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    tot += ms
    nbytes = (x.numel() + du.numel()) * 2
    print(f"upconv {cin}->{cout} @{h}x{w}: {ms:.3f} ms  {nbytes / ms / 1e6:.0f} GB/s")
print(f"total {tot:.3f} ms")
