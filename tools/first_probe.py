"""first-layer kernels (Conv2d / Conv3d 1 -> 32): time of fprop and wgrad at the cfg2 / cfg5 shapes (HIP events, 10 runs)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from retinal_oct_image_segmentation_via_deep_learning_amd import UNet
from retinal_oct_image_segmentation_via_deep_learning_amd.unet3d import UNet3D


def timed(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("both", "2d"):
    m = UNet(1, 8, init_features=32, compute_dtype="bf16").cuda().train()
    x = torch.randn(32, 1, 512, 1024, device="cuda")
    t = torch.randint(0, 8, (32, 512, 1024), device="cuda")
    print("cfg2 step ms", timed(lambda: m.forward_backward(x, t), 10))
if which in ("both", "3d"):
    m3 = UNet3D(1, 4, init_features=32, compute_dtype="bf16").cuda().train()
    x3 = torch.randn(4, 1, 64, 512, 512, device="cuda")
    t3 = torch.randint(0, 4, (4, 64, 512, 512), device="cuda")
    print("cfg5 step ms", timed(lambda: m3.forward_backward(x3, t3), 3))
