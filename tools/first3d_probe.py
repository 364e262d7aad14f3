"""time of the Conv3d(1 -> 32) first-layer forward at the cfg5 shape (HIP events) -- run with OCT_HIP_LIB for A/B builds"""
import os, sys, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from retinal_oct_image_segmentation_via_deep_learning_amd import _lib as L
from retinal_oct_image_segmentation_via_deep_learning_amd.unet3d import UNet3D
m = UNet3D(1, 4, init_features=32, compute_dtype="bf16").cuda().train()
e = m._engine
x = torch.randn(4 * 64, 512, 512, 1, device="cuda").to(torch.bfloat16)
w = m.state_dict()["encoder1.enc1conv1.weight"] if "encoder1.enc1conv1.weight" in m.state_dict() else next(iter(m.parameters()))
print(w.shape)
from retinal_oct_image_segmentation_via_deep_learning_amd.engine import Src
wp = e._pack("probe", w, L.PACK_CONV3D_FPROP, 32, 1)
y = torch.empty(4 * 64, 512, 512, 32, device="cuda", dtype=torch.bfloat16)
nblk = e._stat_blocks(32, 256, 512, 512, Src(x, 1), 9, depth=64)
st = torch.empty(nblk, 2, 32, device="cuda")
def run():
    e._conv(Src(x, 1), wp, 32, 9, 256, 512, 512, y, stats=st, depth=64)
for _ in range(2): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
print("first_fprop3d ms", e0.elapsed_time(e1) / 5)
