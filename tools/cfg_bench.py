"""Throughput of the parity configurations (not the bench line): cfg1 BioNet UNet(1,2) 4x256x256 and
cfg4 AttU_Net(1,3) 16x496x768, bf16, training step = fwd + CE + bwd + SGD.  usage: cfg_bench.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.BioNet_2020 import UNet as BioUNet
from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.SD_Layer_Net.unet import AttU_Net
from retinal_oct_image_segmentation_via_deep_learning_amd.optim import FusedSGD

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = torch.Generator().manual_seed(1234)


def run(name, model, x, t, fused):
    model.cuda().train()
    opt = FusedSGD(model.parameters(), lr=0.01, momentum=0.9) if fused else torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)

    def step():
        if fused:
            model.forward_backward(x, t)
        else:
            opt.zero_grad(set_to_none=True)
            F.cross_entropy(model(x), t).backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name}: {dt * 1e3:.2f} ms/step, {x.shape[0] / dt:.1f} B-scans/s, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


ONLY = os.environ.get("CFG_ONLY", "")
torch.manual_seed(0)
x = torch.randn(4, 1, 256, 256, generator=g).cuda(); t = torch.randint(0, 2, (4, 256, 256), generator=g).cuda()
if ONLY in ("", "1"):
    run("cfg1 BioNet UNet(1,2) 4x256x256", BioUNet(1, 2), x, t, True)
x = torch.randn(32, 1, 256, 256, generator=g).cuda(); t = torch.randint(0, 2, (32, 256, 256), generator=g).cuda()
if ONLY in ("", "1"):
    run("     BioNet UNet(1,2) 32x256x256", BioUNet(1, 2), x, t, True)
x = torch.randn(16, 1, 496, 768, generator=g).cuda(); t = torch.randint(0, 3, (16, 496, 768), generator=g).cuda()
if ONLY in ("", "4"):
    run("cfg4 AttU_Net(1,3) 16x496x768", AttU_Net(1, 3), x, t, False)
