"""Ad-hoc GPU probe: bf16-vs-f32 gradient agreement at a moderate size and a first timing."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from retinal_oct_image_segmentation_via_deep_learning_amd import UNet

def grads(dtype, B, H, W, f, ncls, seed=0):
    torch.manual_seed(seed)
    m = UNet(1, ncls, init_features=f, compute_dtype=dtype).cuda().train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 1, H, W, generator=g).cuda(); t = torch.randint(0, ncls, (B, H, W), generator=g).cuda()
    loss = m.forward_backward(x, t)
    torch.cuda.synchronize()
    return loss.cpu().numpy(), {k: p.grad.double().cpu().numpy().ravel() for k, p in m.named_parameters()}

if "cmp" in sys.argv:
    for (B, H, W, f) in [(2, 32, 32, 4), (4, 128, 128, 16), (2, 256, 256, 32)]:
        l32, g32 = grads("f32", B, H, W, f, 8)
        l16, g16 = grads("bf16", B, H, W, f, 8)
        cs = {k: float(g32[k] @ g16[k] / (np.linalg.norm(g32[k]) * np.linalg.norm(g16[k]) + 1e-30)) for k in g32}
        worst = sorted(cs.items(), key=lambda kv: kv[1])[:5]
        print(f"B{B} {H}x{W} f{f}: loss f32 {l32[0]:.5f} bf16 {l16[0]:.5f}  mean cos {np.mean(list(cs.values())):.4f} worst {worst}")

if "time" in sys.argv:
    B = int(sys.argv[sys.argv.index("time") + 1])
    torch.manual_seed(0)
    m = UNet(1, 8, init_features=32, compute_dtype="bf16").cuda().train()
    x = torch.randn(B, 1, 512, 1024).cuda(); t = torch.randint(0, 8, (B, 512, 1024)).cuda()
    for _ in range(2):
        m.forward_backward(x, t)
    torch.cuda.synchronize()
    t0 = time.time(); n = 3
    for _ in range(n):
        loss = m.forward_backward(x, t)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    print(f"B={B}: {dt*1e3:.1f} ms/step -> {B/dt:.1f} B-scans/s; loss {loss.cpu().numpy()}; mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
