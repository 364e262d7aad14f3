"""Graph-replayed steps must produce the same parameters as eager steps (same seeds, 5 steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from retinal_oct_image_segmentation_via_deep_learning_amd import UNet, ddp
res = []
for use_graph in (False, True):
    torch.manual_seed(0)
    model = UNet(1, 8, init_features=32).cuda().train()
    tr = ddp.DataParallelTrainer(model, lr=0.05, momentum=0.9, use_graph=use_graph, graph_warmup=2)
    g = torch.Generator().manual_seed(1)
    losses = []
    for i in range(6):
        x = torch.randn(2, 1, 128, 256, generator=g).cuda()
        t = torch.randint(0, 8, (2, 128, 256), generator=g).cuda()
        losses.append(float(tr.step(x, t)[0]))
    torch.cuda.synchronize()
    res.append((losses, tr.opt.flat_p.clone(), tr.graph is not None, tr.graph_error))
    print("graph" if use_graph else "eager", [round(l, 5) for l in losses], res[-1][2], res[-1][3])
d = (res[0][1] - res[1][1]).abs().max().item()
print("max |param diff| eager vs graph:", d, " rel:", d / res[0][1].abs().max().item())
