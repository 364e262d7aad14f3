"""Every MFMA launch of one cfg2 training step (HIP events on the launch stream) against its own floor:
max(FLOPs / 2.5 PFLOP/s, minimal bytes / 6 TB/s).  usage: launch_table.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from retinal_oct_image_segmentation_via_deep_learning_amd import UNet, _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
m = UNet(1, 8).cuda().train()
x = torch.randn(B, 1, 512, 1024, device="cuda")
t = torch.randint(0, 8, (B, 512, 1024), device="cuda")
for _ in range(3):
    m.forward_backward(x, t, 1.0, 0.0)
torch.cuda.synchronize()
e = m._engine
rows = {}
for rep in range(4):
    e.prof, e.prof_labels = [], []
    m.forward_backward(x, t, 1.0, 0.0)
    torch.cuda.synchronize()
    for i, ((kind, s, en), lab) in enumerate(zip(e.prof, e.prof_labels)):
        rows.setdefault(i, [kind, lab, []])[2].append(s.elapsed_time(en))
e.prof = e.prof_labels = None
tot = {}
print(f"{'#':>3s} {'kind':5s} {'taps':>4s} {'cin':>5s} {'cout':>5s} {'h x w':>10s} {'mode':>5s} {'ms':>7s} {'floor':>7s} {'x':>5s} {'TF/s':>6s} {'GB/s':>6s}")
for i in sorted(rows):
    kind, (taps, cin, cout, n, h, w, im, om), ms = rows[i]
    ms = sorted(ms)[len(ms) // 2]
    px = n * h * w
    flops = 2.0 * px * taps * cin * cout
    nbytes = px * (cin + cout) * 2
    floor = max(flops / 2.5e15, nbytes / 6e12) * 1e3
    key = (kind, taps)
    a = tot.setdefault(key, [0.0, 0.0]); a[0] += ms; a[1] += floor
    print(f"{i:3d} {kind:5s} {taps:4d} {cin:5d} {cout:5d} {h:4d}x{w:<5d} {im}/{om:<3d} {ms:7.3f} {floor:7.3f} {ms/floor:5.1f} {flops/ms/1e9:6.0f} {nbytes/ms/1e6:6.0f}")
for k, (ms, fl) in sorted(tot.items()):
    print(f"total {k}: {ms:.2f} ms, floor {fl:.2f} ms")
