"""Per-layer timing of the cfg2 conv stack (fprop / dgrad / wgrad) against each launch's own floor:
max(flops / 2.5 PFLOP/s, minimal bytes / 6 TB/s).  usage: layer_sweep.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from retinal_oct_image_segmentation_via_deep_learning_amd import _lib as L, engine as E

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, W, F = 512, 1024, 32
eng = E.UNetEngine(1, 8, F, "bf16")
dev, bf = "cuda", torch.bfloat16
# (name, level, c0, c1, cout, xform0, xform1)
layers = []
for lv in range(5):
    c = F << lv
    if lv > 0:
        layers.append((f"enc{lv+1}c1" if lv < 4 else "bottc1", lv, c // 2, 0, c, 0, 0))
    layers.append((f"enc{lv+1}c2" if lv < 4 else "bottc2", lv, c, 0, c, 1, 0))
for lv in (3, 2, 1, 0):
    c = F << lv
    layers.append((f"dec{lv+1}c1", lv, c, c, c, 0, 1))
    layers.append((f"dec{lv+1}c2", lv, c, 0, c, 1, 0))


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


tot = {"fprop": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}
print(f"{'layer':8s} {'pass':6s} {'ms':>7s} {'floor':>7s} {'x':>5s} {'TF/s':>6s} {'GB/s':>6s}")
for name, lv, c0, c1, cout, xf0, xf1 in [l for l in layers if l[4] >= int(os.environ.get("SWEEP_MIN_COUT", "0"))]:
    h, w = H >> lv, W >> lv
    cin = c0 + c1
    x0 = torch.randn(B, h, w, c0, device=dev).to(bf)
    x1 = torch.randn(B, h, w, c1, device=dev).to(bf) if c1 else None
    mk = lambda c: E.BNState(torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1)
    src = E.Src(x0, c0, mk(c0) if xf0 else None, x1, c1, mk(c1) if (c1 and xf1) else None)
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    y = torch.empty(B, h, w, cout, device=dev, dtype=bf)
    dy = torch.randn(B, h, w, cout, device=dev).to(bf)
    flops = 2.0 * B * h * w * 9 * cin * cout
    px = B * h * w
    wp = eng._pack("f" + name, wt, L.PACK_CONV_FPROP, cout, cin)
    stats = torch.empty(eng._stat_blocks(cout, B, h, w, src), 2, cout, device=dev)
    wpd = eng._pack("d" + name, wt, L.PACK_CONV_DGRAD, cout, cin)
    d0 = torch.empty(B, h, w, c0, device=dev, dtype=bf)
    d1 = torch.empty(B, h, w, c1, device=dev, dtype=bf) if c1 else None
    runs = {
        "fprop": (lambda: eng._conv(src, wp, cout, 9, B, h, w, y, stats=stats), px * (cin + cout) * 2),
        "dgrad": (lambda: eng._conv(E.Src(dy, cout), wpd, cin, 9, B, h, w, d0, y1=d1, split=c0 if c1 else 0), px * (cin + cout) * 2),
        "wgrad": (lambda: eng._wgrad(src, dy, cout, 9, B, h, w), px * (cin + cout) * 2),
    }
    for k, (fn, nbytes) in runs.items():
        ms = timeit(fn)
        floor = max(flops / 2.5e15, nbytes / 6e12) * 1e3
        tot[k][0] += ms; tot[k][1] += floor
        print(f"{name:8s} {k:6s} {ms:7.3f} {floor:7.3f} {ms/floor:5.1f} {flops/ms/1e9:6.0f} {nbytes/ms/1e6:6.0f}")
    del x0, x1, y, dy, d0, d1
for k, (ms, fl) in tot.items():
    print(f"total {k}: {ms:.2f} ms, floor {fl:.2f} ms")
