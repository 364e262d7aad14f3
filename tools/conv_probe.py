"""Micro-benchmark of one conv launch through the C ABI (for rocprofv3 --pmc passes).
usage: conv_probe.py KIND n h w c0 c1 cout [iters]   KIND in fprop|fprop_nostats|wgrad|deconv"""
import sys, time
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retinal_oct_image_segmentation_via_deep_learning_amd import _lib as L, engine as E

kind = sys.argv[1]
n, h, w, c0, c1, cout = (int(v) for v in sys.argv[2:8])
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 20
eng = E.UNetEngine(1, 2, 4, "bf16")
dev = "cuda"
torch.manual_seed(0)
bf = torch.bfloat16
x0 = torch.randn(n, h, w, c0, device=dev).to(bf)
x1 = torch.randn(n, h, w, c1, device=dev).to(bf) if c1 else None
bn0 = E.BNState(torch.rand(c0, device=dev) + 0.5, torch.randn(c0, device=dev) * 0.1)
bn1 = E.BNState(torch.rand(c1, device=dev) + 0.5, torch.randn(c1, device=dev) * 0.1) if c1 else None
src = E.Src(x0, c0, bn0, x1, c1, bn1)
cin = c0 + c1
flops = 2.0 * n * h * w * 9 * cin * cout
if kind.startswith("fprop"):
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    wp = eng._pack("w", wt, L.PACK_CONV_FPROP, cout, cin)
    y = torch.empty(n, h, w, cout, device=dev, dtype=bf)
    stats = None
    if kind == "fprop":
        stats = torch.empty(eng._stat_blocks(cout, n, h, w, src), 2, cout, device=dev)
    run = lambda: eng._conv(src, wp, cout, 9, n, h, w, y, stats=stats)
elif kind == "wgrad":
    dy = torch.randn(n, h, w, cout, device=dev).to(bf)
    run = lambda: eng._wgrad(src, dy, cout, 9, n, h, w)
elif kind == "deconv":
    wt = torch.randn(cin, cout, 2, 2, device=dev) * 0.05
    wp = eng._pack("w", wt, L.PACK_DECONV_FPROP, cout, cin)
    y = torch.empty(n, 2 * h, 2 * w, cout, device=dev, dtype=bf)
    b = torch.zeros(cout, device=dev)
    flops = 2.0 * n * h * w * cin * cout * 4
    run = lambda: eng._conv(src, wp, 4 * cout, 1, n, h, w, y, out_mode=L.OUT_D2S, bias=b)
else:
    raise SystemExit("bad kind")
for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
nbytes = (x0.numel() + (x1.numel() if c1 else 0) + n * h * w * cout) * 2
print(f"{kind} n{n} {h}x{w} {c0}+{c1}->{cout}: {dt*1e6:.1f} us  {flops/dt/1e12:.1f} TFLOP/s  {nbytes/dt/1e9:.0f} GB/s(min traffic)")
