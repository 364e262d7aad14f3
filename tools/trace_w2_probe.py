"""Timeline of workgroup (0,0,0) of one wgrad2 launch from in-kernel s_memtime stamps (diagnostic build
liboct_hip_TRACE.so = tools/build_variant.sh TRACE "-DOCT_TRACE"; never the production library).
usage: trace_w2_probe.py n h w c0 c1 cout [partials]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OCT_HIP_LIB"] = os.path.join(ROOT, "retinal_oct_image_segmentation_via_deep_learning_amd", "liboct_hip_TRACE.so")
from retinal_oct_image_segmentation_via_deep_learning_amd import _lib as L, engine as E
n, h, w, c0, c1, cout = (int(v) for v in sys.argv[1:7])
eng = E.UNetEngine(1, 2, 4, "bf16")
eng.deterministic = len(sys.argv) > 7
bf = torch.bfloat16
x0 = torch.randn(n, h, w, c0, device="cuda").to(bf)
x1 = torch.randn(n, h, w, c1, device="cuda").to(bf) if c1 else None
bn0 = E.BNState(torch.rand(c0, device="cuda") + 0.5, torch.randn(c0, device="cuda") * 0.1)
bn1 = E.BNState(torch.rand(c1, device="cuda") + 0.5, torch.randn(c1, device="cuda") * 0.1) if c1 else None
src = E.Src(x0, c0, bn0, x1, c1, bn1)
dy = torch.randn(n, h, w, cout, device="cuda").to(bf)
trace = torch.zeros(8 * 256, dtype=torch.int64, device="cuda")
h_ = ctypes.CDLL(os.environ["OCT_HIP_LIB"])
for _ in range(3):
    eng._wgrad(src, dy, cout, 9, n, h, w)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5):
    eng._wgrad(src, dy, cout, 9, n, h, w)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
flops = 2.0 * n * h * w * 9 * (c0 + c1) * cout
print(f"launch {ms*1e3:.1f} us, {flops/ms/1e9:.0f} TFLOP/s")
h_.oct_debug_set_trace_w2(ctypes.c_void_p(trace.data_ptr()))
eng._wgrad(src, dy, cout, 9, n, h, w)
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(8, 256).astype(np.int64)
ns = int((t[0] > 0).sum())
print("stages traced:", ns)
phase, cbar = t[1, :ns] - t[0, :ns], t[2, :ns] - t[1, :ns]
period = np.diff(t[0, :ns])
commit, issue, pbar = t[5, :ns] - t[4, :ns], t[6, :ns] - t[5, :ns], t[7, :ns] - t[6, :ns]
f = lambda a: f"med {np.median(a):8.0f}  mean {np.mean(a):8.0f}  max {np.max(a):8.0f}"
print("consumer: MFMA phase ", f(phase)); print("consumer: barrier    ", f(cbar)); print("stage period         ", f(period))
print("producer: commit     ", f(commit)); print("producer: issue      ", f(issue)); print("producer: barrier    ", f(pbar))
print("tail (accumulator write-out):", t[3, 1] - t[3, 0], "cycles; whole loop", t[3, 0] - t[0, 0])
np.set_printoptions(linewidth=220)
k = min(ns, 16)
print("phase ", phase[:k]); print("cbar  ", cbar[:k]); print("commit", commit[:k]); print("issue ", issue[:k]); print("pbar  ", pbar[:k])
