"""Timeline of workgroup 0 of one 1x1 igemm2 launch (a transposed convolution's forward pass: depth-to-space store) from
in-kernel s_memtime stamps -- diagnostic build liboct_hip_TRACE.so (tools/build_variant.sh TRACE -DOCT_TRACE).
usage: trace1_probe.py n h w cin cout_deconv"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OCT_HIP_LIB"] = os.path.join(ROOT, "retinal_oct_image_segmentation_via_deep_learning_amd", os.environ.get("TRACE_LIB", "liboct_hip_TRACE.so"))
from retinal_oct_image_segmentation_via_deep_learning_amd import _lib as L, engine as E
n, h, w, cin, cout = (int(v) for v in sys.argv[1:6])
eng = E.UNetEngine(1, 2, 4, "bf16")
bf = torch.bfloat16
x0 = torch.randn(n, h, w, cin, device="cuda").to(bf)
bn0 = E.BNState(torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1)
src = E.Src(x0, cin, bn0)
wt = torch.randn(cin, cout, 2, 2, device="cuda") * 0.05
bias = torch.randn(cout, device="cuda")
wp = eng._pack("w", wt, L.PACK_DECONV_FPROP, cout, cin)
u = torch.empty(n, 2 * h, 2 * w, cout, device="cuda", dtype=bf)
trace = torch.zeros(8 * 256, dtype=torch.int64, device="cuda")
run = lambda: eng._conv(src, wp, 4 * cout, 1, n, h, w, u, out_mode=L.OUT_D2S, bias=bias)
for _ in range(3):
    run()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); run(); e.record(); torch.cuda.synchronize()
print(f"launch {s.elapsed_time(e)*1e3:.1f} us")
L.lib().oct_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
run()
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(8, 256).astype(np.int64)
ns = int((t[0] > 0).sum())
raw = trace.cpu().numpy()
print("stages traced:", ns, f"; whole WG {t[3, ns-1] - t[0, 0]} cycles in {(raw[2044] - raw[2043]) / 100:.1f} us")
c_phase = (t[1, :ns] - t[0, :ns]); c_epi = (t[2, :ns] - t[1, :ns]); c_bar = (t[3, :ns] - t[2, :ns])
p_commit = (t[5, :ns] - t[4, :ns]); p_issue = (t[6, :ns] - t[5, :ns]); p_bar = (t[7, :ns] - t[6, :ns])
stage = np.diff(t[3, :ns])
f = lambda a: f"med {np.median(a):8.0f}  mean {np.mean(a):8.0f}  max {np.max(a):8.0f}"
print("consumer: MFMA phase ", f(c_phase)); print("consumer: epilogue   ", f(c_epi)); print("consumer: barrier    ", f(c_bar))
print("producer: commit     ", f(p_commit)); print("producer: issue      ", f(p_issue)); print("producer: barrier    ", f(p_bar))
print("stage period         ", f(stage))
np.set_printoptions(linewidth=250)
k = min(ns, 36)
print("phase ", c_phase[:k]); print("epi   ", c_epi[:k]); print("cbar  ", c_bar[:k]); print("commit", p_commit[:k]); print("issue ", p_issue[:k]); print("pbar  ", p_bar[:k]); print("period", stage[:k])
