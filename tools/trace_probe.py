"""Timeline of workgroup 0 of one igemm2 launch from in-kernel s_memtime stamps (diagnostic build
liboct_hip_TRACE.so, never the production library).  usage: trace_probe.py n h w c0 c1 cout [stats]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OCT_HIP_LIB"] = os.path.join(ROOT, "retinal_oct_image_segmentation_via_deep_learning_amd", os.environ.get("TRACE_LIB", "liboct_hip_TRACE.so"))
from retinal_oct_image_segmentation_via_deep_learning_amd import _lib as L, engine as E
n, h, w, c0, c1, cout = (int(v) for v in sys.argv[1:7])
stats_on = len(sys.argv) > 7
eng = E.UNetEngine(1, 2, 4, "bf16")
bf = torch.bfloat16
x0 = torch.randn(n, h, w, c0, device="cuda").to(bf)
x1 = torch.randn(n, h, w, c1, device="cuda").to(bf) if c1 else None
noxf = bool(os.environ.get("TRACE_NOXF"))   # no transform on load: the data-gradient case (LDS-DMA staging when eligible)
bn0 = None if noxf else E.BNState(torch.rand(c0, device="cuda") + 0.5, torch.randn(c0, device="cuda") * 0.1)
bn1 = E.BNState(torch.rand(c1, device="cuda") + 0.5, torch.randn(c1, device="cuda") * 0.1) if (c1 and not noxf) else None
src = E.Src(x0, c0, bn0, x1, c1, bn1)
wt = torch.randn(cout, c0 + c1, 3, 3, device="cuda") * 0.05
wp = eng._pack("w", wt, L.PACK_CONV_FPROP, cout, c0 + c1)
y = torch.empty(n, h, w, cout, device="cuda", dtype=bf)
st = torch.empty(eng._stat_blocks(cout, n, h, w, src), 2, cout, device="cuda") if stats_on else None
trace = torch.zeros(8 * 256, dtype=torch.int64, device="cuda")
h_ = L.lib()
for _ in range(3):
    eng._conv(src, wp, cout, 9, n, h, w, y, stats=st)
h_.oct_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
eng._conv(src, wp, cout, 9, n, h, w, y, stats=st)
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(8, 256).astype(np.int64)
ns = int((t[0] > 0).sum())
print("stages traced:", ns)
raw = trace.cpu().numpy()
print(f"calibration: 64 MFMAs in {raw[2040]} shader cycles = {raw[2040]/64:.1f} cyc/MFMA; realtime ticks {raw[2041]} -> clock {raw[2040]/max(raw[2041],1)*100:.0f} MHz")
span_rt = raw[2044] - raw[2043]; span_sc = t[3, ns-1] - t[0, 0]
print(f"whole WG: {span_sc} shader cycles in {span_rt/100:.1f} us -> {span_sc/max(span_rt,1)*100:.0f} MHz")
c_phase = (t[1, :ns] - t[0, :ns]); c_epi = (t[2, :ns] - t[1, :ns]); c_bar = (t[3, :ns] - t[2, :ns])
p_commit = (t[5, :ns] - t[4, :ns]); p_issue = (t[6, :ns] - t[5, :ns]); p_bar = (t[7, :ns] - t[6, :ns])
stage = np.diff(t[3, :ns])
f = lambda a: f"med {np.median(a):8.0f}  mean {np.mean(a):8.0f}  max {np.max(a):8.0f}"
print("consumer: MFMA phase ", f(c_phase)); print("consumer: epilogue   ", f(c_epi)); print("consumer: barrier    ", f(c_bar))
print("producer: commit     ", f(p_commit)); print("producer: issue      ", f(p_issue)); print("producer: barrier    ", f(p_bar))
print("producer: busy (4->6)", f(t[6, :ns] - t[4, :ns]), " (commit + issue together: the interleaved loop stamps no slot 5)")
print("consumer: top (3->0') ", f(t[0, 1:ns] - t[3, :ns - 1]), " (barrier release -> next MFMA phase start)")
print("stage period         ", f(stage))
print("first 10 stages: phase", c_phase[:10], "epi", c_epi[:10], "cbar", c_bar[:10], "commit", p_commit[:10], "pbar", p_bar[:10])
if os.environ.get("TRACE_FULL"):
    np.set_printoptions(linewidth=250)
    k = min(ns, 24)
    print("phase ", c_phase[:k]); print("epi   ", c_epi[:k]); print("cbar  ", c_bar[:k]); print("commit", p_commit[:k]); print("issue ", p_issue[:k]); print("pbar  ", p_bar[:k]); print("period", stage[:k])
    print("pbusy ", (t[6, :k] - t[4, :k])); print("ctop  ", (t[0, 1:k + 1] - t[3, :k]))
    print("producer start rel. consumer phase start:", (t[4, :k] - t[0, :k]))
