"""Timeline of wave 0 / workgroup 0 of one igemm3 launch (diagnostic build liboct_hip_TRACE.so only).
usage: trace3_probe.py n h w c0 c1 cout [stats]"""
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
bn0 = E.BNState(torch.rand(c0, device="cuda") + 0.5, torch.randn(c0, device="cuda") * 0.1)
bn1 = E.BNState(torch.rand(c1, device="cuda") + 0.5, torch.randn(c1, device="cuda") * 0.1) if c1 else None
src = E.Src(x0, c0, bn0, x1, c1, bn1)
wt = torch.randn(cout, c0 + c1, 3, 3, device="cuda") * 0.05
wp = eng._pack("w", wt, L.PACK_CONV_FPROP, cout, c0 + c1)
y = torch.empty(n, h, w, cout, device="cuda", dtype=bf)
st = torch.empty(eng._stat_blocks(cout, n, h, w, src), 2, cout, device="cuda") if stats_on else None
trace = torch.zeros(48 * 256 + 8 * 9 * 64, dtype=torch.int64, device="cuda")
for _ in range(3):
    eng._conv(src, wp, cout, 9, n, h, w, y, stats=st)
L.lib().oct_debug_set_trace3(ctypes.c_void_p(trace.data_ptr()))
eng._conv(src, wp, cout, 9, n, h, w, y, stats=st)
torch.cuda.synchronize()
raw = trace.cpu().numpy().astype(np.int64)
T = raw[:48 * 256].reshape(8, 6, 256)
TT = raw[48 * 256:].reshape(8, 9, 64)
ns = int((T[0, 0] > 0).sum())
f = lambda a: f"med {np.median(a):6.0f} mean {np.mean(a):6.0f} max {np.max(a):6.0f}"
print("stages", ns)
t00 = T[0, 0, 0]
for wv in range(8):
    t = T[wv]
    print(f"wave {wv}: setup {f(t[1, :ns] - t[0, :ns])} | taps {f(t[2, :ns] - t[1, :ns])} | wait {f(t[3, :ns] - t[2, :ns])} | "
          f"epi med {np.median(t[4, :ns] - t[3, :ns]):5.0f} | bar {f(t[5, :ns] - t[4, :ns])}")
np.set_printoptions(linewidth=220)
s = 5   # one steady stage: absolute times relative to wave 0's stage start
print("stage", s, "times relative to wave 0 start: rows = waves, cols = [start, taps begin, taps end, wait end, at barrier, released]")
print((T[:, :, s] - T[0, 0, s]))
print("stage period", f(np.diff(T[0, 0, :ns])))

# per-tap durations (cycles) of stages 4..23, waves 0 and 4: tap t = stamp(t) - stamp(t-1) (tap 0 from the taps-begin stamp)
for wv in (0, 4):
    st = np.arange(4, min(ns, 24))
    begin = T[wv, 1, st]
    d = np.diff(np.concatenate([begin[None, :], TT[wv][:, st]], axis=0), axis=0)
    print(f"wave {wv} per-tap medians:", np.median(d, axis=1).astype(int))
