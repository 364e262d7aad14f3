"""Metrics kernels on the SURVEY App. B pair (32 x 512 x 1024 masks, seeded): call-level time with HIP events, and --
when run under `rocprofv3 --kernel-trace --stats` (tools/metrics_prof.sh) -- the kernel durations for profiles/.
Prints achieved GB/s against the 6.3 TB/s the chip streams (MI355X_MICROARCH.md) for: binary uint8 confusion counts
(33.5 MB per pass), the same on int64 masks (268 MB), per-class counts of uint8 / int64 class maps, squared error,
column |difference|."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from retinal_oct_image_segmentation_via_deep_learning_amd import Metrics, _lib as L  # noqa: E402

rng = np.random.default_rng(1234)
a8 = (rng.random((32, 512, 1024)) < 0.3).astype(np.uint8)
b8 = (rng.random((32, 512, 1024)) < 0.3).astype(np.uint8)
A8, B8 = torch.from_numpy(a8).cuda(), torch.from_numpy(b8).cuda()
res = Metrics.evaluate(A8, B8)
print("dice", repr(float(res["dice_coefficient"])), "iou", repr(float(res["iou_score"])), "accuracy", repr(float(res["accuracy"])))
assert float(res["dice_coefficient"]) == 0.2996415616703051 and float(res["iou_score"]) == 0.17622258631812396
C8 = torch.from_numpy(rng.integers(0, 8, (32, 512, 1024)).astype(np.uint8)).cuda()
D8 = torch.from_numpy(rng.integers(0, 8, (32, 512, 1024)).astype(np.uint8)).cuda()
A64, B64, C64, D64 = A8.long(), B8.long(), C8.long(), D8.long()
lib, st = L.lib(), torch.cuda.current_stream().cuda_stream
oi, of = torch.empty(6, dtype=torch.int64, device="cuda"), torch.empty(6, dtype=torch.float64, device="cuda")
oc, sc = torch.empty((8, 6), dtype=torch.int64, device="cuda"), torch.empty(48, dtype=torch.int64, device="cuda")
od = torch.empty(1, dtype=torch.float64, device="cuda")
n = A8.numel()
cases = [
    ("confusion_counts uint8 (binary masks)", 2 * n, lambda: lib.oct_confusion_counts(A8.data_ptr(), B8.data_ptr(), 0, n, oi.data_ptr(), of.data_ptr(), st)),
    ("confusion_counts int64", 16 * n, lambda: lib.oct_confusion_counts(A64.data_ptr(), B64.data_ptr(), 2, n, oi.data_ptr(), of.data_ptr(), st)),
    ("class_confusion_counts uint8, 8 classes", 2 * n, lambda: lib.oct_class_confusion_counts(C8.data_ptr(), D8.data_ptr(), 0, n, 8, oc.data_ptr(), sc.data_ptr(), st)),
    ("class_confusion_counts int64, 8 classes", 16 * n, lambda: lib.oct_class_confusion_counts(C64.data_ptr(), D64.data_ptr(), 2, n, 8, oc.data_ptr(), sc.data_ptr(), st)),
    ("sqdiff_sum uint8", 2 * n, lambda: lib.oct_sqdiff_sum(A8.data_ptr(), B8.data_ptr(), 0, n, od.data_ptr(), st)),
    ("column_absdiff_sum uint8 (axis 0 of 32 x 524288)", 2 * n, lambda: lib.oct_column_absdiff_sum(A8.data_ptr(), B8.data_ptr(), 0, 1, 32, 512 * 1024, od.data_ptr(), st)),
]
for name, nbytes, fn in cases:
    for _ in range(3):
        assert fn() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:52s} {nbytes / 1e6:7.1f} MB  {us:8.1f} us per call (all launches of the call)  {nbytes / us / 1e3:7.0f} GB/s  {nbytes / us / 1e3 / 6300:5.2f} of 6.3 TB/s")
