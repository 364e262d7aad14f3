"""CPU oracle #2 (TEST INFRASTRUCTURE ONLY): the reference's U-Net restated on stock torch.nn.

The reference's arithmetic provider IS torch: this module rebuilds the network of
/root/reference/SOTAS/Lesions_Segment/YNet_2022.py:509-602 from its description -- same module
names, hence the same state_dict -- and runs it with torch's own CPU kernels.  It serves as
  * the cross-check of oracle/ref_cpu.py's hand-written backward at sizes numpy finishes slowly,
  * the `cpu_baseline` ("port") that bench.py times on the GPU box's host cores
    (the reference's own files cannot travel to the GPU box).
Pinned by tests/test_oracle.py against the fixtures generated from the reference.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


def _stage(cin, cout, tag):
    mods = OrderedDict()
    for i in (1, 2):
        mods[f"{tag}conv{i}"] = nn.Conv2d(cin if i == 1 else cout, cout, 3, padding=1, bias=False)
        mods[f"{tag}norm{i}"] = nn.BatchNorm2d(cout)
        mods[f"{tag}relu{i}"] = nn.ReLU(inplace=True)
    return nn.Sequential(mods)


class TorchUNet(nn.Module):
    def __init__(self, in_channels=3, out_channels=1, init_features=32):
        super().__init__()
        f = init_features
        widths = [f, 2 * f, 4 * f, 8 * f]
        cin = in_channels
        for lvl, wd in enumerate(widths, start=1):
            setattr(self, f"encoder{lvl}", _stage(cin, wd, f"enc{lvl}"))
            setattr(self, f"pool{lvl}", nn.MaxPool2d(2, 2))
            cin = wd
        self.bottleneck = _stage(8 * f, 16 * f, "bottleneck")
        for lvl in (4, 3, 2, 1):
            wd = widths[lvl - 1]
            setattr(self, f"upconv{lvl}", nn.ConvTranspose2d(2 * wd, wd, 2, stride=2))
            setattr(self, f"decoder{lvl}", _stage(2 * wd, wd, f"dec{lvl}"))
        self.conv = nn.Conv2d(f, out_channels, 1)

    def forward(self, x):
        skips = []
        for lvl in (1, 2, 3, 4):
            x = getattr(self, f"encoder{lvl}")(x)
            skips.append(x)
            x = getattr(self, f"pool{lvl}")(x)
        x = self.bottleneck(x)
        for lvl in (4, 3, 2, 1):
            x = getattr(self, f"upconv{lvl}")(x)
            x = getattr(self, f"decoder{lvl}")(torch.cat((x, skips[lvl - 1]), dim=1))
        return torch.softmax(self.conv(x), dim=1)


def _bio_stage(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
                         nn.Conv2d(cout, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class TorchBioUNet(nn.Module):
    """/root/reference/SOTAS/Layers_Segment/BioNet_2020.py:24-75 restated from its description
    (same module names and construction order, hence the same state_dict and seeded init); pinned
    by tests/golden/bionet_unet_*.npz, which tools/gen_golden_bionet.py made from the reference class."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        widths = [64, 128, 256, 512]
        cin = in_channels
        for lvl, wd in enumerate(widths, start=1):
            setattr(self, f"enc{lvl}", _bio_stage(cin, wd))
            cin = wd
        for lvl in (4, 3, 2):
            wd = widths[lvl - 2]
            setattr(self, f"up{lvl}", nn.ConvTranspose2d(2 * wd, wd, 2, stride=2))
            setattr(self, f"dec{lvl}", _bio_stage(2 * wd, wd))
        self.final = nn.Conv2d(64, out_channels, 1)
        self.maxpool = nn.MaxPool2d(2)

    def forward(self, x):
        skips = []
        for lvl in (1, 2, 3):
            x = getattr(self, f"enc{lvl}")(x)
            skips.append(x)
            x = self.maxpool(x)
        x = self.enc4(x)
        for lvl in (4, 3, 2):
            x = getattr(self, f"dec{lvl}")(torch.cat([skips[lvl - 2], getattr(self, f"up{lvl}")(x)], dim=1))
        return self.final(x)


def loss_fn(probs, target, w_ce=1.0, w_dice=0.0, eps=1e-7):
    ce = F.nll_loss(torch.log(probs), target)
    if w_dice == 0.0:
        return w_ce * ce
    onehot = F.one_hot(target, probs.shape[1]).permute(0, 3, 1, 2).to(probs.dtype)
    inter, ps, ys = (probs * onehot).sum((0, 2, 3)), probs.sum((0, 2, 3)), onehot.sum((0, 2, 3))
    return w_ce * ce + w_dice * (1.0 - ((2 * inter + eps) / (ps + ys + eps)).mean())


def physical_cores():
    """Physical cores this process may use: the affinity mask with SMT siblings counted once, capped by the
    cgroup CPU quota when the container has one (a 16-CPU share of a 128-core host shows all 128 in the mask)."""
    import os
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    seen = set()
    for c in cpus:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as fh:
                seen.add(fh.read().strip())
        except OSError:
            seen.add(str(c))
    n = max(1, len(seen))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                parts = fh.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]) + 0.5)))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh2:
                        n = min(n, max(1, int(q / int(fh2.read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def time_train_steps(batch, height, width, classes=8, features=32, iters=3, threads=None, model="unet",
                     budget_s=None):
    """fwd + loss + bwd + SGD on the host cores (fp32), 1 warm-up + up to `iters` timed iterations (stops
    early once `budget_s` seconds of timed work are spent, never before 2 iterations).
    model "unet": TorchUNet(1, classes, features) with nll_loss(log p); "bionet": TorchBioUNet(1, classes) with
    cross_entropy on its logits (BASELINE cfg1).  Returns a dict (B-scans/s from the minimum and the median)."""
    import statistics
    import time
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(0)
    if model == "bionet":
        net = TorchBioUNet(1, classes).train()
        step_loss = lambda out, t: F.cross_entropy(out, t)   # noqa: E731
    else:
        net = TorchUNet(1, classes, features).train()
        step_loss = loss_fn
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(batch, 1, height, width, generator=g)
    t = torch.randint(0, classes, (batch, height, width), generator=g)
    times = []
    for i in range(iters + 1):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        step_loss(net(x), t).backward()
        opt.step()
        if i:  # first iteration is warm-up
            times.append(time.perf_counter() - t0)
            if budget_s is not None and len(times) >= 2 and sum(times) >= budget_s:
                break
    best, med = min(times), statistics.median(times)
    return {"bscans_per_s_min": batch / best, "bscans_per_s_median": batch / med, "s_per_iter_min": best,
            "s_per_iter_median": med, "iters": len(times), "threads": torch.get_num_threads()}
