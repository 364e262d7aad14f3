"""CPU oracle (TEST INFRASTRUCTURE ONLY) of the volumetric U-Net (cfg5): stock torch.nn, nothing else.

There is no reference counterpart to pin against (only the unused `ffc3d` flag,
/root/reference/SOTAS/Lesions_Segment/YNet_2022.py:161,194) -- PARITY UNPINNED BY THE REFERENCE.  The network is
the 3-D transcription of the reference's 2-D UNet (YNet_2022.py:509-602, restated in oracle/torch_unet.py): every
Conv2d / BatchNorm2d / MaxPool2d / ConvTranspose2d becomes its 3-D sibling with the same hyper-parameters, the
forward is the same graph.  Only tests/, smoke() and bench.py's cpu_baseline leg may import this file.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


def _block(cin, cout, name):
    layers = OrderedDict()
    for i, ci in ((1, cin), (2, cout)):
        layers[f"{name}conv{i}"] = nn.Conv3d(ci, cout, 3, padding=1, bias=False)
        layers[f"{name}norm{i}"] = nn.BatchNorm3d(cout)
        layers[f"{name}relu{i}"] = nn.ReLU(inplace=True)
    return nn.Sequential(layers)


class TorchUNet3D(nn.Module):
    def __init__(self, in_channels=1, out_channels=2, init_features=32):
        super().__init__()
        f = init_features
        self.encoder1, self.pool1 = _block(in_channels, f, "enc1"), nn.MaxPool3d(2, 2)
        self.encoder2, self.pool2 = _block(f, 2 * f, "enc2"), nn.MaxPool3d(2, 2)
        self.encoder3, self.pool3 = _block(2 * f, 4 * f, "enc3"), nn.MaxPool3d(2, 2)
        self.encoder4, self.pool4 = _block(4 * f, 8 * f, "enc4"), nn.MaxPool3d(2, 2)
        self.bottleneck = _block(8 * f, 16 * f, "bottleneck")
        for k, m in ((4, 8), (3, 4), (2, 2), (1, 1)):
            setattr(self, f"upconv{k}", nn.ConvTranspose3d(2 * m * f, m * f, 2, stride=2))
            setattr(self, f"decoder{k}", _block(2 * m * f, m * f, f"dec{k}"))
        self.conv = nn.Conv3d(f, out_channels, 1)
        self.softmax = nn.Softmax(dim=1)

    def forward(self, x):
        e1 = self.encoder1(x)
        e2 = self.encoder2(self.pool1(e1))
        e3 = self.encoder3(self.pool2(e2))
        e4 = self.encoder4(self.pool3(e3))
        d = self.bottleneck(self.pool4(e4))
        for k, e in ((4, e4), (3, e3), (2, e2), (1, e1)):
            d = getattr(self, f"decoder{k}")(torch.cat((getattr(self, f"upconv{k}")(d), e), dim=1))
        return self.softmax(self.conv(d))


def loss_fn(probs, target, w_ce=1.0, w_dice=0.0, eps=1e-7):
    ce = F.nll_loss(torch.log(probs), target)
    if w_dice == 0.0:
        return w_ce * ce
    onehot = F.one_hot(target, probs.shape[1]).permute(0, 4, 1, 2, 3).to(probs.dtype)
    dims = (0, 2, 3, 4)
    inter, ps, ys = (probs * onehot).sum(dims), probs.sum(dims), onehot.sum(dims)
    return w_ce * ce + w_dice * (1.0 - ((2 * inter + eps) / (ps + ys + eps)).mean())
