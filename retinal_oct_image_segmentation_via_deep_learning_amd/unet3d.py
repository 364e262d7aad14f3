"""Volumetric U-Net (BASELINE configs[4], "cfg5": 3-D OCT volumes 64 x 512 x 512).

The reference ships no 3-D network -- only the unused `ffc3d` flag of its Fourier blocks
(SOTAS/Lesions_Segment/YNet_2022.py:161,194) -- so this class is defined BY ANALOGY with the reference's 2-D `UNet`
(YNet_2022.py:509-602): same constructor signature, same module names and nesting with every 2-D layer replaced by
its 3-D sibling (Conv3d 3x3x3 p=1 bias=False, BatchNorm3d, ReLU, MaxPool3d(2), ConvTranspose3d(k=2, s=2), Conv3d 1x1x1,
channel softmax), hence the same 118 state_dict keys with 5-D conv weights and torch's default init.  Its oracle is
the same network on stock torch.nn (oracle/torch_unet3d.py); parity is UNPINNED by the reference.
The torch.nn members are parameter containers: forward runs engine3d.UNet3DEngine on liboct_hip.so.
"""
from __future__ import annotations

from collections import OrderedDict

import torch.nn as nn

from .engine3d import UNet3DEngine
from .unet import _EngineNet


def _block3d(cin: int, cout: int, name: str) -> nn.Sequential:
    layers = OrderedDict()
    for i, ci in ((1, cin), (2, cout)):
        layers[f"{name}conv{i}"] = nn.Conv3d(ci, cout, kernel_size=3, padding=1, bias=False)
        layers[f"{name}norm{i}"] = nn.BatchNorm3d(cout)
        layers[f"{name}relu{i}"] = nn.ReLU(inplace=True)
    return nn.Sequential(layers)


class UNet3D(_EngineNet):
    def __init__(self, in_channels=1, out_channels=2, init_features=32, compute_dtype="bf16"):
        super().__init__()
        f = init_features
        self.encoder1 = _block3d(in_channels, f, "enc1")
        self.pool1 = nn.MaxPool3d(kernel_size=2, stride=2)
        self.encoder2 = _block3d(f, f * 2, "enc2")
        self.pool2 = nn.MaxPool3d(kernel_size=2, stride=2)
        self.encoder3 = _block3d(f * 2, f * 4, "enc3")
        self.pool3 = nn.MaxPool3d(kernel_size=2, stride=2)
        self.encoder4 = _block3d(f * 4, f * 8, "enc4")
        self.pool4 = nn.MaxPool3d(kernel_size=2, stride=2)
        self.bottleneck = _block3d(f * 8, f * 16, "bottleneck")
        for k, mult in ((4, 8), (3, 4), (2, 2), (1, 1)):
            setattr(self, f"upconv{k}", nn.ConvTranspose3d(f * mult * 2, f * mult, kernel_size=2, stride=2))
            setattr(self, f"decoder{k}", _block3d(f * mult * 2, f * mult, f"dec{k}"))
        self.conv = nn.Conv3d(f, out_channels, kernel_size=1)
        self.softmax = nn.Softmax(dim=1)
        self._engine = UNet3DEngine(in_channels, out_channels, f, compute_dtype)
