"""CPU oracle (TEST INFRASTRUCTURE ONLY): the ReLayNet block family restated on stock torch.nn from its
description -- same sub-module names, hence the same state_dict keys and seeded default init:
  /root/reference/SOTAS/Lesions_Segment/ReLayNet_2017.py:133-168  BasicBlock  conv(7x3, pad (3,1), bias) -> BN -> PReLU()
  :171-179  EncoderBlock  + MaxPool2d(2,2,return_indices) -> (pooled, block output, indices)
  :182-191  DecoderBlock  MaxUnpool2d(2,2)(x, indices); cat((skip, unpooled), 1); BasicBlock
  :194-203  ClassifierBlock  1x1 conv, logits out (the Softmax2d member is never applied)
  :21-126   ReLayNet  three encoders, bottleneck, three decoders, classifier
Pinned by tests/test_oracle_relaynet.py against fixtures generated from the reference classes
(tools/gen_golden_relaynet.py).  Only tests/ may import this file.
"""
import torch
import torch.nn as nn


class Basic(nn.Module):
    def __init__(self, cin, cout, kh=7, kw=3):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, (kh, kw), padding=((kh - 1) // 2, (kw - 1) // 2))
        self.batchnorm = nn.BatchNorm2d(cout)
        self.prelu = nn.PReLU()

    def forward(self, x):
        return self.prelu(self.batchnorm(self.conv(x)))


class Encoder(Basic):
    def __init__(self, cin, cout):
        super().__init__(cin, cout)
        self.maxpool = nn.MaxPool2d(2, 2, return_indices=True)

    def forward(self, x):
        blk = Basic.forward(self, x)
        pooled, idx = self.maxpool(blk)
        return pooled, blk, idx


class Decoder(Basic):
    def __init__(self, cin, cout):
        super().__init__(cin, cout)
        self.unpool = nn.MaxUnpool2d(2, 2)

    def forward(self, x, skip, idx):
        return Basic.forward(self, torch.cat((skip, self.unpool(x, idx)), dim=1))


class Classifier(nn.Module):
    def __init__(self, cin, ncls):
        super().__init__()
        self.conv = nn.Conv2d(cin, ncls, 1)
        self.softmax = nn.Softmax2d()

    def forward(self, x):
        return self.conv(x)


class TorchReLayNet(nn.Module):
    def __init__(self, in_channels=1, num_classes=10, num_filters=64):
        super().__init__()
        f = num_filters
        self.encode1, self.encode2, self.encode3 = Encoder(in_channels, f), Encoder(f, f), Encoder(f, f)
        self.bottleneck = Basic(f, f)
        self.decode1, self.decode2, self.decode3 = Decoder(2 * f, f), Decoder(2 * f, f), Decoder(2 * f, f)
        self.classifier = Classifier(f, num_classes)

    def forward(self, x):
        e1, o1, i1 = self.encode1(x)
        e2, o2, i2 = self.encode2(e1)
        e3, o3, i3 = self.encode3(e2)
        d = self.decode1(self.bottleneck(e3), o3, i3)
        d = self.decode2(d, o2, i2)
        return self.classifier(self.decode3(d, o1, i1))
