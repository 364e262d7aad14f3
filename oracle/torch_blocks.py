"""CPU oracle (TEST INFRASTRUCTURE ONLY): the reference's other block families restated on stock
torch.nn, from their description -- same sub-module names, hence the same state_dict keys and
seeded default init:
  /root/reference/SOTAS/Layers_Segment/MGUNet_2021.py:42-108     UnetConv, UnetUp, UnetUp4
  /root/reference/SOTAS/Layers_Segment/SD_Layer_Net/common.py:6-41,64-91   conv_block, up_conv, Attention_block
  /root/reference/SOTAS/Layers_Segment/SD_Layer_Net/unet.py:8-150          U_Net, AttU_Net
Pinned by tests/test_oracle_blocks.py against fixtures generated from the reference classes
(tools/gen_golden_blocks.py).  Only tests/ may import this file.
"""
from collections import OrderedDict

import torch
import torch.nn as nn


def _cbr(cin, cout, bn):
    mods = [nn.Conv2d(cin, cout, 3, 1, 1)] + ([nn.BatchNorm2d(cout)] if bn else []) + [nn.ReLU(inplace=True)]
    return nn.Sequential(*mods)


class UnetConv(nn.Module):
    def __init__(self, in_channels, out_channels, is_batchnorm=True):
        super().__init__()
        self.conv1 = _cbr(in_channels, out_channels, is_batchnorm)
        self.conv2 = _cbr(out_channels, out_channels, is_batchnorm)

    def forward(self, x):
        return self.conv2(self.conv1(x))


class UnetUp(nn.Module):
    K = 2

    def __init__(self, in_channels, out_channels, is_deconv=True):
        super().__init__()
        if is_deconv:
            self.up = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=self.K, stride=self.K)
        else:
            self.up = nn.Sequential(nn.UpsamplingBilinear2d(scale_factor=self.K), nn.Conv2d(in_channels, out_channels, 1))
        self.conv = UnetConv(in_channels, out_channels, True)

    def forward(self, x1, x2):
        return self.conv(torch.cat([x2, self.up(x1)], dim=1))


class UnetUp4(UnetUp):
    K = 4


# Dropout2d (common.py:13,17,34) under a FIXED mask: tests set DROPOUT_MASKS to the keep flags the fixture recorded ([n, c] each, in
# forward order); None is torch's own random Dropout2d.
DROPOUT_MASKS = None
_mask_at = [0]


def set_dropout_masks(masks):
    global DROPOUT_MASKS
    DROPOUT_MASKS = list(masks) if masks else None
    _mask_at[0] = 0


class Drop2d(nn.Dropout2d):
    """nn.Dropout2d (no parameters, no buffers: the state_dict does not see the difference) with the mask injectable"""

    def forward(self, x):
        if DROPOUT_MASKS is None or not self.training or self.p == 0:
            return super().forward(x)
        m = DROPOUT_MASKS[_mask_at[0] % len(DROPOUT_MASKS)]
        _mask_at[0] += 1
        return x * (m.to(x.dtype) / (1.0 - self.p))[:, :, None, None]


class conv_block(nn.Module):
    def __init__(self, ch_in, ch_out, act=nn.ReLU, drop_rate=0.0):
        super().__init__()
        self.init_conv = nn.Conv2d(ch_in, ch_out, 3, 1, 1)
        self.conv = nn.Sequential(nn.Conv2d(ch_out, ch_out, 3, 1, 1), nn.BatchNorm2d(ch_out), Drop2d(drop_rate),
                                  act(), nn.Conv2d(ch_out, ch_out, 3, 1, 1), nn.BatchNorm2d(ch_out), Drop2d(drop_rate))
        self.activation = act()

    def forward(self, x):
        i = self.init_conv(x)
        return self.activation(self.conv(i) + i)


class up_conv(nn.Module):
    def __init__(self, ch_in, ch_out, act=nn.ReLU, drop_rate=0.0, scale_factor=2):
        super().__init__()
        self.up = nn.Sequential(nn.Upsample(scale_factor=scale_factor, mode="bilinear", align_corners=True),
                                nn.Conv2d(ch_in, ch_out, 3, 1, 1), nn.BatchNorm2d(ch_out), Drop2d(drop_rate), act())

    def forward(self, x):
        return self.up(x)


class Attention_block(nn.Module):
    def __init__(self, channels_g, channels_x, F_int):
        super().__init__()
        self.W_g = nn.Sequential(nn.Conv2d(channels_g, F_int, 1), nn.BatchNorm2d(F_int))
        self.W_x = nn.Sequential(nn.Conv2d(channels_x, F_int, 1), nn.BatchNorm2d(F_int))
        self.psi = nn.Sequential(nn.Conv2d(F_int, 1, 1), nn.BatchNorm2d(1), nn.Sigmoid())
        self.relu = nn.ReLU()

    def forward(self, g, x):
        return x * self.psi(self.relu(self.W_g(g) + self.W_x(x)))


class _SDNet(nn.Module):
    def __init__(self, img_ch, output_ch, channels, gates, head_in):
        super().__init__()
        self._n = len(channels)
        self.Maxpool = nn.MaxPool2d(2, 2)
        self.Conv1 = conv_block(img_ch, channels[0])
        for i in range(1, self._n):
            setattr(self, f"Conv{i + 1}", conv_block(channels[i - 1], channels[i]))
        for i in range(self._n, 1, -1):
            setattr(self, f"Up{i}", up_conv(channels[i - 1], channels[i - 2]))
            if gates:
                setattr(self, f"Att{i}", Attention_block(channels[i - 2], channels[i - 2], channels[i - 2] // 2))
            setattr(self, f"Up_conv{i}", conv_block(channels[i - 1], channels[i - 2]))
        self.Conv_1x1 = nn.Conv2d(head_in, output_ch, 1)
        self._gates = gates

    def forward(self, x):
        feats = []
        for i in range(1, self._n + 1):
            x = getattr(self, f"Conv{i}")(x if i == 1 else self.Maxpool(x))
            feats.append(x)
        d = feats[-1]
        for i in range(self._n, 1, -1):
            d = getattr(self, f"Up{i}")(d)
            skip = feats[i - 2]
            if self._gates:
                skip = getattr(self, f"Att{i}")(d, skip)
            d = getattr(self, f"Up_conv{i}")(torch.cat((skip, d), dim=1))
        return self.Conv_1x1(d)


class U_Net(_SDNet):
    def __init__(self, img_ch=3, output_ch=1, channels=(64, 128, 256, 512, 1024)):
        super().__init__(img_ch, output_ch, list(channels), False, 64)


class AttU_Net(_SDNet):
    def __init__(self, img_ch=1, output_ch=1, channels=(64, 128, 256, 512, 1024)):
        super().__init__(img_ch, output_ch, list(channels), True, channels[0])


class AttU_Net4(_SDNet):      # SD_Layer_Net/unet.py:153-214: four levels
    def __init__(self, img_ch=1, output_ch=1, channels=(64, 128, 256, 512)):
        super().__init__(img_ch, output_ch, list(channels), True, channels[0])


# ---- MGU-Net: /root/reference/SOTAS/Layers_Segment/MGUNet_2021.py:29-39 (Basconv), :110-148 (GloRe_Unit), :150-194 (MGR_Module),
# ---- :197-252 (MGUNet), :255-309 (MGUNet_2); pinned by tests/test_oracle_mgunet.py against tools/gen_golden_mgunet.py's fixtures
class Basconv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, padding=1):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size, padding=padding),
                                  nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.conv(x)


class GloRe_Unit(nn.Module):
    """node features S, P = two 1x1 projections; adjacency = row-softmax(S P^T / sqrt(hw)); out = x + extend(adjacency P)"""

    def __init__(self, in_channels, out_channels, kernel=1):
        super().__init__()
        self.conv_state = nn.Conv2d(in_channels, out_channels, 1)
        self.conv_proj = nn.Conv2d(in_channels, out_channels, 1)
        self.conv_extend = nn.Conv2d(out_channels, in_channels, 1)

    def forward(self, x):
        s, p = self.conv_state(x).flatten(2), self.conv_proj(x).flatten(2)          # [n, M, hw]
        adj = torch.softmax(torch.einsum("nip,njp->nij", s, p) / (s.shape[2] ** 0.5), dim=2)
        return x + self.conv_extend(torch.einsum("nij,njp->nip", adj, p).reshape(x.shape[0], -1, *x.shape[2:]))


class MGR_Module(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        for i, (k, m) in enumerate(((1, out_channels), (2, out_channels), (3, out_channels // 2), (5, out_channels // 2))):
            setattr(self, f"conv{i}_1", Basconv(in_channels, out_channels))
            if i:
                setattr(self, f"pool{i}", nn.MaxPool2d(kernel_size=[k, k], stride=k))
                setattr(self, f"conv{i}_2", Basconv(out_channels, out_channels))
            setattr(self, f"glou{i}", nn.Sequential(OrderedDict(GCN00=GloRe_Unit(out_channels, m))))
        self.f1 = Basconv(4 * out_channels, in_channels, kernel_size=1, padding=0)

    def forward(self, x):
        outs = []
        for i in range(4):
            t = getattr(self, f"conv{i}_1")(x)
            if i:
                t = getattr(self, f"conv{i}_2")(getattr(self, f"pool{i}")(t))
            t = getattr(self, f"glou{i}")(t)
            outs.append(t if i == 0 else nn.functional.interpolate(t, size=x.shape[2:], mode="bilinear", align_corners=True))
        return self.f1(torch.cat(outs, 1))


class _MGNet(nn.Module):
    POOLS = (2, 2, 2)

    def __init__(self, in_channels=1, num_classes=11, feature_scale=4, is_deconv=True, is_batchnorm=True):
        super().__init__()
        f = [int(c / feature_scale) for c in (64, 128, 256, 512, 1024)]
        cin = in_channels
        for i, k in enumerate(self.POOLS):
            setattr(self, f"conv{i + 1}", UnetConv(cin, f[i], is_batchnorm))
            setattr(self, f"maxpool{i + 1}", nn.MaxPool2d(kernel_size=k))
            cin = f[i]
        self.mgb = MGR_Module(f[2], f[3])
        self.center = UnetConv(f[2], f[3], is_batchnorm)
        for i in (3, 2, 1):
            setattr(self, f"up_concat{i}", (UnetUp4 if self.POOLS[i - 1] == 4 else UnetUp)(f[i], f[i - 1], is_deconv))
        self.final_1 = nn.Conv2d(f[0], num_classes, 1)
        for m in self.modules():       # kaiming-normal Conv2d weights, N(1, 0.02) BatchNorm weights / zero biases
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, a=0, mode="fan_in")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.normal_(m.weight, 1.0, 0.02)
                nn.init.constant_(m.bias, 0.0)

    def forward(self, x):
        skips = []
        for i in (1, 2, 3):
            x = getattr(self, f"conv{i}")(x)
            skips.append(x)
            x = getattr(self, f"maxpool{i}")(x)
        x = self.center(self.mgb(x))
        for i in (3, 2, 1):
            x = getattr(self, f"up_concat{i}")(x, skips[i - 1])
        return self.final_1(x)


class MGUNet(_MGNet):
    POOLS = (2, 4, 4)


class MGUNet_2(_MGNet):
    POOLS = (2, 2, 2)
