"""The torch restatements of the block families (oracle/torch_blocks.py) against the fixtures made
from the reference's own classes (tools/gen_golden_blocks.py), and host logic of the drop-ins."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_blocks as TB
from oracle.cases import bio_case, bio_grad_errors, bio_weights_match

BLOCKS = {
    "blk_unetconv_bn": lambda m: m.UnetConv(3, 8, True),
    "blk_unetconv_nobn": lambda m: m.UnetConv(1, 8, False),
    "blk_unetup_deconv": lambda m: m.UnetUp(16, 8, True),
    "blk_unetup_bilinear": lambda m: m.UnetUp(16, 8, False),
    "blk_unetup4_deconv": lambda m: m.UnetUp4(16, 8, True),
    "blk_unetup4_bilinear": lambda m: m.UnetUp4(16, 8, False),
    "blk_conv_block": lambda m: m.conv_block(3, 8),
    "blk_up_conv": lambda m: m.up_conv(8, 4),
    "blk_attention": lambda m: m.Attention_block(8, 8, 4),
    # act != nn.ReLU and Dropout2d(p > 0) under the keep masks the fixture recorded (common.py:7,13,17,29,34)
    "blk_conv_block_drop": lambda m: m.conv_block(3, 8, drop_rate=0.2),
    "blk_up_conv_drop": lambda m: m.up_conv(8, 4, drop_rate=0.2),
    "blk_conv_block_leaky": lambda m: m.conv_block(3, 8, act=torch.nn.LeakyReLU),
    "blk_up_conv_tanh_drop": lambda m: m.up_conv(8, 4, act=torch.nn.Tanh, drop_rate=0.2),
}


def load_block(golden_dir, name, module, **kw):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    m = BLOCKS[name](module)
    m.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w0/")}, strict=True)
    xs = [torch.from_numpy(z[f"x{i}"]) for i in range(2) if f"x{i}" in z.files]
    # Dropout2d fixtures: the module family replays the recorded keep masks, one per Dropout2d application in forward order
    module.set_dropout_masks([torch.from_numpy(z[k]) for k in sorted(k for k in z.files if k.startswith("mask"))])
    return z, m.train(), xs


@pytest.mark.parametrize("name", list(BLOCKS))
def test_block_restatement_matches_reference_fixture(golden_dir, name):
    z, m, xs = load_block(golden_dir, name, TB)
    m = m.double()
    xd = [x.double().requires_grad_(True) for x in xs]
    out = m(*xd)
    np.testing.assert_allclose(out.detach().numpy(), z["out"], rtol=1e-9, atol=1e-10)
    (out * torch.from_numpy(z["r"]).double()).sum().backward()
    for i, x in enumerate(xd):
        np.testing.assert_allclose(x.grad.numpy(), z[f"gx{i}"], rtol=1e-7, atol=1e-10)
    for k, p in m.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), z["g/" + k], rtol=1e-7, atol=1e-9, err_msg=k)
    for k, v in m.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.numpy(), z["b1/" + k], rtol=1e-10, atol=1e-12)
    m.eval()
    np.testing.assert_allclose(m(*[x.double() for x in xs]).detach().numpy(), z["out_eval"], rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("name,cls", [("attunet_c3_2x32x48", lambda ci, nc: TB.AttU_Net(ci, nc, channels=[4, 8, 16, 32, 64])),
                                      ("attunet4_c3_2x24x40", lambda ci, nc: TB.AttU_Net4(ci, nc, channels=[4, 8, 16, 32])),
                                      ("sd_unet_c2_1x32x32", lambda ci, nc: TB.U_Net(ci, nc))])
def test_network_restatement_matches_reference_fixture(golden_dir, name, cls):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    seed, n, cin, ncls, h, w = (int(v) for v in z["meta"])
    m, x, t = bio_case(cls, seed, n, cin, ncls, h, w)
    assert bio_weights_match(z, m.state_dict())        # same names, same seeded init as the reference class
    assert np.array_equal(x.numpy(), z["x"])
    m = m.double()
    lg = m(x.double())
    np.testing.assert_allclose(lg.detach().numpy(), z["logits"], rtol=1e-8, atol=1e-9)
    F.cross_entropy(lg, t).backward()
    assert bio_grad_errors(z, {k: p.grad.numpy() for k, p in m.named_parameters()}, 1e-6) == []
    m.eval()
    np.testing.assert_allclose(m(x.double()).detach().numpy(), z["logits_eval"], rtol=1e-8, atol=1e-9)


def test_init_weights_matches_reference_under_seed(golden_dir):
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment import MGUNet_2021 as M
    z = np.load(os.path.join(golden_dir, "mgunet_init.npz"))
    for kind in ("normal", "xavier", "kaiming"):
        torch.manual_seed(77)
        m = M.UnetUp(16, 8, True)
        for c in m.modules():
            if isinstance(c, (torch.nn.Conv2d, torch.nn.BatchNorm2d)):
                M.init_weights(c, init_type=kind)
        for k, v in m.state_dict().items():
            got = np.array([float(v.double().sum()), float(v.double().abs().sum())])
            np.testing.assert_allclose(got, z[f"{kind}/{k}"], rtol=1e-12, atol=1e-12, err_msg=f"{kind}/{k}")
    with pytest.raises(NotImplementedError) as ei:
        M.init_weights(torch.nn.Conv2d(1, 1, 1), "nope")
    assert str(ei.value) == str(z["bad_type_msg"])


def test_dropin_state_dicts_and_ctor_rules(golden_dir):
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment import MGUNet_2021 as M
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.SD_Layer_Net import common as Cm
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.SD_Layer_Net import unet as U

    class Both:   # namespace with every drop-in class under the reference's names
        UnetConv, UnetUp, UnetUp4 = M.UnetConv, M.UnetUp, M.UnetUp4
        conv_block, up_conv, Attention_block = Cm.conv_block, Cm.up_conv, Cm.Attention_block
    for name in BLOCKS:
        z = np.load(os.path.join(golden_dir, name + ".npz"))
        m = BLOCKS[name](Both)
        ref_keys = [k[3:] for k in z.files if k.startswith("w0/")]
        assert list(m.state_dict().keys()) == ref_keys, name
        m.load_state_dict({k: torch.from_numpy(z["w0/" + k]) for k in ref_keys}, strict=True)
    z = np.load(os.path.join(golden_dir, "attunet_c3_2x32x48.npz"))
    seed, n, cin, ncls, h, w = (int(v) for v in z["meta"])
    m, _, _ = bio_case(lambda ci, nc: U.AttU_Net(ci, nc, channels=[4, 8, 16, 32, 64]), seed, n, cin, ncls, h, w)
    assert bio_weights_match(z, m.state_dict())
    api = np.load(os.path.join(golden_dir, "sd_api.npz"))
    assert sum(p.numel() for p in U.AttU_Net(1, 3).parameters()) == int(api["attunet_default_params"])
    assert sum(p.numel() for p in U.U_Net(1, 2).parameters()) == int(api["unet_default_params"])
    assert "F_g" in str(api["attunet_ctor_msg"])        # the reference's ctor bug the drop-in papers over
    Cm.Attention_block(channels_g=8, channels_x=8, F_int=4)
    Cm.Attention_block(F_g=8, F_l=8, F_int=4)
    Cm.conv_block(3, 8, act=torch.nn.LeakyReLU, drop_rate=0.3)      # elementwise activations and Dropout2d: the general schedule
    with pytest.raises(NotImplementedError):
        Cm.conv_block(3, 8, act=torch.nn.Softmax2d)                  # looks at a dimension: refused
    with pytest.raises(TypeError):
        Cm.up_conv(8, 4, act=torch.relu)                             # the reference calls act(): a module class
    with pytest.raises(L_error()):
        M.UnetConv(1, 4)(torch.zeros(1, 1, 8, 8))        # CPU tensor: no fallback


def L_error():
    from retinal_oct_image_segmentation_via_deep_learning_amd import _lib
    return _lib.OctError
