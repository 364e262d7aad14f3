"""GPU parity of the block-family drop-ins against the fixtures produced by the
reference's own classes (tools/gen_golden_blocks.py).  fp32 parity mode: outputs 2e-5 of their
scale, gradients 2e-3 rel of the tensor's max (as test_gpu_unet.py); bf16 production mode:
documented looser bounds."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.cases import bio_case, bio_grad_errors, bio_weights_match
from test_oracle_blocks import BLOCKS, load_block

pytestmark = pytest.mark.gpu


class Dropins:
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.MGUNet_2021 import (  # noqa: E402
        UnetConv, UnetUp, UnetUp4)
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.SD_Layer_Net.common import (  # noqa: E402
        Attention_block, conv_block, up_conv)


def close(got, ref, key, rel, floor=1e-4):
    ref = np.asarray(ref, np.float64)
    tol = rel * max(float(np.abs(ref).max()), floor)
    err = float(np.abs(np.asarray(got, np.float64) - ref).max())
    assert err <= tol, f"{key}: max err {err:.3e} > {tol:.3e}"


@pytest.mark.parametrize("name", list(BLOCKS))
def test_f32_block_matches_reference_fixture(golden_dir, name):
    z, m, xs = load_block(golden_dir, name, Dropins)
    m.set_compute_dtype("f32").cuda()
    xd = [x.cuda().requires_grad_(True) for x in xs]
    out = m(*xd)
    assert out.dtype == torch.float32 and tuple(out.shape) == z["out"].shape
    close(out.detach().cpu().numpy(), z["out"], "out", 2e-5, 1.0)
    (out * torch.from_numpy(z["r"]).cuda()).sum().backward()
    for i, x in enumerate(xd):
        close(x.grad.cpu().numpy(), z[f"gx{i}"], f"gx{i}", 2e-3)
    for k, p in m.named_parameters():
        close(p.grad.cpu().numpy(), z["g/" + k], k, 2e-3)
    sd = m.state_dict()
    for k in z.files:
        if k.startswith("b1/"):
            if "num_batches" in k:
                assert int(sd[k[3:]]) == int(z[k])
            else:
                close(sd[k[3:]].cpu().numpy(), z[k], k, 1e-4)
    m.eval()
    with torch.no_grad():
        close(m(*[x.cuda() for x in xs]).cpu().numpy(), z["out_eval"], "out_eval", 2e-5, 1.0)


@pytest.mark.parametrize("name", list(BLOCKS))
def test_bf16_block_is_close(golden_dir, name):
    z, m, xs = load_block(golden_dir, name, Dropins)
    m.set_compute_dtype("bf16").cuda()
    xd = [x.cuda().requires_grad_(True) for x in xs]
    out = m(*xd)
    ref = z["out"]
    err = np.abs(out.detach().cpu().numpy() - ref)
    assert err.max() < 0.06 * max(1.0, np.abs(ref).max()) and err.mean() < 0.01 * max(1.0, np.abs(ref).mean())
    (out * torch.from_numpy(z["r"]).cuda()).sum().backward()
    cos = []
    for k, p in m.named_parameters():
        a, b = p.grad.flatten().double().cpu(), torch.from_numpy(z["g/" + k]).flatten().double()
        if float(b.norm()) > 1e-6:
            cos.append(float(a @ b / (a.norm() * b.norm() + 1e-30)))
    assert min(cos) > 0.9, (name, cos)


NETS = [("attunet_c3_2x32x48", "AttU_Net", dict(channels=[4, 8, 16, 32, 64])), ("sd_unet_c2_1x32x32", "U_Net", {})]


@pytest.mark.parametrize("name,cls,kw", NETS)
def test_f32_network_matches_reference_fixture(golden_dir, name, cls, kw):
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.SD_Layer_Net import unet as U
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    seed, n, cin, ncls, h, w = (int(v) for v in z["meta"])
    m, x, t = bio_case(lambda ci, nc: getattr(U, cls)(ci, nc, compute_dtype="f32", **kw), seed, n, cin, ncls, h, w)
    assert bio_weights_match(z, m.state_dict())
    m.cuda()
    out = m(x.cuda())
    lg = out.detach().cpu().numpy()
    close(lg, z["logits"], "logits", 2e-5, 1.0)
    assert np.array_equal(lg.argmax(1), z["logits"].argmax(1))
    loss = F.cross_entropy(out, t.cuda())
    np.testing.assert_allclose(float(loss.detach()), float(z["loss"][0]), rtol=2e-5)
    loss.backward()
    assert bio_grad_errors(z, {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}, 2e-3) == []
    sd = m.state_dict()
    for k in z.files:
        if k.startswith("b1/") and "running" in k:
            close(sd[k[3:]].cpu().numpy(), z[k], k, 1e-4)
    m.eval()
    with torch.no_grad():
        close(m(x.cuda()).cpu().numpy(), z["logits_eval"], "logits_eval", 2e-5, 1.0)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, cin, 24, 32, device="cuda"))       # not divisible by 16: the reference raises too


def test_bf16_attunet_tracks_reference(golden_dir):
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.SD_Layer_Net import unet as U
    z = np.load(os.path.join(golden_dir, "attunet_c3_2x32x48.npz"))
    seed, n, cin, ncls, h, w = (int(v) for v in z["meta"])
    m, x, t = bio_case(lambda ci, nc: U.AttU_Net(ci, nc, channels=[4, 8, 16, 32, 64]), seed, n, cin, ncls, h, w)
    m.cuda()
    out = m(x.cuda())
    assert (out.argmax(1).cpu().numpy() == z["logits"].argmax(1)).mean() > 0.9
    loss = F.cross_entropy(out, t.cuda())
    assert abs(float(loss.detach()) - float(z["loss"][0])) < 5e-2
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.mark.parametrize("dt,tol", [("f32", 1e-5), ("bf16", 3e-2)])
@pytest.mark.parametrize("c", [4, 8, 24, 64, 512])
def test_gate_product_backward_all_channel_paths(dt, tol, c):
    """oct_gate_bwd: vector kernel (c/8 a power of two: segmented shuffle reduction, up to a full wave per
    pixel) and scalar fallback, against the closed form dx = dout*p, dp = sum_c dout*x."""
    from retinal_oct_image_segmentation_via_deep_learning_amd import ops
    torch.manual_seed(c)
    n, h, w = 2, 9, 13
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    x = torch.randn(n, h, w, c, device="cuda").to(tdt).requires_grad_(True)
    p = torch.rand(n, h, w, 1, device="cuda").to(tdt).requires_grad_(True)
    out = ops.Gate.apply(dt, x, p)
    r = torch.randn_like(out)
    (out.float() * r.float()).sum().backward()
    assert (out.float() - x.float() * p.float()).abs().max() < tol
    dx_ref, dp_ref = r.float() * p.float(), (r.float() * x.float()).sum(-1, keepdim=True)
    assert (x.grad.float() - dx_ref).abs().max() < tol
    assert ((p.grad.float() - dp_ref).abs().max() / dp_ref.abs().max()) < tol
