"""GPU parity of the BioNet_2020 `UNet` drop-in (BASELINE cfg1) against the
fixtures made from the reference's own class (tests/golden/bionet_unet_*.npz,
tools/gen_golden_bionet.py) and against oracle/ref_cpu.OracleBioUNet on further seeded cases.
Tolerances as in test_gpu_unet.py (fp32 parity mode: logits 2e-5 of their scale, arg-max
identical, loss 2e-5 rel, gradients 2e-3 rel of the tensor's max)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu
from test_oracle import _bio_case

pytestmark = pytest.mark.gpu


def build(m_ref, cin, ncls, dtype):
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.BioNet_2020 import UNet
    model = UNet(cin, ncls, compute_dtype=dtype)
    model.load_state_dict({k: v.float() for k, v in m_ref.state_dict().items()}, strict=True)
    return model.cuda().train()


def gclose(got, ref, key, rel):
    ref = np.asarray(ref, np.float64)
    tol = rel * max(float(np.abs(ref).max()), 1e-4)
    err = float(np.abs(np.asarray(got, np.float64) - ref).max())
    assert err <= tol, f"{key}: max err {err:.3e} > {tol:.3e}"


@pytest.mark.parametrize("name", ["bionet_unet_c2_2x16x24", "bionet_unet_in3_c4_1x32x16"])
def test_f32_matches_reference_fixture(golden_dir, name):
    from oracle.cases import bio_case, bio_grad_errors, bio_weights_match
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.BioNet_2020 import UNet
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    seed, n, cin, ncls, h, w = (int(v) for v in z["meta"])
    w_ce, w_dice, eps = (float(v) for v in z["hyper"])
    model, x, t = bio_case(lambda a, b: UNet(a, b, compute_dtype="f32"), seed, n, cin, ncls, h, w)
    assert bio_weights_match(z, model.state_dict())       # seeded default init == the reference's
    model.cuda()
    xd, td = x.cuda(), t.cuda()
    out = model(xd)                                       # autograd path, raw logits (BioNet_2020.py:75)
    lg = out.detach().cpu().numpy()
    scale = max(1.0, float(np.abs(z["logits"]).max()))
    assert np.abs(lg - z["logits"]).max() < 2e-5 * scale
    assert np.array_equal(lg.argmax(1), z["logits"].argmax(1))
    F.cross_entropy(out, td).backward()
    if w_dice == 0.0:
        assert bio_grad_errors(z, {k: p.grad.cpu().numpy() for k, p in model.named_parameters()}, 2e-3) == []
    # fused loss head + backward; rewind the BN buffers the first pass advanced
    model2, _, _ = bio_case(lambda a, b: UNet(a, b, compute_dtype="f32"), seed, n, cin, ncls, h, w)
    model2.cuda()
    lv = model2.forward_backward(xd, td, w_ce, w_dice, eps)
    np.testing.assert_allclose(lv.cpu().numpy(), z["loss"], rtol=2e-5, atol=1e-6)
    assert bio_grad_errors(z, {k: p.grad.cpu().numpy() for k, p in model2.named_parameters()}, 2e-3) == []
    sd = model2.state_dict()
    for k in z.files:
        if k.startswith("b1/"):
            if "num_batches" in k:
                assert int(sd[k[3:]]) == int(z[k])
            else:
                gclose(sd[k[3:]].cpu().numpy(), z[k], k, 1e-4)
    model2.eval()
    le = model2(xd).detach().cpu().numpy()   # like the reference module: requires_grad without no_grad()
    assert np.abs(le - z["logits_eval"]).max() < 2e-5 * max(1.0, float(np.abs(z["logits_eval"]).max()))
    msg = str(np.load(os.path.join(golden_dir, "bionet_api.npz"))["negative_msg"])
    with pytest.raises(RuntimeError) as ei:
        model2(torch.zeros(1, cin, 20, 16, device="cuda"))
    assert "Sizes of tensors must match except in dimension 1" in msg and \
        "Sizes of tensors must match except in dimension 1" in str(ei.value)


def test_state_dict_and_seeded_init_match_restatement():
    from oracle.torch_unet import TorchBioUNet
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.BioNet_2020 import UNet
    torch.manual_seed(5)
    a = TorchBioUNet(1, 2).state_dict()
    torch.manual_seed(5)
    b = UNet(1, 2).state_dict()
    assert list(a) == list(b)
    for k in a:
        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k


# seeds chosen (CPU, float64) so that no BN output lies within 1e-5 of zero: a ReLU mask flip from
# legitimate fp32 rounding would otherwise move the deepest gradients by ~1 %
@pytest.mark.parametrize("cfg", [(127, 2, 1, 2, 16, 24), (126, 1, 1, 3, 24, 40)])
def test_f32_fused_step_matches_oracle(cfg):
    seed, n, cin, ncls, h, w = cfg
    m, x, t = _bio_case(seed, n, cin, ncls, h, w)
    state0 = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    o = ref_cpu.OracleBioUNet(state0)
    ol, (loss, ce, dice), g = o.loss_and_grads(x.numpy(), t.numpy(), 1.0, 0.5)
    model = build(m, cin, ncls, "f32")
    lv = model.forward_backward(x.float().cuda(), t.cuda(), 1.0, 0.5)
    torch.cuda.synchronize()
    np.testing.assert_allclose(lv.cpu().numpy(), [loss, ce, dice], rtol=2e-5, atol=1e-6)
    for k, p in model.named_parameters():
        if k.endswith((".0.bias", ".3.bias")):
            assert float(p.grad.abs().max()) == 0.0, k      # bias in front of BN: exactly zero here
        else:
            gclose(p.grad.cpu().numpy(), g[k], k, 2e-3)
    sd = model.state_dict()
    for k, v in o.s.items():
        if "running" in k:
            gclose(sd[k].cpu().numpy(), v, k, 1e-4)
        if "num_batches" in k:
            assert int(sd[k]) == int(v)
    # eval mode uses the running statistics (conv bias folded into the shift)
    model.eval()
    lg = model(x.float().cuda()).detach().cpu().numpy()
    ref = o.forward(x.numpy(), train=False)
    assert np.abs(lg - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_f32_autograd_from_logits_matches_oracle():
    seed, n, cin, ncls, h, w = 121, 2, 1, 2, 16, 16
    m, x, t = _bio_case(seed, n, cin, ncls, h, w)
    state0 = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    o = ref_cpu.OracleBioUNet(state0)
    ol, _, g = o.loss_and_grads(x.numpy(), t.numpy())
    model = build(m, cin, ncls, "f32")
    out = model(x.float().cuda())                       # raw logits, as BioNet_2020.py:75
    assert out.shape == (n, ncls, h, w) and out.requires_grad
    assert np.abs(out.detach().cpu().numpy() - ol).max() < 2e-5 * max(1.0, np.abs(ol).max())
    F.cross_entropy(out, t.cuda()).backward()
    for k, p in model.named_parameters():
        if not k.endswith((".0.bias", ".3.bias")):
            gclose(p.grad.cpu().numpy(), g[k], k, 2e-3)
    with pytest.raises(RuntimeError, match="Sizes of tensors must match"):
        model(torch.zeros(1, cin, 20, 16, device="cuda"))
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, cin + 1, 16, 16, device="cuda"))


def test_bf16_production_mode_tracks_f32():
    seed, n, cin, ncls, h, w = 14, 2, 1, 2, 64, 64
    m, x, t = _bio_case(seed, n, cin, ncls, h, w)
    m = m.float()
    ref = m(x.float())
    F.cross_entropy(ref, t).backward()
    model = build(m, cin, ncls, "bf16")
    lv = model.forward_backward(x.float().cuda(), t.cuda())
    assert abs(float(lv[0]) - float(F.cross_entropy(ref, t).detach())) < 3e-2
    pred = model.predict(x.float().cuda()).cpu()
    assert (pred == ref.argmax(1)).float().mean() > 0.97
    cos = []
    for k, p in m.named_parameters():
        if p.dim() == 4:
            a, b = p.grad.flatten().double(), dict(model.named_parameters())[k].grad.flatten().double().cpu()
            cos.append(float(a @ b / (a.norm() * b.norm() + 1e-30)))
    assert min(cos) > 0.7 and np.mean(cos) > 0.9, cos
