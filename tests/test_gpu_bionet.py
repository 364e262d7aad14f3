"""GPU parity of the BioNet_2020 `UNet` drop-in (BASELINE cfg1) against oracle/ref_cpu.OracleBioUNet.

PARITY UNPINNED for this topology: the reference file imports torchvision (absent here), so no
fixture could be generated from it; the oracle's wiring is cross-checked against an independent
torch.nn restatement in tests/test_oracle.py, and every primitive is pinned by the YNet fixtures.
Tolerances as in test_gpu_unet.py (fp32 parity mode: logits 2e-5 abs of their scale, gradients
2e-3 rel of the tensor's max)."""
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


@pytest.mark.parametrize("cfg", [(11, 2, 1, 2, 16, 24), (12, 1, 3, 4, 32, 16)])
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
    lg = model(x.float().cuda()).cpu().numpy()
    ref = o.forward(x.numpy(), train=False)
    assert np.abs(lg - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_f32_autograd_from_logits_matches_oracle():
    seed, n, cin, ncls, h, w = 13, 2, 1, 2, 16, 16
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
