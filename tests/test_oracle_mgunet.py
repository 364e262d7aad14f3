"""MGU-Net: the torch restatement (oracle/torch_blocks.py) against the fixtures made from the reference's own classes
(tools/gen_golden_mgunet.py: MGUNet_2021.py:29-39,110-309), and the host logic of the drop-in (constructor, state_dict keys,
seeded initialisation, API facts).  CPU only."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_blocks as TB
from oracle.cases import bio_case, bio_grad_errors, bio_weights_match

MG_BLOCKS = {
    "blk_basconv1x1": lambda m: m.Basconv(8, 4, kernel_size=1, padding=0),
    "blk_glore": lambda m: m.GloRe_Unit(8, 4),
    "blk_mgr": lambda m: m.MGR_Module(8, 16),
}
MG_NETS = [("mgunet2_c3_2x48x64", "MGUNet_2"), ("mgunet_c2_2x160x192", "MGUNet")]


def load_mg_block(golden_dir, name, module):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    m = MG_BLOCKS[name](module)
    m.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w0/")}, strict=True)
    return z, m.train(), [torch.from_numpy(z["x0"])]


def logits_close(z, key, got, rel, floor=1.0):
    """full tensor, or the compact record (every 97th element, sum / abs-sum, arg-max map) of the large case"""
    got = np.asarray(got, np.float64)
    if key in z.files:
        ref = z[key]
        assert np.abs(got - ref).max() <= rel * max(float(np.abs(ref).max()), floor), key
        return ref.argmax(1)
    ref = z[key + "_sample"]
    assert np.abs(got.reshape(-1)[::97] - ref).max() <= rel * max(float(np.abs(ref).max()), floor), key
    sums = np.array([got.sum(), np.abs(got).sum()])
    assert np.allclose(sums, z[key + "_sums"], rtol=max(rel, 1e-9) * 10, atol=rel * got.size ** 0.5), key
    return z[key + "_argmax"].astype(np.int64)


@pytest.mark.parametrize("name", list(MG_BLOCKS))
def test_block_restatement_matches_reference_fixture(golden_dir, name):
    z, m, xs = load_mg_block(golden_dir, name, TB)
    m = m.double()
    xd = [x.double().requires_grad_(True) for x in xs]
    out = m(*xd)
    np.testing.assert_allclose(out.detach().numpy(), z["out"], rtol=1e-9, atol=1e-10)
    (out * torch.from_numpy(z["r"]).double()).sum().backward()
    np.testing.assert_allclose(xd[0].grad.numpy(), z["gx0"], rtol=1e-7, atol=1e-10)
    for k, p in m.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), z["g/" + k], rtol=1e-7, atol=1e-9, err_msg=k)
    for k, v in m.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.numpy(), z["b1/" + k], rtol=1e-10, atol=1e-12)
    m.eval()
    np.testing.assert_allclose(m(xs[0].double()).detach().numpy(), z["out_eval"], rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("name,cls", MG_NETS)
def test_network_restatement_matches_reference_fixture(golden_dir, name, cls):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    seed, n, cin, ncls, h, w = (int(v) for v in z["meta"])
    m, x, t = bio_case(lambda ci, nc: getattr(TB, cls)(ci, nc, feature_scale=16), seed, n, cin, ncls, h, w)
    assert bio_weights_match(z, m.state_dict())        # same names, same seeded init as the reference class
    assert np.array_equal(x.numpy(), z["x"])
    m = m.double()
    lg = m(x.double())
    assert np.array_equal(lg.detach().numpy().argmax(1), logits_close(z, "logits", lg.detach().numpy(), 1e-8))
    loss = F.cross_entropy(lg, t)
    np.testing.assert_allclose(loss.item(), float(z["loss"][0]), rtol=1e-10)
    loss.backward()
    assert bio_grad_errors(z, {k: p.grad.numpy() for k, p in m.named_parameters()}, 1e-6) == []
    m.eval()
    logits_close(z, "logits_eval", m(x.double()).detach().numpy(), 1e-8)


@pytest.mark.parametrize("name,cls", MG_NETS)
def test_dropin_has_the_reference_keys_and_seeded_init(golden_dir, name, cls):
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment import MGUNet_2021 as M
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    seed, n, cin, ncls, h, w = (int(v) for v in z["meta"])
    m, _, _ = bio_case(lambda ci, nc: getattr(M, cls)(ci, nc, feature_scale=16), seed, n, cin, ncls, h, w)
    assert bio_weights_match(z, m.state_dict())


def test_api_facts(golden_dir):
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment import MGUNet_2021 as M
    z = np.load(os.path.join(golden_dir, "mgunet_api.npz"))
    assert sum(p.numel() for p in M.MGUNet_2().parameters()) == int(z["mgunet2_default_params"])
    assert sum(p.numel() for p in M.MGUNet().parameters()) == int(z["mgunet_default_params"])
    assert list(M.MGUNet_2(1, 3, feature_scale=16).state_dict().keys()) == [str(k) for k in z["mgunet2_keys"]]
    assert "Output size is too small" in str(z["small_msg"]) and "Expected more than 1 value per channel" in str(z["single_msg"])
    with pytest.raises(NotImplementedError):
        M.Basconv(4, 4, kernel_size=5, padding=2)
