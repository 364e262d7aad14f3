"""CPU: pin oracle/ref_cpu.py against the fixtures produced from the reference itself
(tools/gen_golden.py imports /root/reference; fixtures are committed under tests/golden/)."""
import os

import numpy as np
import pytest

from oracle import ref_cpu

CASES = ["unet_c8_f4_2x32x32", "unet_c2_f4_1x48x64_dice", "unet_in3_c3_f4_2x32x48",
         "unet_c8_f8_1x32x64_light"]


def load_case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    w0 = {k[3:]: z[k] for k in z.files if k.startswith("w0/")}
    return z, w0


@pytest.mark.parametrize("name", CASES)
def test_unet_forward_loss_grads(golden_dir, name):
    z, w0 = load_case(golden_dir, name)
    w_ce, w_dice, lr, mom, eps = z["hyper"]
    net = ref_cpu.OracleUNet(w0)
    probs, (loss, ce, dice), grads = net.loss_and_grads(z["x"], z["target"], w_ce, w_dice, eps)
    np.testing.assert_allclose(net.logits, z["logits"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(probs, z["probs"], rtol=1e-4, atol=1e-6)
    am = probs.argmax(1)
    # bit-identical class maps wherever the reference's own top-2 margin exceeds fp32 noise
    top2 = np.sort(z["probs"], axis=1)[:, -2:]
    safe = (top2[:, 1] - top2[:, 0]) > 1e-5
    assert np.array_equal(am[safe], z["argmax"][safe])
    assert (am != z["argmax"]).sum() <= 2
    np.testing.assert_allclose([loss, ce, dice], z["loss"], rtol=2e-6, atol=1e-7)
    n = 0
    for k in z.files:
        if k.startswith("g0/"):
            ref = z[k]
            tol = 1e-4 * max(1e-3, float(np.abs(ref).max()))
            np.testing.assert_allclose(grads[k[3:]], ref, rtol=1e-3, atol=tol, err_msg=k)
            n += 1
    assert n > 0
    for k in z.files:
        if k.startswith("b1/"):
            np.testing.assert_allclose(net.s[k[3:]], z[k], rtol=1e-5, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("name", CASES[:2])
def test_unet_eval_and_sgd_trajectory(golden_dir, name):
    z, w0 = load_case(golden_dir, name)
    w_ce, w_dice, lr, mom, eps = z["hyper"]
    steps = int(z["meta"][6])
    net = ref_cpu.OracleUNet(w0)
    losses = []
    for i in range(steps):
        _, (loss, _, _), grads = net.loss_and_grads(z["x"], z["target"], w_ce, w_dice, eps)
        if i == 0:
            pe = net.forward(z["x"], train=False)
            np.testing.assert_allclose(pe, z["probs_eval"], rtol=1e-4, atol=1e-6)
        losses.append(loss)
        net.sgd_step(grads, lr, mom)
    np.testing.assert_allclose(losses, z["traj_loss"], rtol=1e-5)
    for k in z.files:
        if k.startswith("wN/") and "num_batches" not in k:
            ref = z[k]
            np.testing.assert_allclose(net.s[k[3:]], ref, rtol=1e-3,
                                       atol=1e-4 * max(1e-2, float(np.abs(ref).max())), err_msg=k)
        elif k.startswith("wN/"):
            assert int(net.s[k[3:]]) == int(z[k])


def test_unet_rejects_non_multiple_of_16(golden_dir):
    z = np.load(os.path.join(golden_dir, "api.npz"))
    assert "Sizes of tensors must match" in str(z["negative_msg"])
    _, w0 = load_case(golden_dir, CASES[1])
    net = ref_cpu.OracleUNet(w0)
    with pytest.raises(RuntimeError, match="Sizes of tensors must match"):
        net.forward(np.zeros((1, 1, 62, 96)))


def test_metrics_known_answers(golden_dir):
    z = np.load(os.path.join(golden_dir, "metrics.npz"))
    cases = sorted({k.split("/")[1] for k in z.files if k.startswith("in/")})
    assert len(cases) >= 8
    for c in cases:
        yt, yp = z[f"in/{c}/y_true"], z[f"in/{c}/y_pred"]
        for fname, fn in ref_cpu.METRIC_FUNCS.items():
            key = f"out/{c}/{fname}"
            if key not in z.files:
                continue
            got = fn(yt, yp)
            np.testing.assert_allclose(got, z[key], rtol=1e-6 if yt.dtype == np.float32 else 1e-12,
                                       atol=0, err_msg=key)


def test_metrics_seeded_full_size(golden_dir):
    z = np.load(os.path.join(golden_dir, "metrics.npz"))
    rng = np.random.default_rng(1234)
    a = (rng.random((32, 512, 1024)) < 0.3).astype(np.uint8)
    b = (rng.random((32, 512, 1024)) < 0.3).astype(np.uint8)
    c = ref_cpu.confusion_sums(a, b)
    assert [int(c["tp"]), int(c["t"]), int(c["p"]), c["n"]] == z["seeded_counts"].tolist()
    for fname, fn in ref_cpu.METRIC_FUNCS.items():
        np.testing.assert_allclose(fn(a, b), z[f"out/seeded_32x512x1024/{fname}"], rtol=1e-13)


BIO_CASES = ["bionet_unet_c2_2x16x24", "bionet_unet_in3_c4_1x32x16"]


def _bio_case(seed, n, cin, ncls, h, w):
    """Seeded BioNet-UNet case on the torch restatement, float64."""
    from oracle.cases import bio_case
    from oracle.torch_unet import TorchBioUNet
    m, x, t = bio_case(TorchBioUNet, seed, n, cin, ncls, h, w)
    return m.double(), x.double(), t


@pytest.mark.parametrize("name", BIO_CASES)
def test_bionet_unet_oracle_matches_reference_fixture(golden_dir, name):
    """OracleBioUNet and TorchBioUNet against vectors produced by the reference's own class
    (tools/gen_golden_bionet.py): weights rebuilt from the seed must carry the reference's
    checksums, then logits / loss / gradients / running stats / eval logits must agree."""
    import torch
    from oracle.cases import bio_grad_errors, bio_weights_match
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    seed, n, cin, ncls, h, w = (int(v) for v in z["meta"])
    w_ce, w_dice, eps = (float(v) for v in z["hyper"])
    from oracle.cases import bio_case
    from oracle.torch_unet import TorchBioUNet
    m, x, t = bio_case(TorchBioUNet, seed, n, cin, ncls, h, w)
    assert bio_weights_match(z, m.state_dict())
    assert np.array_equal(x.numpy(), z["x"]) and np.array_equal(t.numpy(), z["target"])
    o = ref_cpu.OracleBioUNet({k: v.numpy() for k, v in m.state_dict().items()})
    logits, (loss, ce, dice), g = o.loss_and_grads(z["x"], z["target"], w_ce, w_dice, eps)
    np.testing.assert_allclose(logits, z["logits"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose([loss, ce, dice], z["loss"], rtol=1e-10)
    assert bio_grad_errors(z, g, 1e-7) == []
    for k in z.files:
        if k.startswith("b1/") and "running" in k:
            np.testing.assert_allclose(o.s[k[3:]], z[k], rtol=1e-10, atol=1e-12, err_msg=k)
    np.testing.assert_allclose(o.forward(z["x"], train=False), z["logits_eval"], rtol=1e-9, atol=1e-9)
    # the torch restatement (CPU baseline / bf16 comparisons) is the same function
    md = m.double()
    np.testing.assert_allclose(md(x.double()).detach().numpy(), z["logits"], rtol=1e-9, atol=1e-9)
    api = np.load(os.path.join(golden_dir, "bionet_api.npz"))
    assert int(api["n_params"]) == 7701890 and "Sizes of tensors must match" in str(api["negative_msg"])


def test_bionet_unet_oracle_matches_torch_restatement():
    import torch
    import torch.nn.functional as F
    m, x, t = _bio_case(11, 2, 1, 2, 16, 24)
    state0 = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    logits = m(x)
    F.cross_entropy(logits, t).backward()
    o = ref_cpu.OracleBioUNet(state0)
    ol, (loss, ce, dice), g = o.loss_and_grads(x.numpy(), t.numpy())
    np.testing.assert_allclose(ol, logits.detach().numpy(), rtol=1e-9, atol=1e-9)
    assert abs(ce - F.cross_entropy(logits, t).item()) < 1e-10
    for k, p in m.named_parameters():
        ref = p.grad.numpy()
        tol = 1e-9 * max(1.0, np.abs(ref).max())
        if k.endswith((".0.bias", ".3.bias")):          # conv bias in front of BN: zero up to rounding
            assert np.abs(g[k]).max() < 1e-9 and np.abs(ref).max() < 1e-9
        else:
            np.testing.assert_allclose(g[k], ref, rtol=1e-7, atol=tol, err_msg=k)
    for k, v in m.state_dict().items():                # running statistics (conv bias enters the mean)
        if "running" in k:
            np.testing.assert_allclose(o.s[k], v.numpy(), rtol=1e-10, atol=1e-12, err_msg=k)
    m.eval()
    np.testing.assert_allclose(o.forward(x.numpy(), train=False), m(x).detach().numpy(), rtol=1e-9, atol=1e-9)
    with pytest.raises(RuntimeError, match="Sizes of tensors must match"):
        o.forward(np.zeros((1, 1, 20, 16)))


# ---- the headline width: UNet(1, 8, init_features=32), fixture made from the reference by tools/gen_golden_wide.py ----
WIDE = "unet_c8_f32_2x64x128_wide"


def wide_case(golden_dir):
    """(fixture, state_dict as numpy, x, target): the weights are rebuilt from the seed with the drop-in's own class
    (same construction order as the reference => same tensors) and verified against the reference's checksums."""
    import torch
    from oracle.cases import ynet_case
    from retinal_oct_image_segmentation_via_deep_learning_amd import UNet
    z = np.load(os.path.join(golden_dir, WIDE + ".npz"))
    in_ch, ncls, feat, b, h, w = (int(v) for v in z["meta"])
    model, x, t = ynet_case(UNet, int(z["seed"]), in_ch, ncls, feat, (b, h, w))
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in z["keys"]]
    for k, v in sd.items():
        v = v.double()
        np.testing.assert_allclose([float(v.sum()), float(v.abs().sum())], z["wsum/" + k], rtol=1e-12, atol=1e-12, err_msg=k)
    assert np.array_equal(x.numpy(), z["x"]) and np.array_equal(t.numpy(), z["target"])
    return z, {k: v.detach().numpy().copy() for k, v in sd.items()}, x, t, model


def wide_grad_errors(z, grads, rel):
    """failures of a {name: ndarray} gradient set against the fixture's full tensors / (norms, strided sample)"""
    from oracle.cases import grad_summary
    bad = []
    for key in z.files:
        kind, _, name = key.partition("/")
        if kind == "g":
            ref, got = z[key].astype(np.float64), np.asarray(grads[name], np.float64)
        elif kind == "gs":
            ref, got = z[key], grad_summary(grads[name])[1]
        elif kind == "gn":
            ref, got = z[key][:1], grad_summary(grads[name])[0][:1]
        else:
            continue
        tol = rel * max(float(np.abs(ref).max()), 1e-5)
        err = float(np.abs(got - ref).max())
        if err > tol:
            bad.append(f"{key}: max err {err:.3e} > {tol:.3e}")
    return bad


def test_wide_unet_oracle_matches_reference_fixture(golden_dir):
    z, state, x, t, _ = wide_case(golden_dir)
    net = ref_cpu.OracleUNet(state)
    probs, (loss, _, _), grads = net.loss_and_grads(x.numpy(), t.numpy())
    # the fixture is the reference module run in float64 on the same fp32 weights: agreement to round-off
    np.testing.assert_allclose(probs, z["probs"], rtol=1e-9, atol=1e-12)
    assert np.array_equal(probs.argmax(1), z["argmax"])
    np.testing.assert_allclose(loss, float(z["loss"]), rtol=1e-12)
    assert not wide_grad_errors(z, grads, 1e-8)
    for k in z.files:
        if k.startswith("b1/"):
            np.testing.assert_allclose(net.s[k[3:]], z[k], rtol=1e-10, atol=1e-12, err_msg=k)


def test_bf16_storage_mode_rounds_and_stays_close(golden_dir):
    """storage="bf16" is the same restatement with roundings at the HIP path's storage points: every stored tensor
    is representable in bf16, and the result stays within bf16 noise of the wide model (sanity of the mode itself;
    the kernels are compared against it in tests/test_gpu_unet.py)."""
    z, w0 = load_case(golden_dir, CASES[0])
    wide = ref_cpu.OracleUNet(w0)
    pw, (lw, _, _), gw = wide.loss_and_grads(z["x"], z["target"])
    net = ref_cpu.OracleUNet(w0, storage="bf16")
    p, (l, _, _), g = net.loss_and_grads(z["x"], z["target"])
    for lvl, c in net._cache[0].items():
        for (_, _, _, xin, _, _, _) in c:
            assert np.array_equal(ref_cpu.round_bf16(xin), xin), lvl      # staged activations are bf16 values
    assert np.abs(p - pw).max() < 5e-2 and abs(l - lw) < 2e-2 * abs(lw)
    cos = [float(g[k].ravel() @ gw[k].ravel() / (np.linalg.norm(g[k]) * np.linalg.norm(gw[k]) + 1e-30)) for k in g if g[k].size >= 16]
    assert np.mean(cos) > 0.85        # same regime the bf16 kernels showed against fp32 in round 1 (mask flips)
    x = np.array([1.0, 1.00390625, 1.005859375, -3.14159, 1e-40, 65504.0])
    assert np.array_equal(ref_cpu.round_bf16(x)[:3], [1.0, 1.0, 1.0078125])    # ties to even, then above the tie
