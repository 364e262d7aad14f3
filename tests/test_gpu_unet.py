"""GPU parity of the whole U-Net path against the fixtures generated from the reference
(tests/golden/, tools/gen_golden.py) -- through the drop-in module API and the C ABI underneath.

Tolerances: fp32 parity mode -- probabilities 2e-5 abs, arg-max maps identical wherever the
reference's own top-2 margin exceeds 1e-5, loss 1e-5 rel, gradients 1e-3 rel of the tensor's
max.  bf16 production mode -- documented looser bounds (bf16 has 8 mantissa bits)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = ["unet_c8_f4_2x32x32", "unet_c2_f4_1x48x64_dice", "unet_in3_c3_f4_2x32x48", "unet_c8_f8_1x32x64_light"]


def load(golden_dir, name, dtype):
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Lesions_Segment.YNet_2022 import UNet
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    in_ch, n_cls, feat = (int(v) for v in z["meta"][:3])
    model = UNet(in_ch, n_cls, init_features=feat, compute_dtype=dtype)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w0/")}
    model.load_state_dict(sd, strict=True)
    model.cuda().train()
    return z, model


def grad_close(got, ref, key, rel):
    ref = np.asarray(ref, np.float64)
    got = np.asarray(got, np.float64)
    tol = rel * max(float(np.abs(ref).max()), 1e-4)
    err = float(np.abs(got - ref).max())
    assert err <= tol, f"{key}: max err {err:.3e} > {tol:.3e}"


@pytest.mark.parametrize("name", CASES)
def test_f32_forward_loss_grads_match_reference(golden_dir, name):
    z, model = load(golden_dir, name, "f32")
    w_ce, w_dice, lr, mom, eps = (float(v) for v in z["hyper"])
    x, tgt = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["target"]).cuda()
    loss, probs = model.forward_backward(x, tgt, w_ce, w_dice, eps, want_probs=True)
    torch.cuda.synchronize()
    p = probs.cpu().numpy()
    assert np.abs(p - z["probs"]).max() < 2e-5
    top2 = np.sort(z["probs"], axis=1)[:, -2:]
    safe = (top2[:, 1] - top2[:, 0]) > 1e-5
    assert safe.mean() > 0.999
    assert np.array_equal(p.argmax(1)[safe], z["argmax"][safe])
    np.testing.assert_allclose(loss.cpu().numpy(), z["loss"], rtol=2e-5, atol=1e-6)
    n = 0
    for k in z.files:
        if k.startswith("g0/"):
            grad_close(dict(model.named_parameters())[k[3:]].grad.cpu().numpy(), z[k], k, 2e-3)
            n += 1
    assert n > 5
    sd = model.state_dict()
    for k in z.files:
        if k.startswith("b1/"):
            np.testing.assert_allclose(sd[k[3:]].cpu().numpy(), z[k], rtol=1e-4, atol=1e-5, err_msg=k)
    # predict(): fused arg-max equals arg-max of the probabilities (BN buffers moved on: eval differs, so use train)
    am = model.predict(x).cpu().numpy()
    assert np.array_equal(am[safe], z["argmax"][safe])


@pytest.mark.parametrize("name", CASES[:2])
def test_f32_autograd_path_and_eval(golden_dir, name):
    """model(x) -> torch loss -> .backward(): the generic d(probs) path of the one-node autograd Function."""
    z, model = load(golden_dir, name, "f32")
    w_ce, w_dice, lr, mom, eps = (float(v) for v in z["hyper"])
    n_cls = int(z["meta"][1])
    x, tgt = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["target"]).cuda()
    probs = model(x)
    assert probs.requires_grad
    ce = F.nll_loss(torch.log(probs), tgt)
    onehot = F.one_hot(tgt, n_cls).permute(0, 3, 1, 2).float()
    inter, ps, ys = (probs * onehot).sum((0, 2, 3)), probs.sum((0, 2, 3)), onehot.sum((0, 2, 3))
    dice = 1.0 - ((2 * inter + eps) / (ps + ys + eps)).mean()
    loss = w_ce * ce + w_dice * dice
    loss.backward()
    np.testing.assert_allclose(loss.item(), z["loss"][0], rtol=2e-5)
    for k in z.files:
        if k.startswith("g0/"):
            grad_close(dict(model.named_parameters())[k[3:]].grad.cpu().numpy(), z[k], k, 2e-3)
    model.eval()
    with torch.no_grad():
        pe = model(x).cpu().numpy()
    assert np.abs(pe - z["probs_eval"]).max() < 2e-5
    top2 = np.sort(z["probs_eval"], axis=1)[:, -2:]
    safe = (top2[:, 1] - top2[:, 0]) > 1e-5
    assert np.array_equal(pe.argmax(1)[safe], z["argmax_eval"][safe])


@pytest.mark.parametrize("name", CASES[:2])
def test_f32_sgd_trajectory(golden_dir, name):
    from retinal_oct_image_segmentation_via_deep_learning_amd.optim import FusedSGD
    z, model = load(golden_dir, name, "f32")
    w_ce, w_dice, lr, mom, eps = (float(v) for v in z["hyper"])
    steps = int(z["meta"][6])
    x, tgt = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["target"]).cuda()
    opt = FusedSGD(model.parameters(), lr=lr, momentum=mom)
    losses = []
    for _ in range(steps):
        losses.append(model.forward_backward(x, tgt, w_ce, w_dice, eps)[0].item())
        opt.step()
    np.testing.assert_allclose(losses, z["traj_loss"], rtol=5e-5)
    sd = model.state_dict()
    for k in z.files:
        if k.startswith("wN/") and "num_batches" not in k:
            grad_close(sd[k[3:]].cpu().numpy(), z[k], k, 2e-3)
        elif k.startswith("wN/"):
            assert int(sd[k[3:]]) == int(z[k])


def rel_l2(got, ref):
    got, ref = np.asarray(got, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
    return float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30))


def bf16_against_rounding_aware_oracle(model, state, x, t, w_ce, w_dice, eps, tag, grad_tol=(0.4, 0.25)):
    """bf16 HIP step vs oracle/ref_cpu.OracleUNet(storage="bf16", dtype=float32): the same network with a bf16
    rounding at every point where the HIP path stores bf16, accumulating in fp32 like the MFMA does.  Forward
    quantities (probabilities, arg-max, loss) agree tightly.  Gradients of a FREE-RUNNING comparison cannot: a
    1-ulp difference (summation order) in an early layer shifts the small-batch BatchNorm statistics of the deep
    levels and re-rounds most of the network, so the per-tensor bound here is loose and the tight per-layer bound
    lives in test_headline_width_bf16_layers_teacher_forced_against_oracle."""
    from oracle import ref_cpu
    net = ref_cpu.OracleUNet(state, dtype=np.float32, storage="bf16")
    rp, (rl, rce, rdice), rg = net.loss_and_grads(x.numpy(), t.numpy(), w_ce, w_dice, eps)
    loss, probs = model.forward_backward(x.cuda(), t.cuda(), w_ce, w_dice, eps, want_probs=True)
    torch.cuda.synchronize()
    p = probs.cpu().numpy()
    assert np.abs(p - rp).max() < 2e-2, (tag, np.abs(p - rp).max())
    top2 = np.sort(rp, axis=1)[:, -2:]
    safe = (top2[:, 1] - top2[:, 0]) > 4e-2
    assert safe.mean() > 0.3 and np.array_equal(p.argmax(1)[safe], rp.argmax(1)[safe]), tag
    np.testing.assert_allclose(loss.cpu().numpy()[0], rl, rtol=3e-3, err_msg=tag)
    worst = []
    for k, prm in model.named_parameters():
        g = prm.grad.detach().cpu().numpy()
        if np.abs(rg[k]).max() < 1e-12:
            continue
        worst.append((rel_l2(g, rg[k]), k))
    worst.sort(reverse=True)
    assert worst[0][0] < grad_tol[0] and np.mean([w for w, _ in worst]) < grad_tol[1], (tag, worst[:5])
    return worst


@pytest.mark.parametrize("name", CASES)
def test_bf16_mode_matches_rounding_aware_oracle(golden_dir, name):
    """the reference fixtures' networks (narrow: generic bf16 kernels)"""
    z, model = load(golden_dir, name, "bf16")
    w_ce, w_dice, lr, mom, eps = (float(v) for v in z["hyper"])
    state = {k[3:]: z[k] for k in z.files if k.startswith("w0/")}
    bf16_against_rounding_aware_oracle(model, state, torch.from_numpy(z["x"]), torch.from_numpy(z["target"]),
                                       w_ce, w_dice, eps, name)
    # and the unrounded reference stays within bf16's own noise of it
    loss, probs = model.forward_backward(torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["target"]).cuda(),
                                         w_ce, w_dice, eps, want_probs=True)
    assert np.abs(probs.cpu().numpy() - z["probs"]).max() < 6e-2
    np.testing.assert_allclose(loss.cpu().numpy()[0], z["loss"][0], rtol=2e-2)


@pytest.mark.parametrize("name", CASES[:2])
def test_bf16_sgd_trajectory_descends_like_reference(golden_dir, name):
    from retinal_oct_image_segmentation_via_deep_learning_amd.optim import FusedSGD
    z, model = load(golden_dir, name, "bf16")
    w_ce, w_dice, lr, mom, eps = (float(v) for v in z["hyper"])
    x, tgt = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["target"]).cuda()
    opt = FusedSGD(model.parameters(), lr=lr, momentum=mom)
    losses = []
    for _ in range(int(z["meta"][6])):
        losses.append(model.forward_backward(x, tgt, w_ce, w_dice, eps)[0].item())
        opt.step()
    np.testing.assert_allclose(losses, z["traj_loss"], rtol=1.5e-2)
    assert losses[-1] < losses[0]


def _wide_fixture(golden_dir):
    import importlib.util
    spec = importlib.util.spec_from_file_location("test_oracle_helpers", os.path.join(os.path.dirname(__file__), "test_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.wide_case(golden_dir), mod.wide_grad_errors


def test_headline_width_f32_matches_reference_fixture(golden_dir):
    """UNet(1, 8, init_features=32) -- the benchmarked network -- in fp32 parity mode against the fixture
    tools/gen_golden_wide.py recorded from the reference module (run in float64)."""
    (z, state, x, t, model), grad_errors = _wide_fixture(golden_dir)
    model = model.cuda().train().set_compute_dtype("f32")
    loss, probs = model.forward_backward(x.cuda(), t.cuda(), want_probs=True)
    torch.cuda.synchronize()
    p = probs.cpu().numpy()
    assert np.abs(p - z["probs"]).max() < 2e-5
    top2 = np.sort(z["probs"], axis=1)[:, -2:]
    safe = (top2[:, 1] - top2[:, 0]) > 1e-5
    assert safe.mean() > 0.999 and np.array_equal(p.argmax(1)[safe], z["argmax"][safe])
    np.testing.assert_allclose(loss[0].item(), float(z["loss"]), rtol=2e-5)
    grads = {k: prm.grad.detach().cpu().numpy() for k, prm in model.named_parameters()}
    # a handful of the 3.5 M ReLU inputs sit within fp32 rounding of zero (fixture: min |z| = 1.4e-6): the deep,
    # small gradients may move by a flipped mask or two -- 2e-2 of the tensor's max still convicts a wrong tap
    assert not grad_errors(z, grads, 2e-2)


def test_headline_width_bf16_production_kernels_match_rounding_aware_oracle(golden_dir):
    """The kernels bench.py times (igemm2 / wgrad2 / first / head_mfma) end to end, per tensor, against the
    rounding-aware oracle, on the reference-pinned wide fixture's weights (64 x 128) and on 128 x 256."""
    (z, state, x, t, model), _ = _wide_fixture(golden_dir)
    model = model.cuda().train().set_compute_dtype("bf16")
    bf16_against_rounding_aware_oracle(model, state, x, t, 1.0, 0.0, 1e-7, "wide fixture 2x64x128, CE (deferred head)")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model.cuda()
    bf16_against_rounding_aware_oracle(model, state, x, t, 1.0, 0.25, 1e-7, "wide fixture 2x64x128, CE + Dice")
    g = torch.Generator().manual_seed(12)
    x2 = torch.randn(2, 1, 128, 256, generator=g)
    t2 = torch.randint(0, 8, (2, 128, 256), generator=g)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model.cuda()
    bf16_against_rounding_aware_oracle(model, state, x2, t2, 1.0, 0.0, 1e-7, "2x128x256")


def test_default_width_network_matches_oracle_in_f32():
    """init_features = 32 takes the fused head backward (feat == 32); fp32 mode against the fp64 oracle."""
    from oracle import ref_cpu
    from retinal_oct_image_segmentation_via_deep_learning_amd import UNet
    torch.manual_seed(11)
    model = UNet(1, 8, init_features=32, compute_dtype="f32").cuda().train()
    state = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 1, 32, 64, generator=g)
    t = torch.randint(0, 8, (2, 32, 64), generator=g)
    loss, probs = model.forward_backward(x.cuda(), t.cuda(), 1.0, 0.25, want_probs=True)
    net = ref_cpu.OracleUNet(state)
    rp, (rl, _, _), rg = net.loss_and_grads(x.numpy(), t.numpy(), 1.0, 0.25)
    assert np.abs(probs.cpu().numpy() - rp).max() < 2e-5
    np.testing.assert_allclose(loss[0].item(), rl, rtol=2e-5)
    for k, p in model.named_parameters():
        grad_close(p.grad.detach().cpu().numpy(), rg[k], k, 3e-3)


def test_api_errors_like_reference(golden_dir):
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.YNet_2022 import UNet, get_model
    z = np.load(os.path.join(golden_dir, "api.npz"))
    m = UNet(1, 2, init_features=4).cuda()
    with pytest.raises(RuntimeError, match="Sizes of tensors must match"):
        m(torch.zeros(1, 1, 62, 96, device="cuda"))
    with pytest.raises(AssertionError):
        get_model("nope")
    g = get_model("unet", in_channels=1, num_classes=9)
    assert sum(p.numel() for p in g.parameters()) == int(z["n_params"])


def test_trainer_hip_graph_replay_matches_eager_steps():
    """DataParallelTrainer(use_graph=True) records forward+loss+backward once and replays it: the
    parameters after 6 steps on fresh batches must equal the eager run's (up to the fp32 atomics'
    summation order in the weight-gradient kernels)."""
    from retinal_oct_image_segmentation_via_deep_learning_amd import UNet, ddp
    res = []
    for use_graph in (False, True):
        torch.manual_seed(0)
        model = UNet(1, 8, init_features=32).cuda().train()
        tr = ddp.DataParallelTrainer(model, lr=0.05, momentum=0.9, use_graph=use_graph, graph_warmup=2)
        g = torch.Generator().manual_seed(1)
        losses = []
        for _ in range(6):
            x = torch.randn(2, 1, 128, 256, generator=g).cuda()
            t = torch.randint(0, 8, (2, 128, 256), generator=g).cuda()
            losses.append(float(tr.step(x, t)[0]))
        torch.cuda.synchronize()
        assert (tr.graph is not None) == use_graph and tr.graph_error is None
        res.append((losses, tr.opt.flat_p.clone()))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=2e-4)
    rel = float((res[0][1] - res[1][1]).abs().max() / res[0][1].abs().max())
    assert rel < 5e-3, rel


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_deferred_cross_entropy_in_fused_head_backward(dtype):
    """forward_backward without a Dice term skips the forward head pass (init_features = 32, <= 8 classes):
    the loss then comes out of the fused head backward and must equal both the oracle's and the value of
    the ordinary forward head (`model.loss`)."""
    from oracle import ref_cpu
    from retinal_oct_image_segmentation_via_deep_learning_amd import UNet
    torch.manual_seed(21)
    model = UNet(1, 8, init_features=32, compute_dtype=dtype).cuda().train()
    state = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(22)
    x = torch.randn(2, 1, 32, 64, generator=g)
    t = torch.randint(0, 8, (2, 32, 64), generator=g)
    loss = model.forward_backward(x.cuda(), t.cuda(), 0.7, 0.0)
    grads = {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})     # rewind the BN buffers
    model.cuda()
    ref_fwd = model.loss(x.cuda(), t.cuda(), 0.7, 0.0)
    np.testing.assert_allclose(loss[:2].cpu().numpy(), ref_fwd[:2].cpu().numpy(), rtol=1e-5 if dtype == "f32" else 1e-4)
    assert float(loss[2]) == 0.0 and float(ref_fwd[2]) > 0.0   # Dice sums are not accumulated when its weight is 0
    if dtype == "f32":
        net = ref_cpu.OracleUNet(state)
        _, (rl, rce, _), rg = net.loss_and_grads(x.numpy(), t.numpy(), 0.7, 0.0)
        np.testing.assert_allclose(loss[0].item(), rl, rtol=2e-5)
        np.testing.assert_allclose(loss[1].item(), rce, rtol=2e-5)
        for k, v in grads.items():
            grad_close(v, rg[k], k, 3e-3)


def test_mfma_head_backward_matches_vector_formulation():
    """head_mfma.hip against head.hip's vector kernel (child process with OCT_HEAD_MFMA=0) on the same
    bf16 step: loss, every gradient and the BN buffers.  Differences come from the bf16 activation in the
    logits GEMM (the vector kernel keeps it in fp32) and the hi/lo split of W: ~1e-3 relative."""
    import subprocess
    import sys
    import tempfile
    code = r"""
import sys, torch
sys.path.insert(0, %r)
from retinal_oct_image_segmentation_via_deep_learning_amd import UNet
torch.manual_seed(31)
m = UNet(1, 8, init_features=32).cuda().train()
g = torch.Generator().manual_seed(32)
x = torch.randn(2, 1, 64, 96, generator=g).cuda(); t = torch.randint(0, 8, (2, 64, 96), generator=g).cuda()
out = {}
for wd in (0.0, 0.4):
    loss = m.forward_backward(x, t, 1.0, wd)
    out[f"loss{wd}"] = loss.cpu()
    for k, p in m.named_parameters():
        out[f"g{wd}/" + k] = p.grad.detach().cpu().clone()
torch.save(out, sys.argv[1])
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for flag in ("1", "0"):
            f = os.path.join(td, f"o{flag}.pt")
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, OCT_HEAD_MFMA=flag),
                               capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-2000:]
            res[flag] = torch.load(f, weights_only=True)
    a, b = res["1"], res["0"]
    for k in a:
        if k.startswith("loss"):
            np.testing.assert_allclose(a[k][:2].numpy(), b[k][:2].numpy(), rtol=3e-3)
        else:
            x, y = a[k].double().flatten(), b[k].double().flatten()
            cos = float(x @ y / (x.norm() * y.norm() + 1e-30))
            rel = float((x - y).norm() / (y.norm() + 1e-30))
            # the two runs also differ through fp32 atomics and ReLU-mask flips further down the network
            assert cos > 0.98 and (rel < 0.15 or "conv.weight" not in k), (k, cos, rel)
    for k in ("g0.0/conv.weight", "g0.0/conv.bias", "g0.4/conv.weight", "g0.4/conv.bias"):
        x, y = a[k].double().flatten(), b[k].double().flatten()
        assert float((x - y).norm() / y.norm()) < 2e-2, k


# ---- API edges where the nn.Module contract differs: loud, not silent ----------------------
def _tiny(dtype="f32", f=8, c=3):
    from retinal_oct_image_segmentation_via_deep_learning_amd import UNet
    torch.manual_seed(3)
    return UNet(1, c, init_features=f, compute_dtype=dtype).cuda()


def _randomise_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(1.0 + 0.3 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.3 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))


@pytest.mark.parametrize("net", ["ynet", "bionet"])
def test_frozen_batchnorm_fine_tune_backward_matches_torch_oracle(net):
    """model.eval() + .backward(): BatchNorm on its running statistics is a per-channel affine -- every parameter gradient
    (conv biases in front of a frozen BatchNorm included: no longer zero) against stock torch.nn in float64, the arithmetic
    the reference nn.Module (YNet_2022.py:548-569, BioNet_2020.py:24-75) runs on; buffers stay untouched."""
    from oracle import torch_unet
    torch.manual_seed(21)
    if net == "ynet":
        from retinal_oct_image_segmentation_via_deep_learning_amd import UNet
        model = UNet(1, 3, init_features=8, compute_dtype="f32")
        ref = torch_unet.TorchUNet(1, 3, 8)
        x = torch.randn(2, 1, 32, 48)
    else:
        from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.BioNet_2020 import UNet as BioUNet
        model = BioUNet(1, 2, compute_dtype="f32")
        ref = torch_unet.TorchBioUNet(1, 2)
        x = torch.randn(1, 1, 16, 24)
    _randomise_bn(model, 5)
    ref.load_state_dict(model.state_dict())
    ref = ref.double().eval()
    model = model.cuda().eval()
    buffers = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
    out = model(x.cuda())
    assert out.grad_fn is not None
    r = torch.randn(out.shape, generator=torch.Generator().manual_seed(9))
    (out * r.cuda()).sum().backward()
    rout = ref(x.double())
    (rout * r.double()).sum().backward()
    assert float((out.detach().cpu().double() - rout.detach()).abs().max()) < 2e-5 * max(1.0, float(rout.abs().max()))
    rg = dict(ref.named_parameters())
    for k, p in model.named_parameters():
        e = float((p.grad.cpu().double() - rg[k].grad).abs().max())
        assert e <= 2e-3 * max(float(rg[k].grad.abs().max()), 1e-4), (k, e)
    for k, v in model.state_dict().items():
        if k in buffers:
            assert torch.equal(v, buffers[k]), k        # eval mode: no statistics update, no step count
    with torch.no_grad():
        assert model(x.cuda()).grad_fn is None


def test_input_gradients_raise_clearly_and_second_backward_too():
    model = _tiny().train()
    x = torch.randn(1, 1, 32, 32, device="cuda")
    with pytest.raises(NotImplementedError, match="INPUT"):
        model(x.clone().requires_grad_(True))
    out = model(x)
    out.sum().backward()
    with pytest.raises(RuntimeError, match="second time"):
        out.sum().backward()


def test_out_of_range_target_gives_nan_loss():
    """torch's nll_loss raises on a label outside [0, C); the fused head cannot raise without a sync, so the
    loss is NaN -- on the forward head, the vector backward head and the MFMA backward head alike."""
    x = torch.randn(1, 1, 32, 64, device="cuda")
    for dtype, f, wd in (("f32", 8, 0.5), ("f32", 32, 0.0), ("bf16", 32, 0.0)):
        model = _tiny(dtype, f, 3).train()
        t = torch.randint(0, 3, (1, 32, 64), device="cuda")
        assert torch.isfinite(model.forward_backward(x, t, 1.0, wd)[0])
        t[0, 5, 7] = 3
        assert torch.isnan(model.forward_backward(x, t, 1.0, wd)[0])
        t[0, 5, 7] = -1
        assert torch.isnan(model.loss(x, t)[0])


def test_disable_v2_switch_runs_the_headline_width_in_bf16():
    """OCT_DISABLE_V2=1 (generic kernels only) must still train UNet(1,C,32) in bf16: the engine
    asks the library whether the fused first-layer weight gradient applies instead of assuming it."""
    import subprocess
    import sys
    code = ("import torch; from retinal_oct_image_segmentation_via_deep_learning_amd import UNet;"
            "torch.manual_seed(0); m = UNet(1, 4, init_features=32, compute_dtype='bf16').cuda().train();"
            "x = torch.randn(1, 1, 32, 64, device='cuda'); t = torch.randint(0, 4, (1, 32, 64), device='cuda');"
            "l = m.forward_backward(x, t); torch.cuda.synchronize(); assert torch.isfinite(l[0]);"
            "assert all(torch.isfinite(p.grad).all() for p in m.parameters()); print('ok', float(l[0]))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for flag in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OCT_DISABLE_V2=flag, PYTHONPATH=root),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
        outs[flag] = float(r.stdout.split()[-1])
    assert abs(outs["1"] - outs["0"]) < 0.05 * abs(outs["0"])


def test_headline_width_bf16_layers_teacher_forced_against_oracle(golden_dir):
    """Every 3x3 layer and every transposed convolution of the benchmarked network, IN SITU: the bf16 step runs once on the reference-pinned wide
    fixture, and each layer is then redone by the oracle from the tensors the kernels actually read (its own bf16
    inputs, BatchNorm coefficients and incoming dY).  Errors cannot compound this way, so the bound is the one a single
    bf16 store allows -- a free-running comparison cannot have it: with bf16 storage a 1-ulp difference in an early
    layer moves the small-batch BatchNorm statistics of the deep levels and re-rounds most of the network (measured:
    78 % of the bottleneck's elements differ by a few ulps between fp32- and fp64-accumulating evaluations)."""
    from oracle import ref_cpu as R
    (z, state, x, t, model), _ = _wide_fixture(golden_dir)
    model = model.cuda().train().set_compute_dtype("bf16")
    model._engine.debug = {}
    model.forward_backward(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    dbg, model._engine.debug = model._engine.debug, None
    grads = {k: p.grad.detach().cpu().numpy() for k, p in model.named_parameters()}
    layers = [k for k in dbg if k.startswith("layer:")]
    assert len(layers) == 17                       # 18 convs minus the first layer (its backward is one fused kernel)

    def nchw(tn):
        return tn.cpu().numpy().transpose(0, 3, 1, 2).astype(np.float64)

    def act(xk, bn):
        a = nchw(xk)
        if bn is not None:
            sc, sh = (v.cpu().numpy().astype(np.float64) for v in bn)
            a = R.round_bf16(np.maximum(a * sc[None, :, None, None] + sh[None, :, None, None], 0))
        return a

    def ulps(got, ref):
        d = np.abs(got - ref)
        return d / np.maximum(2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 1e-30))) - 7), 1e-30)   # bf16 spacing at |ref|

    for key in layers:
        rec, wk = dbg[key], key[6:]
        a = act(rec["x0"], rec["bn0"])
        if rec["x1"] is not None:
            a = np.concatenate([a, act(rec["x1"], rec["bn1"])], axis=1)
        wq = R.round_bf16(state[wk])
        # forward: y = bf16(conv(a, w)): all but a handful of elements identical, none further than one rounding away
        y_ref = R.conv3x3_fwd(a, wq)
        u = ulps(nchw(rec["y"]), R.round_bf16(y_ref))
        assert (u > 0).mean() < 1e-2 and (u > 1.01).mean() < 1e-4 and u.max() < 64, (wk, "fprop", (u > 0).mean(), u.max())
        # backward from the dY this layer received
        dy = nchw(rec["dy"])
        dx_ref, dw_ref = R.conv3x3_bwd(a, wq, dy)
        assert rel_l2(grads[wk], dw_ref) < 2e-3, (wk, "wgrad", rel_l2(grads[wk], dw_ref))
        dx = nchw(rec["d0"]) if rec["d1"] is None else np.concatenate([nchw(rec["d0"]), nchw(rec["d1"])], axis=1)
        u = ulps(dx, R.round_bf16(dx_ref))
        # (a heavily cancelling sum can sit several bf16 ulps of its own small value away from the fp32 accumulation)
        assert (u > 0).mean() < 1e-2 and (u > 1.01).mean() < 1e-4 and u.max() < 64, (wk, "dgrad", (u > 0).mean(), u.max())
    # the four transposed convolutions, the same way: forward with BN + ReLU on load and the depth-to-space
    # store (+ bias), data gradient through the space-to-depth gather, weight and bias gradient from the dU they received
    ups = [k for k in dbg if k.startswith("up:")]
    assert len(ups) == 4
    for key in ups:
        rec, wk = dbg[key], key[3:]
        a = act(rec["x"], rec["bn"])
        wq = R.round_bf16(state[wk])
        u_ref = R.deconv2x2_fwd(a, wq, state[rec["bkey"]].astype(np.float64))
        uu = ulps(nchw(rec["u"]), R.round_bf16(u_ref))
        assert (uu > 0).mean() < 1e-2 and (uu > 1.01).mean() < 1e-4 and uu.max() < 64, (wk, "deconv fprop", (uu > 0).mean(), uu.max())
        du = nchw(rec["du"])
        da_ref, dw_ref, db_ref = R.deconv2x2_bwd(a, wq, du)
        assert rel_l2(grads[wk], dw_ref) < 2e-3, (wk, "deconv wgrad", rel_l2(grads[wk], dw_ref))
        assert rel_l2(grads[rec["bkey"]], db_ref) < 2e-3, (wk, "deconv bias gradient", rel_l2(grads[rec["bkey"]], db_ref))
        uu = ulps(nchw(rec["da"]), R.round_bf16(da_ref))
        assert (uu > 0).mean() < 1e-2 and (uu > 1.01).mean() < 1e-4 and uu.max() < 64, (wk, "deconv dgrad", (uu > 0).mean(), uu.max())


@pytest.mark.parametrize("dtype,shape", [("bf16", (2, 128, 256)), ("f32", (2, 32, 64))])
def test_deterministic_weight_gradients_are_bit_reproducible(dtype, shape):
    """engine.deterministic (OctWgradDesc.partials): per-workgroup partial slabs summed in index order instead of fp32
    atomics.  Two runs give bit-identical convolution / transposed-convolution weight and bias gradients; against the
    atomics mode they differ by fp32 round-off only.  (The 1x1 head's 8 x 32 weights and its bias still meet through
    atomics inside the fused head kernel and are not part of the bit-equality claim.)"""
    from retinal_oct_image_segmentation_via_deep_learning_amd import UNet
    torch.manual_seed(5)
    model = UNet(1, 8, init_features=32, compute_dtype=dtype).cuda().train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(6)
    b, h, w = shape
    x = torch.randn(b, 1, h, w, generator=g).cuda()
    t = torch.randint(0, 8, (b, h, w), generator=g).cuda()

    def grads(det):
        model.load_state_dict(state)
        model._engine.deterministic = det
        model.forward_backward(x, t)
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    a, b2, c = grads(True), grads(True), grads(False)
    model._engine.deterministic = False
    conv_keys = [k for k in a if ("conv" in k and k.endswith("weight") and not k.startswith("conv.")) or k.startswith("upconv")]
    assert len(conv_keys) == 18 + 8
    for k in conv_keys:
        assert torch.equal(a[k], b2[k]), f"{k}: deterministic mode is not reproducible"
        ref = c[k].double()
        assert float((a[k].double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-12, k
    for k in a:     # everything, loosely: the two modes compute the same gradients
        ref = c[k].double()
        assert float((a[k].double() - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + 1e-9, k
