"""GPU parity of the ReLayNet drop-ins against the fixtures produced by the reference's own
classes (tools/gen_golden_relaynet.py): 7x3 implicit-GEMM convolution (fprop, dgrad, wgrad in three row groups),
BatchNorm + PReLU (with the slope's gradient), MaxPool2d with indices, MaxUnpool2d, cat((skip, unpooled)).
fp32 parity mode: outputs 2e-5 of their scale, gradients 2e-3 of the tensor's max, pooling indices identical;
bf16 production mode: against the same fixtures with bf16-sized bounds."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_oracle_relaynet import NETS, block_io

pytestmark = pytest.mark.gpu

P = {"num_channels": 3, "num_filters": 8, "kernel_h": 7, "kernel_w": 3, "stride_conv": 1, "pool": 2, "stride_pool": 2,
     "kernel_c": 1}


def dropin(name, dtype):
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Lesions_Segment import ReLayNet_2017 as R
    return {"relay_basic": lambda: R.BasicBlock(dict(P), dtype), "relay_encoder": lambda: R.EncoderBlock(dict(P), dtype),
            "relay_decoder": lambda: R.DecoderBlock(dict(P, num_channels=16), dtype),
            "relay_classifier": lambda: R.ClassifierBlock(dict(P, num_channels=8, num_class=5), dtype)}[name]()


def close(got, ref, key, rel, floor=1e-4):
    ref = np.asarray(ref, np.float64)
    tol = rel * max(float(np.abs(ref).max()), floor)
    err = float(np.abs(np.asarray(got, np.float64) - ref).max())
    assert err <= tol, f"{key}: max err {err:.3e} > {tol:.3e}"


@pytest.mark.parametrize("name", ["relay_basic", "relay_encoder", "relay_decoder", "relay_classifier"])
def test_f32_block_matches_reference_fixture(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    m = dropin(name, "f32")
    m.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w0/")}, strict=True)
    m.cuda().train()
    xs, extra = block_io(z)
    xd = [x.cuda().requires_grad_(True) for x in xs]
    out = m(*xd, *[e.cuda() for e in extra])
    outs = out if isinstance(out, tuple) else (out,)
    assert len(outs) == int(z["n_out"])
    loss = 0
    for i, o in enumerate(outs):
        if o.dtype.is_floating_point:
            close(o.detach().cpu().numpy(), z[f"out{i}"], f"out{i}", 2e-5, 1.0)
            loss = loss + (o * torch.from_numpy(z[f"r{i}"]).cuda()).sum()
        else:
            assert o.dtype == torch.int64 and np.array_equal(o.cpu().numpy(), z[f"out{i}"]), "pooling indices"
    loss.backward()
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
        oe = m(*[x.cuda() for x in xs], *[e.cuda() for e in extra])
        close((oe[0] if isinstance(oe, tuple) else oe).cpu().numpy(), z["out_eval"], "out_eval", 2e-5, 1.0)


def load_net(golden_dir, name, dtype):
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Lesions_Segment.ReLayNet_2017 import ReLayNet
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    seed, n, cin, ncls, nf, h, w = (int(v) for v in z["meta"])
    m = ReLayNet(in_channels=cin, num_classes=ncls, num_filters=nf, compute_dtype=dtype)
    m.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w0/")}, strict=True)
    return z, m.cuda().train()


@pytest.mark.parametrize("name", NETS)
def test_f32_network_matches_reference_fixture(golden_dir, name):
    z, m = load_net(golden_dir, name, "f32")
    x, t = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["target"]).cuda()
    out = m(x)
    lg = out.detach().cpu().numpy()
    close(lg, z["logits"], "logits", 2e-5, 1.0)
    assert np.array_equal(lg.argmax(1), z["argmax"])
    loss = F.cross_entropy(out, t)
    np.testing.assert_allclose(float(loss.detach()), float(z["loss"][0]), rtol=2e-5)
    loss.backward()
    for k, p in m.named_parameters():
        close(p.grad.cpu().numpy(), z["g/" + k], k, 2e-3)
    sd = m.state_dict()
    for k in z.files:
        if k.startswith("b1/") and "running" in k:
            close(sd[k[3:]].cpu().numpy(), z[k], k, 1e-4)
    m.eval()
    with torch.no_grad():
        close(m(x).cpu().numpy(), z["logits_eval"], "logits_eval", 2e-5, 1.0)


@pytest.mark.parametrize("name", NETS)
def test_bf16_network_is_close_and_trains(golden_dir, name):
    z, m = load_net(golden_dir, name, "bf16")
    x, t = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["target"]).cuda()
    out = m(x)
    ref = z["logits"]
    err = np.abs(out.detach().cpu().numpy() - ref)
    assert err.max() < 0.08 * max(1.0, np.abs(ref).max()) and err.mean() < 0.012 * max(1.0, np.abs(ref).mean())
    loss = F.cross_entropy(out, t)
    np.testing.assert_allclose(float(loss.detach()), float(z["loss"][0]), rtol=2e-2)
    loss.backward()
    # per tensor: relative L2 distance to the reference's fp32 gradient (bf16 storage of every activation and of dY: ~2^-8
    # per element, accumulating over the 3-7 layers between the loss and the tensor; round 2 only asked for a MEAN cosine > 0.9)
    worst = {}
    for k, p in m.named_parameters():
        a, b = p.grad.flatten().double().cpu(), torch.from_numpy(z["g/" + k]).flatten().double()
        if float(b.norm()) > 1e-6 and b.numel() >= 8:
            worst[k] = float((a - b).norm() / b.norm())
    # (measured: 0.003-0.22 for the convolutions of the full-resolution levels, 0.32-0.42 for the bottleneck of the tiny fixture,
    #  whose BatchNorm sees 48 samples per channel: tests/probes/bf16_probe.py -- two valid bf16 evaluations differ as much there)
    assert max(worst.values()) < 0.5 and np.median(list(worst.values())) < 0.3, (name, worst)
    opt = torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9)
    losses = []
    for _ in range(5):
        opt.zero_grad()
        l = F.cross_entropy(m(x), t)
        l.backward()
        opt.step()
        losses.append(float(l))
    assert losses[-1] < losses[0], losses


def test_api_edges_like_reference(golden_dir):
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Lesions_Segment.ReLayNet_2017 import ReLayNet
    api = np.load(os.path.join(golden_dir, "relaynet_api.npz"))
    assert "Sizes of tensors must match" in str(api["negative_msg"])
    m = ReLayNet(1, 4, num_filters=4, compute_dtype="f32").cuda()
    with pytest.raises(RuntimeError, match="Sizes of tensors must match"):
        m(torch.zeros(1, 1, 20, 24, device="cuda"))         # 20 / 8 is not whole: the reference fails at torch.cat too
    out = m(torch.randn(1, 1, 24, 16, device="cuda"))
    assert out.shape == (1, 4, 24, 16)                       # logits (the reference never applies its Softmax2d)
    assert float((out.softmax(1).sum(1) - 1).abs().max()) < 1e-5


def test_headline_width_7x3_convolution_matches_torch():
    """num_filters = 64 (the reference default) at 2 x 64 x 96: the 7x3 kernels on K = 21*64 and 21*128 (concat),
    fp32 mode against stock torch's convolution of the same weights on the device."""
    from oracle.torch_relaynet import TorchReLayNet
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Lesions_Segment.ReLayNet_2017 import ReLayNet
    torch.manual_seed(3)
    m = ReLayNet(1, 10, compute_dtype="f32").cuda().train()
    ref = TorchReLayNet(1, 10, 64)
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    ref = ref.double().train()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 1, 64, 96, generator=g)
    t = torch.randint(0, 10, (2, 64, 96), generator=g)
    out = m(x.cuda())
    ro = ref(x.double())
    close(out.detach().cpu().numpy(), ro.detach().numpy(), "logits", 5e-5, 1.0)
    F.cross_entropy(out, t.cuda()).backward()
    F.cross_entropy(ro, t).backward()
    rg = dict(ref.named_parameters())
    for k, p in m.named_parameters():
        close(p.grad.cpu().numpy(), rg[k].grad.numpy(), k, 5e-3)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_deterministic_mode_matches_atomics_on_7x3_weight_gradients(golden_dir, dtype):
    """engine.deterministic must not change a 7x3 weight gradient beyond summation order: its unpack entry point
    (oct_unpack_wgrad_kk) reads ONE slab, so these launches stay on atomics (UNetEngine._wgrad, partials_ok)."""
    from retinal_oct_image_segmentation_via_deep_learning_amd import ops
    z = np.load(os.path.join(golden_dir, "relay_encoder.npz"))
    m = dropin("relay_encoder", dtype)
    m.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w0/")}, strict=True)
    m.cuda().train()
    xs, extra = block_io(z)
    e = ops.kernels(dtype)
    keep = e.deterministic
    grads = {}
    try:
        for det in (False, True):
            e.deterministic = det
            m.zero_grad(set_to_none=True)
            out = m(*[x.cuda() for x in xs], *[t.cuda() for t in extra])
            outs = out if isinstance(out, tuple) else (out,)
            sum((o * torch.from_numpy(z[f"r{i}"]).cuda()).sum() for i, o in enumerate(outs) if o.dtype.is_floating_point).backward()
            grads[det] = {k: p.grad.clone() for k, p in m.named_parameters()}
    finally:
        e.deterministic = keep
    for k in grads[False]:
        a, b = grads[False][k].double(), grads[True][k].double()
        assert float((a - b).abs().max()) <= 1e-5 * max(float(a.abs().max()), 1e-6), k


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("shape,k", [((2, 12, 16, 8), 2), ((1, 9, 6, 16), 3), ((3, 4, 4, 5), 2)])
def test_window_code_pool_unpool_equals_the_index_form(dtype, shape, k):
    """ops.MaxPoolCode / MaxUnpoolCode (one-byte window codes inside the network) against ops.MaxPoolIdx / MaxUnpool (torch's
    int64 plane indices, ReLayNet_2017.py:174-188) and against torch itself: same winners under ties (first maximum in row-major
    window order), same un-pooled tensor without a zero fill, same gradients -- bit for bit."""
    from retinal_oct_image_segmentation_via_deep_learning_amd import ops
    n, h, w, c = shape
    g = torch.Generator().manual_seed(h * 131 + w + k)
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    a = torch.randint(-3, 4, shape, generator=g).to(tdt).cuda()          # few distinct values: plenty of ties
    a1, a2 = a.clone().requires_grad_(True), a.clone().requires_grad_(True)
    p1, idx = ops.MaxPoolIdx.apply(dtype, k, a1)
    p2, code = ops.MaxPoolCode.apply(dtype, k, a2)
    assert torch.equal(p1, p2) and code.dtype == torch.uint8
    yo = torch.arange(h // k, device="cuda").view(1, -1, 1, 1)
    xo = torch.arange(w // k, device="cuda").view(1, 1, -1, 1)
    assert torch.equal((yo * k + code.long() // k) * w + xo * k + code.long() % k, idx), "codes name torch's winners"
    tp, tidx = F.max_pool2d(a.float().permute(0, 3, 1, 2), k, k, return_indices=True)
    assert torch.equal(tidx.permute(0, 2, 3, 1), idx) and torch.equal(tp.permute(0, 2, 3, 1).to(tdt), p1)
    v = torch.randint(-4, 5, p1.shape, generator=g).to(tdt).cuda()
    v1, v2 = v.clone().requires_grad_(True), v.clone().requires_grad_(True)
    u1 = ops.MaxUnpool.apply(dtype, k, v1, idx)
    u2 = ops.MaxUnpoolCode.apply(dtype, k, v2, code)
    assert torch.equal(u1, u2) and not torch.isnan(u2.float()).any()
    r = torch.randint(-2, 3, u1.shape, generator=g).to(tdt).cuda()
    (u1.float() * r.float()).sum().backward()
    (u2.float() * r.float()).sum().backward()
    assert torch.equal(v1.grad, v2.grad)
    rp = torch.randint(-2, 3, p1.shape, generator=g).to(tdt).cuda()
    (p1.float() * rp.float()).sum().backward()
    (p2.float() * rp.float()).sum().backward()
    assert torch.equal(a1.grad, a2.grad)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 12, 20, 64), (1, 7, 9, 8), (3, 16, 16, 128)])
def test_prelu_inside_the_batchnorm_backward_passes_equals_the_separate_pass(dtype, shape):
    """oct_dact_bn_reduce_prelu + oct_bn_bwd_apply_prelu_to (dz = dA * (z > 0 ? 1 : alpha) re-derived in both BatchNorm-backward
    passes, never written) against oct_affine_prelu_bwd followed by the plain passes: partial sums and dy BIT-identical (the value
    that enters them is rounded to the storage type where the separate pass stored it), d(alpha) equal up to the order of its
    fp32 atomics."""
    from retinal_oct_image_segmentation_via_deep_learning_amd import _lib as L
    lib = L.lib()
    n, h, w, c = shape
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + c)
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    dt = L.DT_F32 if dtype == "f32" else L.DT_BF16
    assert lib.oct_prelu_bn_fused_ok(dt, c) == 1 and lib.oct_prelu_bn_fused_ok(dt, 12) == 0
    y = torch.randn(shape, generator=g).to(tdt).cuda()
    da = torch.randn(shape, generator=g).to(tdt).cuda()
    sc = (torch.rand(c, generator=g) + 0.5).cuda(); sh = (torch.randn(c, generator=g) * 0.3).cuda()
    mean = (torch.randn(c, generator=g) * 0.1).cuda(); invstd = (torch.rand(c, generator=g) + 0.5).cuda()
    alpha = torch.tensor([0.25], device="cuda")
    coef = torch.randn(3, c, generator=g).cuda()
    st = torch.cuda.current_stream().cuda_stream
    npix = n * h * w
    nblk = lib.oct_dact_bn_reduce_blocks(n, h, w, c, 0)
    # separate pass, then the plain reduction (mask held open) and apply
    dz = torch.empty_like(da); dal1 = torch.zeros(1, device="cuda")
    L.check(lib.oct_affine_prelu_bwd(dt, da.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr(), alpha.data_ptr(), dz.data_ptr(),
                                     dal1.data_ptr(), npix, c, st))
    p1 = torch.full((nblk, 2, c), float("nan"), device="cuda")
    zero, one = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    L.check(lib.oct_dact_bn_reduce(dt, dz.data_ptr(), None, y.data_ptr(), zero.data_ptr(), one.data_ptr(), mean.data_ptr(),
                                   invstd.data_ptr(), None, p1.data_ptr(), n, h, w, c, st))
    dy1 = torch.empty_like(da)
    L.check(lib.oct_bn_bwd_apply_to(dt, dy1.data_ptr(), dz.data_ptr(), y.data_ptr(), coef.data_ptr(), None, None, npix, c, st))
    # fused
    p2 = torch.full((nblk, 2, c), float("nan"), device="cuda"); dal2 = torch.zeros(1, device="cuda")
    L.check(lib.oct_dact_bn_reduce_prelu(dt, da.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr(), alpha.data_ptr(),
                                         mean.data_ptr(), invstd.data_ptr(), p2.data_ptr(), dal2.data_ptr(), n, h, w, c, st))
    dy2 = torch.full_like(da, float("nan"))
    L.check(lib.oct_bn_bwd_apply_prelu_to(dt, dy2.data_ptr(), da.data_ptr(), y.data_ptr(), coef.data_ptr(), sc.data_ptr(),
                                          sh.data_ptr(), alpha.data_ptr(), npix, c, st))
    torch.cuda.synchronize()
    assert torch.equal(p1, p2), "BatchNorm-backward partial sums"
    assert torch.equal(dy1, dy2), "dy"
    torch.testing.assert_close(dal1, dal2, rtol=1e-4, atol=1e-4)
