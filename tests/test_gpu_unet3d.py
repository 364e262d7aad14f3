"""Volumetric U-Net (cfg5) on the GPU.  There is NO reference implementation of this network
(only the unused `ffc3d` flag, YNet_2022.py:161,194): the oracle is the same graph on stock torch.nn
(oracle/torch_unet3d.py) evaluated in float64 on the host -- PARITY UNPINNED BY THE REFERENCE.
fp32 parity mode: probabilities 2e-5, arg-max identical on safe-margin voxels, gradients 2e-3 of the tensor's max;
kernel level: the depth-tap implicit GEMM against torch's conv3d / conv_transpose3d, exact-arithmetic operands."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from retinal_oct_image_segmentation_via_deep_learning_amd import _lib as L
    from retinal_oct_image_segmentation_via_deep_learning_amd import engine as E
    L.lib()
    return L, E


def ndhwc(t5, dt):
    """(B,C,D,H,W) -> (B*D, H, W, C) device tensor of the compute dtype"""
    b, c, d, h, w = t5.shape
    return t5.permute(0, 2, 3, 4, 1).reshape(b * d, h, w, c).contiguous().to("cuda", torch.float32 if dt == "f32" else torch.bfloat16)


def from_ndhwc(t4, b):
    n, h, w, c = t4.shape
    return t4.float().cpu().reshape(b, n // b, h, w, c).permute(0, 4, 1, 2, 3).double()


def ints(g, shape, lo=-3, hi=3):
    return torch.randint(lo, hi + 1, shape, generator=g).double()


def pow2(g, shape, density=0.5):
    mag = torch.exp2(torch.randint(-3, 1, shape, generator=g).double())
    sign = torch.randint(0, 2, shape, generator=g).double() * 2 - 1
    keep = (torch.rand(shape, generator=g) < density).double()
    return mag * sign * keep


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 4, 8, 32, 16, 0, 32), (1, 6, 10, 20, 1, 0, 8), (1, 4, 8, 32, 8, 8, 16), (2, 2, 8, 32, 32, 32, 64),
                                   (2, 4, 16, 64, 1, 0, 32), (1, 3, 8, 24, 1, 0, 16),      # first layer: direct 27-tap kernels (bf16)
                                   (1, 4, 16, 32, 64, 0, 128), (2, 2, 16, 64, 32, 0, 32),   # pipelined kernels, whole tiles
                                   (1, 2, 16, 32, 32, 32, 32), (1, 2, 16, 32, 32, 0, 64)])  # 32x64 / 64x32 weight-gradient blocks
def test_conv3d_fprop_dgrad_wgrad_exact(env, dt, shape):
    """Conv3d 3x3x3 through the depth-tap GEMM (fprop + stats, dgrad with concat split, three-launch wgrad):
    integer activations, power-of-two weights -> every sum exact -> bit equality with torch's float64 conv3d."""
    L, E = env
    b, d, h, w, c0, c1, cout = shape
    cin = c0 + c1
    g = torch.Generator().manual_seed(abs(hash(shape)) % 2**31)
    eng = E.UNetEngine(1, 2, 4, dt)
    x = ints(g, (b, cin, d, h, w))
    wt = pow2(g, (cout, cin, 3, 3, 3))
    dy = ints(g, (b, cout, d, h, w), -2, 2)
    x0 = ndhwc(x[:, :c0], dt)
    x1 = ndhwc(x[:, c0:], dt) if c1 else None
    src = E.Src(x0, c0, None, x1, c1, None)
    wd = wt.float().cuda().contiguous()
    n = b * d
    y = torch.full((n, h, w, cout), float("nan"), dtype=x0.dtype, device="cuda")
    part = torch.full((eng._stat_blocks(cout, n, h, w, src, depth=d), 2, cout), float("nan"), dtype=torch.float32, device="cuda")
    eng._conv(src, eng._pack("w3", wd, L.PACK_CONV3D_FPROP, cout, cin), cout, 9, n, h, w, y, stats=part, depth=d)
    ref = F.conv3d(x, wt, padding=1)
    rnd = (lambda t: t.float().to(torch.bfloat16).double()) if dt == "bf16" else (lambda t: t)
    assert torch.equal(from_ndhwc(y, b), rnd(ref)), "Conv3d fprop"
    assert torch.equal(part.double().sum(0)[0].cpu(), ref.sum(dim=(0, 2, 3, 4))), "BatchNorm3d sum(y)"
    # dgrad (split into the two concat sources)
    dyd = ndhwc(dy, dt)
    d0 = torch.full((n, h, w, c0), float("nan"), dtype=x0.dtype, device="cuda")
    d1 = torch.full((n, h, w, c1), float("nan"), dtype=x0.dtype, device="cuda") if c1 else None
    eng._conv(E.Src(dyd, cout), eng._pack("w3", wd, L.PACK_CONV3D_DGRAD, cout, cin), cin, 9, n, h, w, d0, y1=d1,
              split=c0 if c1 else 0, depth=d)
    dx = torch.nn.grad.conv3d_input((b, cin, d, h, w), wt, dy, padding=1)
    assert torch.equal(from_ndhwc(d0, b), rnd(dx[:, :c0])), "Conv3d dgrad part 0"
    if c1:
        assert torch.equal(from_ndhwc(d1, b), rnd(dx[:, c0:])), "Conv3d dgrad part 1"
    # wgrad: one launch per depth tap into its slab
    slab = torch.zeros((3, 9, cout, cin), dtype=torch.float32, device="cuda")
    for kdi in range(3):
        eng._wgrad(src, dyd, cout, 9, n, h, w, depth=d, in_shift=kdi - 1, dwp=slab[kdi])
    grad = torch.full((cout, cin, 3, 3, 3), float("nan"), dtype=torch.float32, device="cuda")
    L.check(L.lib().oct_unpack_wgrad3d(L.PACK_CONV3D_FPROP, slab.data_ptr(), grad.data_ptr(), cout, cin, 0, 0,
                                       torch.cuda.current_stream().cuda_stream))
    dw = torch.nn.grad.conv3d_weight(x, (cout, cin, 3, 3, 3), dy, padding=1)
    assert torch.equal(grad.double().cpu(), dw), "Conv3d wgrad"
    # first layer on the matrix pipe: all 27 taps in ONE launch (OCT_IMG_SHIFT_ALL) must give the same slab
    import ctypes as C
    qd = L.WgradDesc(eng.dt, n, h, w, c0, c1, cout, 9, 0, 0, L.IN_PLAIN, 0, 0, d, L.IMG_SHIFT_ALL, 0, 0, 0)
    ok = L.lib().oct_conv_wgrad_all_depth_taps_ok(C.byref(qd))
    assert ok == (1 if (dt == "bf16" and cin == 1 and w % 32 == 0 and cout >= 32) else 0)
    if ok:
        slab1 = torch.zeros((3, 9, cout, cin), dtype=torch.float32, device="cuda")
        eng._wgrad(src, dyd, cout, 9, n, h, w, depth=d, in_shift=L.IMG_SHIFT_ALL, dwp=slab1)
        assert torch.equal(slab1, slab), "Conv3d(1 -> F) wgrad, all depth taps in one launch"


# depth-rolling kernel (roll3d.hip: Cin = 32 per depth tap, Cout = 32 / 64, whole 8 x 32 tiles): (volumes, depth, H, W, Cout, split,
# transform on load).  320 columns for 256 workgroups (several columns per workgroup, ring slots running on across columns),
# the shortest volume (depth 2: every item touches a padding slice), a 64-channel output split over two tensors (the data
# gradient of a concat), BN + ReLU and plain-affine transforms on load (padding applies to the ACTIVATED tensor)
ROLL = [(5, 3, 128, 128, 32, 0, "relu"), (2, 2, 8, 32, 32, 0, None), (1, 7, 24, 64, 64, 32, "affine"), (1, 5, 16, 32, 64, 0, "relu"),
        (3, 4, 8, 96, 32, 0, None)]


@pytest.mark.parametrize("shape", ROLL)
def test_conv3d_depth_rolling_kernel_bit_exact(env, shape):
    """3x3x3 convolution of 32 channels on the depth-rolling kernel against torch's float64 conv3d on exactly representable
    operands: output, BatchNorm sums, and the same launch with OCT_ROLL3D-ineligible routing (igemm2's depth-tap mode) as a
    second witness.  No reference counterpart (parity unpinned by the reference)."""
    L, E = env
    b, d, h, w, cout, split, xf = shape
    g = torch.Generator().manual_seed(abs(hash(shape)) % 2**31 + 5)
    eng = E.UNetEngine(1, 2, 4, "bf16")
    x = ints(g, (b, 32, d, h, w))
    bn = None
    eff = x
    if xf:
        sc = torch.tensor([-1.0, -0.5, 0.5, 1.0, 2.0])[torch.randint(0, 5, (32,), generator=g)]
        sh = torch.randint(-2, 3, (32,), generator=g) * 0.5
        eff = x * sc.double().view(1, 32, 1, 1, 1) + sh.double().view(1, 32, 1, 1, 1)
        if xf == "relu":
            eff = eff.clamp_min(0)
        bn = E.BNState(sc.float().cuda(), sh.float().cuda(), relu=(xf == "relu"))
    wt = pow2(g, (cout, 32, 3, 3, 3))
    src = E.Src(ndhwc(x, "bf16"), 32, bn)
    n = b * d
    from retinal_oct_image_segmentation_via_deep_learning_amd import _lib
    import ctypes as C
    ref = F.conv3d(eff, wt, padding=1)
    assert ref.abs().max() * 8 < 2 ** 22
    rnd = lambda t: t.float().to(torch.bfloat16).double()
    c_a = split if split else cout
    y0 = torch.full((n, h, w, c_a), float("nan"), dtype=torch.bfloat16, device="cuda")
    y1 = torch.full((n, h, w, cout - split), float("nan"), dtype=torch.bfloat16, device="cuda") if split else None
    rows = eng._stat_blocks(cout, n, h, w, src, depth=d)
    assert rows == min(256, (w // 32) * (h // 8) * b), "the rolling kernel writes one BatchNorm row per workgroup"
    # (a 32-channel output WITHOUT BatchNorm sums stays on igemm2's 16-row tiles, see roll3d.hip: every case here asks for
    #  the sums or has 64 output channels)
    part = torch.full((rows, 2, cout), float("nan"), dtype=torch.float32, device="cuda")
    wp = eng._pack("w3", wt.float().cuda().contiguous(), L.PACK_CONV3D_FPROP, cout, 32)
    eng._conv(src, wp, cout, 9, n, h, w, y0, y1=y1, split=split, stats=None if split else part, depth=d)
    got = from_ndhwc(y0, b) if not split else torch.cat([from_ndhwc(y0, b), from_ndhwc(y1, b)], dim=1)
    assert torch.equal(got, rnd(ref)), "depth-rolling Conv3d"
    if not split:
        assert torch.equal(part.double().sum(0)[0].cpu(), ref.sum(dim=(0, 2, 3, 4))), "BatchNorm3d sum(y)"
        # (the squares leave the exactly representable range of the fp32 partial sums: a tolerance, as in test_gpu_exact.py)
        torch.testing.assert_close(part.double().sum(0)[1].cpu(), (ref * ref).sum(dim=(0, 2, 3, 4)), rtol=1e-5, atol=1e-2)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 2, 4, 8, 16, 8), (1, 3, 8, 32, 64, 32)])
def test_deconv3d_fwd_dgrad_wgrad_exact(env, dt, shape):
    L, E = env
    b, d, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(abs(hash(shape)) % 2**31 + 1)
    eng = E.UNetEngine(1, 2, 4, dt)
    a = ints(g, (b, cin, d, h, w))
    wt = pow2(g, (cin, cout, 2, 2, 2))
    bias = torch.randint(-4, 5, (cout,), generator=g).double() * 0.5
    du = ints(g, (b, cout, 2 * d, 2 * h, 2 * w), -2, 2)
    n = b * d
    ad, wd, bd = ndhwc(a, dt), wt.float().cuda().contiguous(), bias.float().cuda()
    u = torch.full((2 * n, 2 * h, 2 * w, cout), float("nan"), dtype=ad.dtype, device="cuda")
    for kdi in (0, 1):
        eng._conv(E.Src(ad, cin), eng._pack(f"u#{kdi}", wd, L.PACK_DECONV3D_FPROP, cout, cin, kdi), 4 * cout, 1, n, h, w, u,
                  out_mode=L.OUT_D2S, bias=bd, oimg=(2, kdi))
    ref = F.conv_transpose3d(a, wt, bias, stride=2)
    rnd = (lambda t: t.float().to(torch.bfloat16).double()) if dt == "bf16" else (lambda t: t)
    assert torch.equal(from_ndhwc(u, b), rnd(ref)), "ConvTranspose3d forward"
    dud = ndhwc(du, dt)
    da = torch.full((n, h, w, cin), float("nan"), dtype=ad.dtype, device="cuda")
    eng._conv(E.Src(dud, cout), eng._pack("u", wd, L.PACK_DECONV3D_DGRAD, cout, cin), cin, 1, n, h, w, da, in_mode=L.IN_S2D, depth=d)
    assert torch.equal(from_ndhwc(da, b), rnd(F.conv3d(du, wt, stride=2))), "ConvTranspose3d dgrad"
    grad = torch.full((cin, cout, 2, 2, 2), float("nan"), dtype=torch.float32, device="cuda")
    db = torch.zeros(cout, dtype=torch.float32, device="cuda")
    for kdi in (0, 1):
        dwp = eng._wgrad(E.Src(ad, cin), dud, 4 * cout, 1, n, h, w, dy_mode=L.IN_S2D, dbias=db, dy_img=(2, kdi))
        L.check(L.lib().oct_unpack_wgrad3d(L.PACK_DECONV3D_FPROP, dwp.data_ptr(), grad.data_ptr(), cout, cin, kdi, 0,
                                           torch.cuda.current_stream().cuda_stream))
    ar = a.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    (F.conv_transpose3d(ar, wr, br, stride=2) * du).sum().backward()
    assert torch.equal(grad.double().cpu(), wr.grad), "ConvTranspose3d wgrad"
    assert torch.equal(db.double().cpu(), br.grad), "ConvTranspose3d bias gradient"


def test_depth_pool_matches_maxpool3d(env):
    L, E = env
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    b, c, d, h, w = 2, 8, 4, 8, 12
    a = torch.randint(0, 4, (b, c, d, h, w), generator=g).float()        # many ties: the first maximum must win
    ar = a.double().clone().requires_grad_(True)
    pr = F.max_pool3d(ar, 2)
    dout = torch.randn(pr.shape, generator=g, dtype=torch.float64)
    (pr * dout).sum().backward()
    for dt in ("f32", "bf16"):
        eng = E.UNetEngine(1, 2, 4, dt)
        an = ndhwc(a, dt)
        ones, zeros = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
        p2 = torch.empty((b * d, h // 2, w // 2, c), dtype=an.dtype, device="cuda")
        L.check(lib.oct_bn_relu_pool_fwd(eng.dt, an.data_ptr(), ones.data_ptr(), zeros.data_ptr(), p2.data_ptr(), b * d, h, w, c, st))
        out = torch.empty((b * d // 2, h // 2, w // 2, c), dtype=an.dtype, device="cuda")
        m = (h // 2) * (w // 2) * c
        L.check(lib.oct_depth_pool_fwd(eng.dt, p2.data_ptr(), out.data_ptr(), b * d // 2, m, st))
        assert torch.equal(from_ndhwc(out, b), pr.detach())
        dp2 = torch.empty_like(p2)
        dn = ndhwc(dout.float(), dt)
        L.check(lib.oct_depth_pool_bwd(eng.dt, p2.data_ptr(), dn.data_ptr(), dp2.data_ptr(), b * d // 2, m, st))
        # finish the routing inside the slices with the 2-D kernel (mean 0 / invstd 1: the BatchNorm sums are not used here)
        gout = torch.empty_like(an)
        nblk = lib.oct_dact_bn_reduce_blocks(b * d, h, w, c, 1)
        part = torch.empty((nblk, 2, c), dtype=torch.float32, device="cuda")
        shift = torch.full((c,), 1e-3, device="cuda")      # relu(a + 1e-3) keeps zeros "active" like MaxPool3d on raw values
        L.check(lib.oct_dact_bn_reduce(eng.dt, None, dp2.data_ptr(), an.data_ptr(), ones.data_ptr(), shift.data_ptr(),
                                       zeros.data_ptr(), ones.data_ptr(), gout.data_ptr(), part.data_ptr(), b * d, h, w, c, st))
        ref = ar.grad if dt == "f32" else None
        if dt == "f32":
            got = from_ndhwc(gout, b)
            assert torch.allclose(got, ref, rtol=1e-6, atol=1e-6), "MaxPool3d backward routing (first maximum in d, h, w order)"


def _case(seed, b, d, h, w, f, ncls):
    from oracle.torch_unet3d import TorchUNet3D
    from retinal_oct_image_segmentation_via_deep_learning_amd.unet3d import UNet3D
    torch.manual_seed(seed)
    model = UNet3D(1, ncls, init_features=f, compute_dtype="f32")
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm3d):
                mod.weight.copy_(1.0 + 0.3 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.2 * torch.randn(mod.bias.shape, generator=g))
    ref = TorchUNet3D(1, ncls, f)
    ref.load_state_dict(model.state_dict())
    x = torch.randn(b, 1, d, h, w, generator=g)
    t = torch.randint(0, ncls, (b, d, h, w), generator=g)
    return model, ref.double().train(), x, t


@pytest.mark.parametrize("cfg", [(0, 1, 16, 32, 32, 4, 3, 0.0), (7, 2, 16, 16, 48, 8, 2, 0.5)])
def test_f32_volumetric_unet_matches_torch_oracle(cfg):
    from oracle.torch_unet3d import loss_fn
    seed, b, d, h, w, f, ncls, w_dice = cfg
    model, ref, x, t = _case(seed, b, d, h, w, f, ncls)
    model.cuda().train()
    loss, probs = model.forward_backward(x.cuda(), t.cuda(), 1.0, w_dice, want_probs=True)
    rp = ref(x.double())
    rl = loss_fn(rp, t, 1.0, w_dice)
    rl.backward()
    p = probs.cpu().numpy()
    assert probs.shape == (b, ncls, d, h, w)
    assert np.abs(p - rp.detach().numpy()).max() < 2e-5
    top2 = np.sort(rp.detach().numpy(), axis=1)[:, -2:]
    safe = (top2[:, 1] - top2[:, 0]) > 1e-5
    assert safe.mean() > 0.99 and np.array_equal(p.argmax(1)[safe], rp.detach().numpy().argmax(1)[safe])
    np.testing.assert_allclose(loss[0].item(), rl.item(), rtol=2e-5)
    rg = dict(ref.named_parameters())
    for k, prm in model.named_parameters():
        r = rg[k].grad.numpy()
        err = np.abs(prm.grad.cpu().numpy() - r).max()
        assert err <= 3e-3 * max(np.abs(r).max(), 1e-4), (k, err, np.abs(r).max())
    sd, rsd = model.state_dict(), ref.state_dict()
    for k in sd:
        if "running" in k:
            np.testing.assert_allclose(sd[k].cpu().numpy(), rsd[k].numpy(), rtol=1e-4, atol=1e-6, err_msg=k)
    # the nn.Module path (autograd) gives the same numbers, eval mode uses the running statistics
    model.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    model.cuda().eval()
    ref.eval()
    with torch.no_grad():
        pe = model(x.cuda()).cpu().numpy()
        assert np.abs(pe - ref(x.double()).numpy()).max() < 2e-5
        assert torch.equal(model.predict(x.cuda()).cpu(), torch.from_numpy(pe).argmax(1))
    with pytest.raises(RuntimeError, match="Sizes of tensors must match"):
        model(torch.zeros(1, 1, 8, 32, 32, device="cuda"))


def test_bf16_volumetric_unet_trains():
    from retinal_oct_image_segmentation_via_deep_learning_amd.optim import FusedSGD
    model, ref, x, t = _case(3, 1, 16, 32, 64, 32, 4)
    model.set_compute_dtype("bf16").cuda().train()
    rl = torch.nn.functional.nll_loss(torch.log(ref(x.double())), t).item()
    opt = FusedSGD(list(model.named_parameters()), lr=0.05, momentum=0.9)
    losses = []
    for _ in range(4):
        losses.append(float(model.forward_backward(x.cuda(), t.cuda())[0]))
        opt.step()
    assert abs(losses[0] - rl) < 0.03 * rl, (losses[0], rl)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_deterministic_mode_matches_atomics_on_the_volumetric_network():
    """engine.deterministic: the transposed 3-D convolutions unpack through oct_unpack_wgrad3d (one slab), so their weight
    gradients must stay on atomics -- with per-workgroup partial slabs only the first strip's sum came back."""
    model, ref, x, t = _case(5, 1, 16, 32, 32, 8, 3)
    model.cuda().train()
    grads = {}
    keep = model._engine.deterministic
    try:
        for det in (False, True):
            model._engine.deterministic = det
            model.forward_backward(x.cuda(), t.cuda(), 1.0, 0.0)
            grads[det] = {k: p.grad.clone() for k, p in model.named_parameters()}
    finally:
        model._engine.deterministic = keep
    for k in grads[False]:
        a, b = grads[False][k].double(), grads[True][k].double()
        assert float((a - b).abs().max()) <= 1e-5 * max(float(a.abs().max()), 1e-6), k
