"""GPU parity of the MGU-Net drop-ins (MGUNet_2021.py: Basconv, GloRe_Unit, MGR_Module, MGUNet, MGUNet_2) against the
fixtures made from the reference's own classes (tools/gen_golden_mgunet.py).  fp32 parity mode: outputs within 2e-5 of their
scale, gradients 2e-3 of each tensor's max, arg-max maps identical; bf16 production mode: documented looser bounds."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.cases import bio_case, bio_grad_errors, bio_weights_match
from test_gpu_blocks import close
from test_oracle_mgunet import MG_BLOCKS, MG_NETS, load_mg_block, logits_close

pytestmark = pytest.mark.gpu


def _M():
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment import MGUNet_2021 as M
    return M


@pytest.mark.parametrize("name", list(MG_BLOCKS))
def test_f32_block_matches_reference_fixture(golden_dir, name):
    z, m, xs = load_mg_block(golden_dir, name, _M())
    m.set_compute_dtype("f32").cuda()
    x = xs[0].cuda().requires_grad_(True)
    out = m(x)
    assert out.dtype == torch.float32 and tuple(out.shape) == z["out"].shape
    close(out.detach().cpu().numpy(), z["out"], "out", 2e-5, 1.0)
    (out * torch.from_numpy(z["r"]).cuda()).sum().backward()
    close(x.grad.cpu().numpy(), z["gx0"], "gx0", 2e-3)
    for k, p in m.named_parameters():
        if float(np.abs(z["g/" + k]).max()) < 1e-9:
            # analytically zero: a conv bias in front of a train-mode BN, and conv_extend's bias (a per-channel constant that
            # the 1x1 fusion conv + BN after the concatenation cancels) -- rounding noise on both sides
            assert float(p.grad.abs().max()) < 2e-5, k
            continue
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
        close(m(xs[0].cuda()).cpu().numpy(), z["out_eval"], "out_eval", 2e-5, 1.0)


@pytest.mark.parametrize("name", list(MG_BLOCKS))
def test_bf16_block_is_close(golden_dir, name):
    z, m, xs = load_mg_block(golden_dir, name, _M())
    m.set_compute_dtype("bf16").cuda()
    x = xs[0].cuda().requires_grad_(True)
    out = m(x)
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


@pytest.mark.parametrize("name,cls", MG_NETS)
def test_f32_network_matches_reference_fixture(golden_dir, name, cls):
    M = _M()
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    seed, n, cin, ncls, h, w = (int(v) for v in z["meta"])
    m, x, t = bio_case(lambda ci, nc: getattr(M, cls)(ci, nc, feature_scale=16, compute_dtype="f32"), seed, n, cin, ncls, h, w)
    assert bio_weights_match(z, m.state_dict())
    m.cuda()
    out = m(x.cuda())
    lg = out.detach().cpu().numpy()
    assert np.array_equal(lg.argmax(1), logits_close(z, "logits", lg, 2e-5))
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
        logits_close(z, "logits_eval", m(x.cuda()).cpu().numpy(), 2e-5)


def test_bf16_mgunet2_tracks_reference_and_trains(golden_dir):
    M = _M()
    z = np.load(os.path.join(golden_dir, "mgunet2_c3_2x48x64.npz"))
    seed, n, cin, ncls, h, w = (int(v) for v in z["meta"])
    m, x, t = bio_case(lambda ci, nc: M.MGUNet_2(ci, nc, feature_scale=16), seed, n, cin, ncls, h, w)
    m.cuda()
    out = m(x.cuda())
    assert (out.argmax(1).cpu().numpy() == z["logits"].argmax(1)).mean() > 0.9
    loss = F.cross_entropy(out, t.cuda())
    assert abs(float(loss.detach()) - float(z["loss"][0])) < 5e-2
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    first = last = None
    for _ in range(6):
        opt.zero_grad()
        l = F.cross_entropy(m(x.cuda()), t.cuda())
        l.backward()
        opt.step()
        first = float(l.detach()) if first is None else first
        last = float(l.detach())
    assert last < first


def test_default_width_mgunet2_runs_the_pipelined_kernels():
    """feature_scale=4 (the reference default): 16..128 channels -- the 32-multiple layers take igemm2 / wgrad2 in bf16."""
    M = _M()
    torch.manual_seed(0)
    m = M.MGUNet_2(1, 11).cuda()
    x = torch.randn(2, 1, 64, 96, device="cuda")
    t = torch.randint(0, 11, (2, 64, 96), device="cuda")
    out = m(x)
    assert tuple(out.shape) == (2, 11, 64, 96) and torch.isfinite(out).all()
    ref = m.set_compute_dtype("f32")(x)
    m.set_compute_dtype("bf16")
    assert float((out - ref).detach().abs().max()) < 0.08 * max(1.0, float(ref.detach().abs().max()))
    F.cross_entropy(out, t).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_error_paths_match_the_reference(golden_dir):
    M = _M()
    z = np.load(os.path.join(golden_dir, "mgunet_api.npz"))
    m = M.MGUNet_2(1, 3, feature_scale=16).cuda()
    with pytest.raises(RuntimeError, match="Output size is too small"):
        m(torch.zeros(2, 1, 32, 32, device="cuda"))     # bottleneck 4 x 4 < the 5 x 5 pooling (reference: same error)
    with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
        m(torch.zeros(1, 1, 48, 64, device="cuda"))     # one image: the 5 x 5 branch is one pixel, train-mode BN refuses
    assert "Expected more than 1 value" in str(z["single_msg"])


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-5), ("bf16", 2e-2)])
@pytest.mark.parametrize("shape,size", [((2, 3, 4, 8), (11, 17)), ((1, 1, 1, 16), (6, 8)), ((2, 7, 5, 8), (7, 5)), ((1, 4, 6, 24), (4, 6))])
def test_bilinear_resize_matches_torch_both_ways(dtype, tol, shape, size):
    from retinal_oct_image_segmentation_via_deep_learning_amd import ops
    n, h, w, c = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, h, w, c, generator=g)
    r = torch.randn(n, size[0], size[1], c, generator=g)
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    xd = x.to(tdt).cuda().requires_grad_(True)
    out = ops.BilinearResize.apply(dtype, size, xd)
    (out.float() * r.cuda()).sum().backward()
    xr = x.to(tdt).double().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.interpolate(xr, size=size, mode="bilinear", align_corners=True)
    (ref * r.double().permute(0, 3, 1, 2)).sum().backward()
    assert float((out.detach().float().cpu().double() - ref.detach().permute(0, 2, 3, 1)).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    gref = xr.grad.permute(0, 2, 3, 1)
    assert float((xd.grad.float().cpu().double() - gref).abs().max()) <= tol * max(1.0, float(gref.abs().max())) * 4


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("k,h,w", [(3, 11, 17), (5, 11, 17), (2, 7, 9), (4, 8, 12)])
def test_floor_mode_maxpool_matches_torch(dtype, k, h, w):
    from retinal_oct_image_segmentation_via_deep_learning_amd import ops
    g = torch.Generator().manual_seed(k)
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    x = torch.randn(2, h, w, 8, generator=g).to(tdt)
    r = torch.randn(2, h // k, w // k, 8, generator=g).to(tdt)
    xd = x.cuda().requires_grad_(True)
    out = ops.MaxPool.apply(dtype, k, xd)
    out.backward(r.cuda())
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.max_pool2d(xr, k)
    ref.backward(r.float().permute(0, 3, 1, 2))
    assert torch.equal(out.float().cpu(), ref.detach().permute(0, 2, 3, 1))
    assert torch.equal(xd.grad.float().cpu(), xr.grad.permute(0, 2, 3, 1))


@pytest.mark.parametrize("shape", [(2, 12, 64, 16, 0, 16), (1, 9, 40, 16, 16, 16), (1, 16, 96, 16, 0, 32), (2, 8, 32, 48, 16, 16),
                                   (1, 6, 8, 32, 0, 16)])
def test_sixteen_channel_conv3x3_runs_pixel_pair_folded_and_is_bit_exact(shape):
    """MGU-Net's 16-channel levels (MGUNet_2021.py:118-126 with feature_scale 4): a 3x3 convolution whose channel counts are
    multiples of 16 but not all of 32 runs as the pixel-pair-FOLDED convolution on the pipelined kernels (ops.fold16_ok) --
    output, BatchNorm statistics, input gradients (concat split) and the un-folded weight gradient against torch's float64
    convolution on exactly representable operands: bit equality (odd and even tile counts, ragged widths, concat sources)."""
    from retinal_oct_image_segmentation_via_deep_learning_amd import ops, _lib as L
    n, h, w, c0, c1, cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    e = ops.kernels("bf16")
    assert ops.fold16_ok(e, 9, None, w, c0, c1, cout) and not ops.fold16_ok(e, 9, None, w + 1, c0, c1, cout)
    cin = c0 + c1
    x = torch.randint(-3, 4, (n, cin, h, w), generator=g).double()
    wt = (torch.exp2(torch.randint(-3, 1, (cout, cin, 3, 3), generator=g).double()) * (torch.randint(0, 2, (cout, cin, 3, 3), generator=g) * 2 - 1)
          * (torch.rand((cout, cin, 3, 3), generator=g) < 0.4))
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1, bias=False).cuda()
    with torch.no_grad():
        conv.weight.copy_(wt.float())
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()
    a0 = nhwc(x[:, :c0]).requires_grad_(True)
    a1 = nhwc(x[:, c0:]).requires_grad_(True) if c1 else None
    bn = torch.nn.BatchNorm2d(cout).cuda().train()
    y, scale, shift = ops.ConvAffineAct.apply("bf16", bn, L.ACT_RELU, a0, a1, conv.weight, None, bn.weight, bn.bias, None, None, None,
                                              None, "relu")       # the raw conv output of a deferred BN + ReLU layer
    ref = F.conv2d(x, wt, padding=1)
    rnd = lambda t: t.float().to(torch.bfloat16).double()
    assert torch.equal(y.detach().float().cpu().double().permute(0, 3, 1, 2), rnd(ref)), "folded forward"
    mean = ref.mean(dim=(0, 2, 3))
    torch.testing.assert_close(bn.running_mean.double().cpu(), 0.1 * mean, rtol=1e-5, atol=1e-6)
    # backward of the raw output alone: a second, bare (no BatchNorm) application gives dx / dW without the BN terms
    a0b = nhwc(x[:, :c0]).requires_grad_(True)
    a1b = nhwc(x[:, c0:]).requires_grad_(True) if c1 else None
    yb = ops.conv_bn_act("bf16", a0b, conv, x1=a1b)
    assert torch.equal(yb.detach().float().cpu().double().permute(0, 3, 1, 2), rnd(ref))
    dy = torch.randint(-2, 3, (n, cout, h, w), generator=g).double() * (torch.rand((n, cout, h, w), generator=g) < 0.5)
    conv.weight.grad = None
    yb.backward(nhwc(dy))
    dx = torch.nn.grad.conv2d_input((n, cin, h, w), wt, dy, padding=1)
    dw = torch.nn.grad.conv2d_weight(x, (cout, cin, 3, 3), dy, padding=1)
    assert torch.equal(a0b.grad.float().cpu().double().permute(0, 3, 1, 2), rnd(dx[:, :c0])), "folded dgrad, source 0"
    if c1:
        assert torch.equal(a1b.grad.float().cpu().double().permute(0, 3, 1, 2), rnd(dx[:, c0:])), "folded dgrad, source 1"
    assert dw.abs().max() * 2 < 2 ** 23
    assert torch.equal(conv.weight.grad.double().cpu(), dw), "un-folded weight gradient"


@pytest.mark.parametrize("shape", [(2, 12, 64, 16, 0, 16), (1, 9, 40, 16, 16, 32), (1, 5, 8, 48, 0, 16)])
def test_sixteen_channel_conv1x1_runs_pixel_pair_folded_and_is_bit_exact(shape):
    """the same for 1x1 convolutions (the folded filter is [[W, 0], [0, W]]: the two column parities do not mix)"""
    from retinal_oct_image_segmentation_via_deep_learning_amd import ops
    n, h, w, c0, c1, cout = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    e = ops.kernels("bf16")
    assert ops.fold16_ok(e, 1, None, w, c0, c1, cout)
    cin = c0 + c1
    x = torch.randint(-3, 4, (n, cin, h, w), generator=g).double()
    wt = torch.exp2(torch.randint(-3, 1, (cout, cin, 1, 1), generator=g).double()) * (torch.randint(0, 2, (cout, cin, 1, 1), generator=g) * 2 - 1)
    conv = torch.nn.Conv2d(cin, cout, 1, bias=False).cuda()
    with torch.no_grad():
        conv.weight.copy_(wt.float())
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()
    a0 = nhwc(x[:, :c0]).requires_grad_(True)
    a1 = nhwc(x[:, c0:]).requires_grad_(True) if c1 else None
    y = ops.conv_bn_act("bf16", a0, conv, x1=a1)
    ref = F.conv2d(x, wt)
    rnd = lambda t: t.float().to(torch.bfloat16).double()
    assert torch.equal(y.detach().float().cpu().double().permute(0, 3, 1, 2), rnd(ref)), "folded 1x1 forward"
    dy = torch.randint(-2, 3, (n, cout, h, w), generator=g).double()
    y.backward(nhwc(dy))
    dx = torch.nn.grad.conv2d_input((n, cin, h, w), wt, dy)
    dw = torch.nn.grad.conv2d_weight(x, (cout, cin, 1, 1), dy)
    assert torch.equal(a0.grad.float().cpu().double().permute(0, 3, 1, 2), rnd(dx[:, :c0]))
    if c1:
        assert torch.equal(a1.grad.float().cpu().double().permute(0, 3, 1, 2), rnd(dx[:, c0:]))
    assert torch.equal(conv.weight.grad.double().cpu(), dw), "un-folded 1x1 weight gradient"


@pytest.mark.parametrize("shape", [(2, 6, 32, 32, 16), (1, 5, 8, 16, 16), (1, 8, 64, 48, 16), (2, 4, 12, 16, 32)])
def test_sixteen_channel_deconv_runs_pixel_pair_folded_and_is_bit_exact(shape):
    """ConvTranspose2d(k = 2, s = 2) with 16-channel sides (MGU-Net's UnetUp at feature_scale 4, MGUNet_2021.py:72-89) as the folded
    transposed convolution (ops.fold16_deconv_ok): output with bias, input gradient, un-folded weight gradient and bias gradient
    against torch's float64 conv_transpose2d on exactly representable operands."""
    from retinal_oct_image_segmentation_via_deep_learning_amd import ops
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(sum(shape) + 3)
    e = ops.kernels("bf16")
    assert ops.fold16_deconv_ok(e, 2, w, cin, cout)
    x = torch.randint(-3, 4, (n, cin, h, w), generator=g).double()
    wt = torch.exp2(torch.randint(-3, 1, (cin, cout, 2, 2), generator=g).double()) * (torch.randint(0, 2, (cin, cout, 2, 2), generator=g) * 2 - 1)
    b = torch.randint(-2, 3, (cout,), generator=g).double() * 0.5
    a = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda().requires_grad_(True)
    wd_ = wt.float().cuda().requires_grad_(True)
    bd = b.float().cuda().requires_grad_(True)
    out = ops.Deconv.apply("bf16", a, wd_, bd)
    ref = F.conv_transpose2d(x, wt, b, stride=2)
    rnd = lambda t: t.float().to(torch.bfloat16).double()
    assert torch.equal(out.detach().float().cpu().double().permute(0, 3, 1, 2), rnd(ref)), "folded deconv forward"
    dy = torch.randint(-2, 3, ref.shape, generator=g).double()
    out.backward(dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda())
    xr = x.clone().requires_grad_(True); wr = wt.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    (F.conv_transpose2d(xr, wr, br, stride=2) * dy).sum().backward()
    assert torch.equal(a.grad.float().cpu().double().permute(0, 3, 1, 2), rnd(xr.grad)), "folded deconv dgrad"
    assert torch.equal(wd_.grad.double().cpu(), wr.grad), "un-folded deconv weight gradient"
    assert torch.equal(bd.grad.double().cpu(), br.grad), "deconv bias gradient"
