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

    @staticmethod
    def set_dropout_masks(masks):
        """ops.DROPOUT_MASK_HOOK: the fixture's keep masks in forward order (rewound by every call of this function)"""
        from retinal_oct_image_segmentation_via_deep_learning_amd import ops
        if not masks:
            ops.DROPOUT_MASK_HOOK[0] = None
            return
        at = [0]

        def hook(n, c, p):
            m = masks[at[0] % len(masks)]
            at[0] += 1
            assert tuple(m.shape) == (n, c)
            return m
        ops.DROPOUT_MASK_HOOK[0] = hook


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


NETS = [("attunet_c3_2x32x48", "AttU_Net", dict(channels=[4, 8, 16, 32, 64])), ("sd_unet_c2_1x32x32", "U_Net", {}),
        ("attunet4_c3_2x24x40", "AttU_Net4", dict(channels=[4, 8, 16, 32]))]


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
        m(torch.zeros(1, cin, 20, 36, device="cuda"))       # not divisible by 16 (8 for AttU_Net4): the reference raises too


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


@pytest.mark.parametrize("dtype,c,k", [("f32", 32, 1), ("bf16", 32, 1), ("bf16", 64, 3), ("f32", 8, 4), ("bf16", 512, 1), ("f32", 64, 2),
                                       ("bf16", 512, 4), ("bf16", 64, 10), ("f32", 64, 11), ("bf16", 128, 12), ("bf16", 32, 5), ("bf16", 64, 7), ("bf16", 16, 9)])
def test_rowdot_kernels_match_numpy(dtype, c, k):
    """oct_rowdot_{fwd,bwd_data,bwd_weight} (1x1 convolutions with 1-12 output channels: Attention_block.psi,
    common.py:79-83, the heads Conv_1x1, unet.py:38,113, ReLayNet's 10-class classifier, ReLayNet_2017.py:118-126) against numpy on a pixel count that is not a multiple of
    anything; integer-valued operands make every product and sum exact, so the comparison is bit-for-bit."""
    from retinal_oct_image_segmentation_via_deep_learning_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(c + k)
    npix = 3 * 37 * 53 + 1
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    dt = L.DT_F32 if dtype == "f32" else L.DT_BF16
    x = rng.integers(-1, 2, (npix, c)).astype(np.float32)
    w = (rng.integers(-1, 2, (k, c)) * (rng.random((k, c)) < 64.0 / c)).astype(np.float32)   # |y| <= 64-ish: exact in bf16
    dy = rng.integers(-2, 3, (npix, k)).astype(np.float32)
    X, W, DY = torch.from_numpy(x).to("cuda", tdt), torch.from_numpy(w).cuda(), torch.from_numpy(dy).to("cuda", tdt)
    assert lib.oct_rowdot_ok(c, k) == 1 and lib.oct_rowdot_ok(24, 1) == 0 and lib.oct_rowdot_ok(1024, 1) == 0 and lib.oct_rowdot_ok(64, 13) == 0 and lib.oct_rowdot_ok(256, 5) == 0
    nblk = lib.oct_rowdot_blocks(npix, c)
    Y = torch.empty((npix, k), dtype=tdt, device="cuda")
    ST = torch.empty((nblk, 2, k), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.oct_rowdot_fwd(dt, X.data_ptr(), W.data_ptr(), Y.data_ptr(), ST.data_ptr(), npix, c, k, st))
    y = x @ w.T
    assert np.abs(y).max() <= 256        # 8 significant bits: representable in bf16
    assert np.array_equal(Y.float().cpu().numpy(), y)
    s = ST.sum(0).cpu().numpy()
    assert np.array_equal(s[0], y.sum(0)) and np.array_equal(s[1], (y * y).sum(0))
    # without statistics (a class head): 5-12 classes in bf16 with c % 32 == 0 take the matrix-pipe forward kernel
    Y2 = torch.full((npix, k), float("nan"), dtype=tdt, device="cuda")
    L.check(lib.oct_rowdot_fwd(dt, X.data_ptr(), W.data_ptr(), Y2.data_ptr(), None, npix, c, k, st))
    assert np.array_equal(Y2.float().cpu().numpy(), y)
    DX = torch.empty_like(X)
    L.check(lib.oct_rowdot_bwd_data(dt, DY.data_ptr(), W.data_ptr(), DX.data_ptr(), npix, c, k, st))
    assert np.array_equal(DX.float().cpu().numpy(), dy @ w)
    DW = torch.empty((k, c), dtype=torch.float32, device="cuda")
    SC = torch.empty((nblk, k * c), dtype=torch.float32, device="cuda")
    L.check(lib.oct_rowdot_bwd_weight(dt, DY.data_ptr(), X.data_ptr(), DW.data_ptr(), SC.data_ptr(), npix, c, k, 0, st))
    assert np.array_equal(DW.cpu().numpy(), dy.T @ x)
    # the same pass with the bias gradient riding on it (class heads): dw unchanged, db = column sums of dy, accumulate mode adds
    SCB = torch.empty((nblk, k * c + k), dtype=torch.float32, device="cuda")
    DW2 = torch.full((k, c), float("nan"), dtype=torch.float32, device="cuda")
    DB = torch.full((k,), float("nan"), dtype=torch.float32, device="cuda")
    L.check(lib.oct_rowdot_bwd_weight_bias(dt, DY.data_ptr(), X.data_ptr(), DW2.data_ptr(), DB.data_ptr(), SCB.data_ptr(), npix, c, k, 0, st))
    assert np.array_equal(DW2.cpu().numpy(), dy.T @ x) and np.array_equal(DB.cpu().numpy(), dy.sum(0))
    L.check(lib.oct_rowdot_bwd_weight_bias(dt, DY.data_ptr(), X.data_ptr(), DW2.data_ptr(), DB.data_ptr(), SCB.data_ptr(), npix, c, k, 1, st))
    assert np.array_equal(DB.cpu().numpy(), 2 * dy.sum(0))
    L.check(lib.oct_rowdot_bwd_weight(dt, DY.data_ptr(), X.data_ptr(), DW.data_ptr(), SC.data_ptr(), npix, c, k, 1, st))
    assert np.array_equal(DW.cpu().numpy(), 2 * (dy.T @ x))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_attention_gate_rowdot_path_matches_padded_gemm_path(dtype):
    """Attention_block(F_g = F_l = 64, F_int = 32): psi on the streaming kernels vs the same module with the switch off
    (psi as a 1 -> 32 padded GEMM on the MFMA kernels): output, input gradients, every parameter gradient, BN buffers."""
    from retinal_oct_image_segmentation_via_deep_learning_amd import ops
    torch.manual_seed(3)
    m = Dropins.Attention_block(F_g=64, F_l=64, F_int=32, compute_dtype=dtype).cuda().train()
    g0, x0 = torch.randn(2, 64, 24, 40, device="cuda"), torch.randn(2, 64, 24, 40, device="cuda")
    r = torch.randn(2, 64, 24, 40, device="cuda")
    res = {}
    e = ops.kernels(dtype)
    init = {k: v.clone() for k, v in m.state_dict().items()}
    for off in (False, True):
        m.load_state_dict(init)
        m.zero_grad(set_to_none=True)
        e.rowdot_off = off
        try:
            g, x = g0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
            out = m(g, x)
            (out * r).sum().backward()
        finally:
            e.rowdot_off = False
        res[off] = dict(out=out.detach(), gg=g.grad, gx=x.grad, **{k: p.grad for k, p in m.named_parameters()},
                        **{"b/" + k: v.clone() for k, v in m.state_dict().items() if "running" in k})
    rel = 2e-4 if dtype == "f32" else 2e-2     # fp32: the two paths only differ in summation order
    for k in res[False]:
        a, b = res[False][k].float(), res[True][k].float()
        tol = rel * max(float(b.abs().max()), 1e-6)
        assert float((a - b).abs().max()) <= tol, (k, float((a - b).abs().max()), tol)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("cls,kw,shape", [("AttU_Net", dict(channels=[16, 32, 64, 128, 256]), (2, 1, 32, 48)),
                                          ("U_Net", {}, (1, 1, 32, 32))])
def test_deferred_activation_schedule_is_bit_identical_to_the_materialised_one(dtype, cls, kw, shape):
    """ops.LazyAct (BN + ReLU and bias adds applied by the consuming convolution, never written) against the same
    network with every activation materialised (ops.LAZY[0] = False): logits, input gradient and -- with the ordered
    weight-gradient reduction -- every parameter gradient are EQUAL, in fp32 and in bf16."""
    from retinal_oct_image_segmentation_via_deep_learning_amd import ops
    from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.SD_Layer_Net import unet as U
    torch.manual_seed(11)
    m = getattr(U, cls)(shape[1], 3, compute_dtype=dtype, **kw).cuda().train()
    x0 = torch.randn(*shape, device="cuda")
    t = torch.randint(0, 3, (shape[0], shape[2], shape[3]), device="cuda")
    init = {k: v.clone() for k, v in m.state_dict().items()}
    e = ops.kernels(dtype)
    res = {}
    keep_det, keep_lazy = e.deterministic, ops.LAZY[0]
    try:
        e.deterministic = True
        for setting in (True, False):
            ops.LAZY[0] = setting
            m.load_state_dict(init)
            m.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            out = m(x)
            F.cross_entropy(out, t).backward()
            res[setting] = dict(out=out.detach().clone(), gx=x.grad.clone(), **{k: p.grad.clone() for k, p in m.named_parameters()},
                                **{"b/" + k: v.clone() for k, v in m.state_dict().items() if "running" in k})
    finally:
        e.deterministic, ops.LAZY[0] = keep_det, keep_lazy
    # (these bias gradients go through oct_channel_sum's atomics -- convolutions without BN whose Cin is not a multiple of
    # 32 -- and are equal up to summation order)
    loose = ("Conv_1x1.bias", "Conv1.init_conv.bias", "Conv2.init_conv.bias")   # Conv2: Cin = 16 in the narrow AttU_Net case
    bad = [k for k in res[True] if not (torch.allclose(res[True][k], res[False][k], rtol=1e-4, atol=1e-7)
                                        if k in loose else torch.equal(res[True][k], res[False][k]))]
    assert bad == [], bad


@pytest.mark.parametrize("name", list(BLOCKS))
def test_f32_block_backward_through_frozen_batchnorm_matches_torch(golden_dir, name):
    """eval() + backward (frozen-BN fine-tuning): BatchNorm on running statistics is a per-channel affine; every gradient,
    conv biases in front of the BatchNorm included, against the stock-torch restatement (oracle/torch_blocks.py, pinned to
    the reference's fixtures by tests/test_oracle_blocks.py) in float64."""
    from oracle import torch_blocks as TB
    z, m, xs = load_block(golden_dir, name, Dropins)
    _, ref, _ = load_block(golden_dir, name, TB)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for a, b in zip(m.modules(), ref.modules()):
            if isinstance(a, torch.nn.BatchNorm2d):
                a.running_mean.copy_(0.3 * torch.randn(a.running_mean.shape, generator=g))
                a.running_var.copy_(0.5 + torch.rand(a.running_var.shape, generator=g))
                b.running_mean.copy_(a.running_mean)
                b.running_var.copy_(a.running_var)
    m.set_compute_dtype("f32").cuda().eval()
    ref = ref.double().eval()
    xd = [x.cuda().requires_grad_(True) for x in xs]
    xr = [x.double().requires_grad_(True) for x in xs]
    out, rout = m(*xd), ref(*xr)
    close(out.detach().cpu().numpy(), rout.detach().numpy(), "out", 2e-5, 1.0)
    r = torch.from_numpy(z["r"])
    (out * r.cuda()).sum().backward()
    (rout * r.double()).sum().backward()
    for i, (a, b) in enumerate(zip(xd, xr)):
        close(a.grad.cpu().numpy(), b.grad.numpy(), f"gx{i}", 2e-3)
    rg = dict(ref.named_parameters())
    for k, p in m.named_parameters():
        close(p.grad.cpu().numpy(), rg[k].grad.numpy(), k, 2e-3)
