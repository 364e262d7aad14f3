"""GPU parity of every HIP kernel against the CPU oracle (oracle/ref_cpu.py), through the C ABI.

f32 = parity mode (exact fp32 MFMA, tolerance ~1e-5); bf16 = production mode, checked against the
oracle evaluated on bf16-rounded operands (tolerance: one bf16 rounding of the result, 2^-8 rel.).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu

DTYPES = ["f32", "bf16"]


@pytest.fixture(scope="module")
def env():
    from retinal_oct_image_segmentation_via_deep_learning_amd import _lib as L
    from retinal_oct_image_segmentation_via_deep_learning_amd import engine as E
    L.lib()
    return L, E


def tdt(dt):
    return torch.float32 if dt == "f32" else torch.bfloat16


def rnd(a, dt):
    """value as stored in the activation dtype"""
    if dt == "f32":
        return np.asarray(a, dtype=np.float32)
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def dev(a_nchw, dt):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a_nchw, np.float32).transpose(0, 2, 3, 1))).to("cuda", tdt(dt))


def host(t_nhwc):
    return t_nhwc.float().cpu().numpy().transpose(0, 3, 1, 2)


def fdev(a):
    return torch.from_numpy(np.asarray(a, np.float32)).cuda().contiguous()


def close(got, ref, dt, what, scale_tol=None):
    ref = np.asarray(ref, np.float64)
    got = np.asarray(got, np.float64)
    mag = max(float(np.abs(ref).max()), 1e-6)
    tol = (3e-5 if dt == "f32" else 1.2e-2) * mag if scale_tol is None else scale_tol * mag
    err = float(np.abs(got - ref).max())
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    assert err <= tol, f"{what}: max abs err {err:.3e} > {tol:.3e} (|ref|max {mag:.3e}) at {np.unravel_index(np.abs(got-ref).argmax(), ref.shape)}"


CONV_SHAPES = [
    # n, h, w, c0, c1, cout
    (2, 16, 32, 16, 0, 32),
    (1, 24, 40, 4, 0, 8),      # scalar channel path, partial tiles
    (2, 8, 64, 32, 0, 64),
    (1, 16, 32, 64, 0, 128),
    (1, 8, 32, 48, 0, 160),    # cout not a multiple of the 128 tile
    (2, 16, 32, 8, 8, 16),     # virtual concat, vector path
    (1, 16, 16, 4, 4, 4),      # virtual concat, scalar path (tiny fixture shapes)
    (1, 2, 2, 32, 0, 64),      # bottleneck of the 32x32 fixtures
    (1, 16, 32, 3, 0, 8),      # in_channels = 3
    # regular shapes -> pipelined bf16 kernel (igemm2.hip); fp32 stays on the generic kernel
    (2, 16, 64, 32, 0, 32),    # weights resident, Cout 32
    (1, 24, 32, 32, 0, 64),    # weights resident, Cout 64
    (2, 8, 64, 64, 0, 32),     # two chunks, streamed weights
    (1, 16, 32, 32, 32, 32),   # virtual concat (dec1conv1)
    (1, 8, 32, 64, 64, 64),    # virtual concat, Cout 64
    (1, 16, 32, 64, 0, 128),
    (1, 8, 32, 128, 128, 256), # two channel blocks of 128, concat
    (3, 8, 32, 96, 0, 192),    # Cout multiple of 64 only
    # Cout % 128 == 0, H % 16 == 0
    (2, 32, 64, 64, 64, 128),  # several tiles per image, concat, dgrad splits 64/64
    (1, 16, 64, 128, 0, 256),  # two channel blocks
    (3, 16, 32, 32, 0, 128),   # two stages per item only
    (1, 48, 32, 96, 32, 128),  # three tile rows: interior tile has no zero border
    # ragged sizes on the pipelined kernels (H % 8 != 0, W % 32 != 0): last tiles predicated
    (2, 31, 48, 32, 0, 64),    # AttU_Net's deepest level shape (31 x 48)
    (1, 62, 96, 32, 32, 32),   # H ragged only, concat, TH = 8 path for Cout = 32
    (1, 12, 40, 64, 0, 128),   # both ragged, Cout 128
    (2, 16, 64, 1, 0, 32),     # first layer, in_channels = 1: direct stencil kernels (bf16)
    (1, 24, 40, 1, 0, 16),
]


def make_src(E, rng, dt, n, h, w, c0, c1, xform):
    """random sources + BN coefficients; returns (Src, effective NCHW input as the conv sees it)"""
    x0 = rnd(rng.standard_normal((n, c0, h, w)), dt)
    parts = []
    bn0 = bn1 = None
    if xform:
        s0, b0 = rng.uniform(0.5, 1.5, c0) * rng.choice([-1, 1], c0), rng.standard_normal(c0) * 0.3
        bn0 = E.BNState(fdev(s0), fdev(b0))
        parts.append(np.maximum(x0 * np.float32(s0)[None, :, None, None] + np.float32(b0)[None, :, None, None], 0))
    else:
        parts.append(x0)
    x1d = None
    if c1:
        x1 = rnd(rng.standard_normal((n, c1, h, w)), dt)
        s1, b1 = rng.uniform(0.5, 1.5, c1), rng.standard_normal(c1) * 0.3
        bn1 = E.BNState(fdev(s1), fdev(b1))
        parts.append(np.maximum(x1 * np.float32(s1)[None, :, None, None] + np.float32(b1)[None, :, None, None], 0))
        x1d = dev(x1, dt)
    eff = np.concatenate(parts, axis=1)
    return E.Src(dev(x0, dt), c0, bn0, x1d, c1, bn1), rnd(eff, dt)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", CONV_SHAPES)
@pytest.mark.parametrize("xform", [False, True])
def test_conv3x3_fprop_and_stats(env, dt, shape, xform):
    L, E = env
    n, h, w, c0, c1, cout = shape
    rng = np.random.default_rng(hash((shape, xform)) % 2**32)
    eng = E.UNetEngine(1, 2, 4, dt)
    src, eff = make_src(E, rng, dt, n, h, w, c0, c1, xform)
    wt = (rng.standard_normal((cout, c0 + c1, 3, 3)) / np.sqrt(9 * (c0 + c1))).astype(np.float32)
    wd = fdev(wt)
    wp = eng._pack("w", wd, L.PACK_CONV_FPROP, cout, c0 + c1)
    y = torch.full((n, h, w, cout), float("nan"), dtype=tdt(dt), device="cuda")
    nblk = eng._stat_blocks(cout, n, h, w, src)
    part = torch.full((nblk, 2, cout), float("nan"), dtype=torch.float32, device="cuda")
    eng._conv(src, wp, cout, 9, n, h, w, y, stats=part)
    torch.cuda.synchronize()
    ref = O.conv3x3_fwd(eff.astype(np.float64), rnd(wt, dt).astype(np.float64))
    close(host(y), ref, dt, "conv3x3 fprop")
    s = part.double().sum(0).cpu().numpy()
    close(s[0], ref.sum(axis=(0, 2, 3)), dt, "sum(y)", scale_tol=2e-3)
    close(s[1], (ref ** 2).sum(axis=(0, 2, 3)), dt, "sum(y^2)", scale_tol=2e-3)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv3x3_dgrad_split(env, dt, shape):
    L, E = env
    n, h, w, c0, c1, cout = shape
    rng = np.random.default_rng(hash(shape) % 2**32 + 1)
    eng = E.UNetEngine(1, 2, 4, dt)
    cin = c0 + c1
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cout)).astype(np.float32)
    dy = rnd(rng.standard_normal((n, cout, h, w)), dt)
    wp = eng._pack("w", fdev(wt), L.PACK_CONV_DGRAD, cout, cin)
    d0 = torch.full((n, h, w, c0), float("nan"), dtype=tdt(dt), device="cuda")
    d1 = torch.full((n, h, w, c1), float("nan"), dtype=tdt(dt), device="cuda") if c1 else None
    eng._conv(E.Src(dev(dy, dt), cout), wp, cin, 9, n, h, w, d0, y1=d1, split=c0 if c1 else 0)
    torch.cuda.synchronize()
    dx, _ = O.conv3x3_bwd(np.zeros((n, cin, h, w)), rnd(wt, dt).astype(np.float64), dy.astype(np.float64))
    close(host(d0), dx[:, :c0], dt, "dgrad part 0")
    if c1:
        close(host(d1), dx[:, c0:], dt, "dgrad part 1")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv3x3_wgrad(env, dt, shape):
    L, E = env
    n, h, w, c0, c1, cout = shape
    rng = np.random.default_rng(hash(shape) % 2**32 + 2)
    eng = E.UNetEngine(1, 2, 4, dt)
    src, eff = make_src(E, rng, dt, n, h, w, c0, c1, True)
    dy = rnd(rng.standard_normal((n, cout, h, w)), dt)
    dwp = eng._wgrad(src, dev(dy, dt), cout, 9, n, h, w)
    grad = torch.full((cout, c0 + c1, 3, 3), float("nan"), dtype=torch.float32, device="cuda")
    eng._unpack(L.PACK_CONV_FPROP, dwp, grad, cout, c0 + c1, False)
    torch.cuda.synchronize()
    _, dw = O.conv3x3_bwd(eff.astype(np.float64), np.zeros((cout, c0 + c1, 3, 3)), dy.astype(np.float64), need_dx=False)
    close(grad.cpu().numpy(), dw, dt, "wgrad", scale_tol=3e-5 if dt == "f32" else 4e-3)


# Layers with >= 512 tiles interleave the persistent workgroups over the tiles (igemm2.hip / wgrad2.hip).
# The numpy oracle is too slow at these sizes: the reference is torch's own fp32 convolution on the GPU
# applied to the SAME bf16-rounded operands (a different implementation of the same contraction).
INTERLEAVED_SHAPES = [
    (16, 64, 256, 32, 0, 32),   # resident weights, 16-row tiles: 512 tiles
    (4, 64, 512, 64, 0, 256),   # two channel blocks per tile, streamed weights: 512 tiles
    (4, 62, 530, 32, 0, 64),    # ragged last tile row and column: 544 tiles
    (4, 128, 512, 32, 32, 32),  # virtual concat at full-resolution proportions: 512 tiles
]


@pytest.mark.parametrize("shape", INTERLEAVED_SHAPES)
def test_interleaved_tile_walk_matches_torch_convolution(env, shape):
    L, E = env
    dt = "bf16"
    n, h, w, c0, c1, cout = shape
    cin = c0 + c1
    rng = np.random.default_rng(hash(shape) % 2**32 + 7)
    eng = E.UNetEngine(1, 2, 4, dt)
    src, eff = make_src(E, rng, dt, n, h, w, c0, c1, True)
    xe = torch.from_numpy(eff).cuda()                                   # what the convolution sees (NCHW, fp32 values of bf16)
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    wr = torch.from_numpy(rnd(wt, dt)).cuda()
    # fprop + BatchNorm partial sums
    wp = eng._pack("w", fdev(wt), L.PACK_CONV_FPROP, cout, cin)
    y = torch.full((n, h, w, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    part = torch.full((eng._stat_blocks(cout, n, h, w, src), 2, cout), float("nan"), dtype=torch.float32, device="cuda")
    eng._conv(src, wp, cout, 9, n, h, w, y, stats=part)
    ref = torch.nn.functional.conv2d(xe, wr, padding=1).double()
    close(host(y), ref.cpu().numpy(), dt, "interleaved fprop")
    s = part.double().sum(0).cpu().numpy()
    close(s[0], ref.sum(dim=(0, 2, 3)).cpu().numpy(), dt, "sum(y)", scale_tol=2e-3)
    close(s[1], (ref ** 2).sum(dim=(0, 2, 3)).cpu().numpy(), dt, "sum(y^2)", scale_tol=2e-3)
    # dgrad (split into the two sources of a virtual concat)
    dy = rnd(rng.standard_normal((n, cout, h, w)), dt)
    dyt = torch.from_numpy(dy).cuda()
    wpd = eng._pack("wd", fdev(wt), L.PACK_CONV_DGRAD, cout, cin)
    d0 = torch.full((n, h, w, c0), float("nan"), dtype=torch.bfloat16, device="cuda")
    d1 = torch.full((n, h, w, c1), float("nan"), dtype=torch.bfloat16, device="cuda") if c1 else None
    eng._conv(E.Src(dev(dy, dt), cout), wpd, cin, 9, n, h, w, d0, y1=d1, split=c0 if c1 else 0)
    dx = torch.nn.grad.conv2d_input((n, cin, h, w), wr, dyt, padding=1).double().cpu().numpy()
    close(host(d0), dx[:, :c0], dt, "interleaved dgrad part 0")
    if c1:
        close(host(d1), dx[:, c0:], dt, "interleaved dgrad part 1")
    # wgrad
    dwp = eng._wgrad(src, dev(dy, dt), cout, 9, n, h, w)
    grad = torch.full((cout, cin, 3, 3), float("nan"), dtype=torch.float32, device="cuda")
    eng._unpack(L.PACK_CONV_FPROP, dwp, grad, cout, cin, False)
    dw = torch.nn.grad.conv2d_weight(xe, (cout, cin, 3, 3), dyt, padding=1).double().cpu().numpy()
    close(grad.cpu().numpy(), dw, dt, "interleaved wgrad", scale_tol=4e-3)


DECONV_SHAPES = [(2, 4, 8, 16, 8), (1, 8, 16, 64, 32), (1, 2, 2, 64, 32), (1, 4, 4, 8, 4), (1, 8, 32, 128, 64),
                 (2, 16, 32, 64, 32), (1, 8, 64, 256, 128), (1, 8, 32, 512, 256)]  # last three: pipelined bf16 path


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", DECONV_SHAPES)
def test_deconv2x2_fwd_dgrad_wgrad(env, dt, shape):
    L, E = env
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(hash(shape) % 2**32 + 3)
    eng = E.UNetEngine(1, 2, 4, dt)
    src, eff = make_src(E, rng, dt, n, h, w, cin, 0, True)
    wt = (rng.standard_normal((cin, cout, 2, 2)) / np.sqrt(cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    wd = fdev(wt)
    u = torch.full((n, 2 * h, 2 * w, cout), float("nan"), dtype=tdt(dt), device="cuda")
    eng._conv(src, eng._pack("u", wd, L.PACK_DECONV_FPROP, cout, cin), 4 * cout, 1, n, h, w, u,
              out_mode=L.OUT_D2S, bias=fdev(b))
    torch.cuda.synchronize()
    wq = rnd(wt, dt).astype(np.float64)
    close(host(u), O.deconv2x2_fwd(eff.astype(np.float64), wq, b.astype(np.float64)), dt, "deconv fwd")
    du = rnd(rng.standard_normal((n, cout, 2 * h, 2 * w)), dt)
    dud = dev(du, dt)
    da = torch.full((n, h, w, cin), float("nan"), dtype=tdt(dt), device="cuda")
    eng._conv(E.Src(dud, cout), eng._pack("u", wd, L.PACK_DECONV_DGRAD, cout, cin), cin, 1, n, h, w, da,
              in_mode=L.IN_S2D)
    db = torch.zeros((cout,), dtype=torch.float32, device="cuda")
    dwp = eng._wgrad(src, dud, 4 * cout, 1, n, h, w, dy_mode=L.IN_S2D, dbias=db)  # bias grad fused into wgrad
    grad = torch.full((cin, cout, 2, 2), float("nan"), dtype=torch.float32, device="cuda")
    eng._unpack(L.PACK_DECONV_FPROP, dwp, grad, cout, cin, False)
    db2 = torch.full((cout,), float("nan"), dtype=torch.float32, device="cuda")
    L.check(L.lib().oct_channel_sum(eng.dt, dud.data_ptr(), db2.data_ptr(), n * 4 * h * w, cout, 0,
                                    torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    rda, rdw, rdb = O.deconv2x2_bwd(eff.astype(np.float64), wq, du.astype(np.float64))
    close(host(da), rda, dt, "deconv dgrad")
    close(grad.cpu().numpy(), rdw, dt, "deconv wgrad", scale_tol=3e-5 if dt == "f32" else 4e-3)
    close(db.cpu().numpy(), rdb, dt, "deconv bias grad (fused)", scale_tol=1e-4)
    close(db2.cpu().numpy(), rdb, dt, "channel_sum", scale_tol=1e-4)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 16, 32, 32), (1, 8, 8, 4), (2, 4, 6, 24), (1, 32, 64, 64)])
def test_bn_forward_pool_backward(env, dt, shape):
    """finalize (stats, running update) -> relu+pool forward -> dact/pool-route/BN backward"""
    L, E = env
    n, h, w, c = shape
    rng = np.random.default_rng(hash(shape) % 2**32 + 4)
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    edt = L.DT_F32 if dt == "f32" else L.DT_BF16
    y = rnd(rng.standard_normal((n, c, h, w)) * 1.5 + 0.3, dt)
    gamma = (1 + 0.3 * rng.standard_normal(c)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(c)).astype(np.float32)
    yd = dev(y, dt)
    # partial statistics as a conv would have produced them: two fake blocks
    y64 = y.astype(np.float64)
    half = y64[:, :, : h // 2]
    part = np.stack([np.stack([half.sum((0, 2, 3)), (half ** 2).sum((0, 2, 3))]),
                     np.stack([y64[:, :, h // 2:].sum((0, 2, 3)), (y64[:, :, h // 2:] ** 2).sum((0, 2, 3))])])
    rm, rv = fdev(np.zeros(c)), fdev(np.ones(c))
    mean, invstd, scale, shift = (torch.empty(c, device="cuda") for _ in range(4))
    partd, gammad, betad = fdev(part), fdev(gamma), fdev(beta)  # keep alive: raw pointers below
    L.check(lib.oct_bn_finalize(partd.data_ptr(), 2, c, float(n * h * w), gammad.data_ptr(),
                                betad.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), mean.data_ptr(),
                                invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, st))
    z, rmean, rvar, rinv, xhat = O.bn_train_fwd(y64, gamma.astype(np.float64), beta.astype(np.float64))
    close(mean.cpu().numpy(), rmean, "f32", "mean", scale_tol=1e-5)
    close(invstd.cpu().numpy(), rinv, "f32", "invstd", scale_tol=1e-5)
    erm, erv = O.bn_running_update(np.zeros(c), np.ones(c), rmean, rvar, n * h * w)
    close(rm.cpu().numpy(), erm, "f32", "running_mean", scale_tol=1e-5)
    close(rv.cpu().numpy(), erv, "f32", "running_var", scale_tol=1e-5)
    # forward pool
    pooled = torch.full((n, h // 2, w // 2, c), float("nan"), dtype=tdt(dt), device="cuda")
    L.check(lib.oct_bn_relu_pool_fwd(edt, yd.data_ptr(), scale.data_ptr(), shift.data_ptr(), pooled.data_ptr(),
                                     n, h, w, c, st))
    a = np.maximum(z, 0)
    rp, idx = O.maxpool2x2_fwd(a)
    close(host(pooled), rp, dt, "bn+relu+pool")
    # backward with skip gradient + pooled gradient
    da = rnd(rng.standard_normal((n, c, h, w)), dt)
    dp = rnd(rng.standard_normal((n, c, h // 2, w // 2)), dt)
    for use_da, use_dp in ((True, True), (True, False), (False, True)):
        g = dev(da, dt) if use_da else torch.full((n, h, w, c), float("nan"), dtype=tdt(dt), device="cuda")
        dpd = dev(dp, dt) if use_dp else None
        nblk = lib.oct_dact_bn_reduce_blocks(n, h, w, c, 1 if use_dp else 0)
        partials = torch.full((nblk, 2, c), float("nan"), dtype=torch.float32, device="cuda")
        L.check(lib.oct_dact_bn_reduce(edt, g.data_ptr() if use_da else None, L.ptr(dpd), yd.data_ptr(),
                                       scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                       g.data_ptr(), partials.data_ptr(), n, h, w, c, st))
        dgam, dbet = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
        coef = torch.empty((3, c), device="cuda")
        L.check(lib.oct_bn_bwd_finalize(partials.data_ptr(), nblk, c, float(n * h * w), gammad.data_ptr(),
                                        mean.data_ptr(), invstd.data_ptr(), dgam.data_ptr(), dbet.data_ptr(),
                                        coef.data_ptr(), 0, st))
        L.check(lib.oct_bn_bwd_apply(edt, g.data_ptr(), yd.data_ptr(), coef.data_ptr(), None, None, n * h * w, c, st))
        torch.cuda.synchronize()
        dtot = (da.astype(np.float64) if use_da else 0) + (O.maxpool2x2_bwd(dp.astype(np.float64), idx, a.shape) if use_dp else 0)
        dz = dtot * (z > 0)
        rdy, rdg, rdb = O.bn_train_bwd(dz, xhat, gamma.astype(np.float64), rinv)
        tag = f"(da={use_da}, dpool={use_dp})"
        close(dgam.cpu().numpy(), rdg, dt, "dgamma " + tag, scale_tol=1e-4 if dt == "f32" else 6e-3)
        close(dbet.cpu().numpy(), rdb, dt, "dbeta " + tag, scale_tol=1e-4 if dt == "f32" else 6e-3)
        close(host(g), rdy, dt, "dy " + tag, scale_tol=3e-5 if dt == "f32" else 2e-2)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 16, 32, 32, 8), (1, 8, 8, 4, 2), (2, 8, 16, 8, 3), (1, 16, 16, 16, 1), (1, 8, 8, 64, 16)])
def test_head_forward_loss_and_dlogits(env, dt, cfg):
    L, E = env
    n, h, w, f, ncls = cfg
    rng = np.random.default_rng(hash(cfg) % 2**32 + 5)
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    edt = L.DT_F32 if dt == "f32" else L.DT_BF16
    y = rnd(rng.standard_normal((n, f, h, w)), dt)
    sc, sh = rng.uniform(0.5, 1.5, f).astype(np.float32), (0.3 * rng.standard_normal(f)).astype(np.float32)
    wh = (rng.standard_normal((ncls, f)) / np.sqrt(f)).astype(np.float32)
    bh = (0.1 * rng.standard_normal(ncls)).astype(np.float32)
    tgt = rng.integers(0, ncls, (n, h, w))
    yd, scd, shd, whd, bhd = dev(y, dt), fdev(sc), fdev(sh), fdev(wh), fdev(bh)
    tg = torch.from_numpy(tgt).cuda()
    hd = L.HeadDesc(edt, n, h, w, f, ncls)
    nb = lib.oct_head_blocks(C.byref(hd))
    probs = torch.empty((n, ncls, h, w), device="cuda")
    logits = torch.empty((n, ncls, h, w), device="cuda")
    amax = torch.empty((n, h, w), dtype=torch.int64, device="cuda")
    part = torch.empty((nb, L.HEAD_LOSS_SLOTS), dtype=torch.float64, device="cuda")
    L.check(lib.oct_head_forward(C.byref(hd), yd.data_ptr(), scd.data_ptr(), shd.data_ptr(), whd.data_ptr(),
                                 bhd.data_ptr(), tg.data_ptr(), probs.data_ptr(), amax.data_ptr(), logits.data_ptr(),
                                 part.data_ptr(), st))
    w_ce, w_dice, eps = 0.7, 0.6, 1e-7
    loss = torch.empty(3, device="cuda")
    dc = torch.zeros(2 * L.MAX_CLASSES, device="cuda")
    L.check(lib.oct_head_loss_finalize(C.byref(hd), part.data_ptr(), nb, w_ce, w_dice, eps, loss.data_ptr(),
                                       dc.data_ptr(), st))
    dl = torch.full((n, h, w, ncls), float("nan"), dtype=tdt(dt), device="cuda")
    L.check(lib.oct_head_dlogits(C.byref(hd), yd.data_ptr(), scd.data_ptr(), shd.data_ptr(), whd.data_ptr(),
                                 bhd.data_ptr(), tg.data_ptr(), dc.data_ptr(), w_ce, None, dl.data_ptr(), st))
    dpr = rng.standard_normal((n, ncls, h, w)).astype(np.float32)
    dprd = fdev(dpr)
    dl2 = torch.full((n, h, w, ncls), float("nan"), dtype=tdt(dt), device="cuda")
    L.check(lib.oct_head_dlogits(C.byref(hd), yd.data_ptr(), scd.data_ptr(), shd.data_ptr(), whd.data_ptr(),
                                 bhd.data_ptr(), None, None, 0.0, dprd.data_ptr(), dl2.data_ptr(), st))
    torch.cuda.synchronize()
    a = np.maximum(y.astype(np.float64) * sc[None, :, None, None] + sh[None, :, None, None], 0)
    rlog = np.einsum("bchw,oc->bohw", a, wh.astype(np.float64)) + bh[None, :, None, None]
    rl, rce, rdice, cache = O.loss_head_fwd(rlog, tgt, w_ce, w_dice, eps)
    close(logits.cpu().numpy(), rlog, "f32", "logits", scale_tol=2e-5)
    close(probs.cpu().numpy(), cache[0], "f32", "probs", scale_tol=2e-5)
    close(loss.cpu().numpy(), [rl, rce, rdice], "f32", "loss", scale_tol=2e-5)
    pr = cache[0]
    top2 = np.sort(pr, axis=1)[:, -2:] if ncls > 1 else None
    safe = (top2[:, 1] - top2[:, 0]) > 1e-5 if ncls > 1 else np.ones((n, h, w), bool)
    assert np.array_equal(amax.cpu().numpy()[safe], pr.argmax(1)[safe])
    rdl = O.loss_head_bwd(cache, w_ce, w_dice, eps)
    close(host(dl), rdl, dt, "dlogits (fused loss)", scale_tol=3e-5 if dt == "f32" else 1e-2)
    close(host(dl2), O.softmax_bwd(pr, dpr.astype(np.float64)), dt, "dlogits (dprobs)", scale_tol=3e-5 if dt == "f32" else 1e-2)


# (n, h, w, classes, w_dice, want_dlogits, explicit dprobs): which kernel takes it
#   bf16 + classes <= 8 + labels + fused dW + no dlogits -> head_bwd_mfma_kernel (head_mfma.hip, the benchmarked one)
#   everything else                                      -> head_bwd_fused_kernel (head.hip, vector form)
HEAD_BWD_CASES = [(2, 16, 32, 8, 0.0, False, False), (1, 8, 64, 3, 0.6, False, False), (2, 8, 24, 2, 0.0, False, False),
                  (1, 16, 32, 8, 0.5, True, False), (1, 8, 32, 16, 0.0, True, False), (2, 8, 32, 4, 0.0, False, True),
                  (3, 5, 7, 8, 0.0, False, False)]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", HEAD_BWD_CASES)
def test_head_backward_fused_matches_oracle(env, dt, cfg):
    """oct_head_backward_fused -- both forms -- against oracle/ref_cpu.loss_head_* from the tensors the kernel reads
    (teacher-forced: the stored y, the BN coefficients, the labels): dA, dW, db, the cross-entropy rows and the BN-backward
    partial sums of the last block.  Round 2 checked the matrix-pipe form only against the vector form."""
    L, E = env
    n, h, w, ncls, w_dice, want_dl, use_dprobs = cfg
    f = 32
    rng = np.random.default_rng(hash(cfg) % 2**32 + 11)
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    edt = L.DT_F32 if dt == "f32" else L.DT_BF16
    y = rnd(rng.standard_normal((n, f, h, w)), dt)
    mean = y.mean(axis=(0, 2, 3)).astype(np.float32)
    invstd = (1.0 / np.sqrt(y.var(axis=(0, 2, 3)) + 1e-5)).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, f).astype(np.float32), (0.3 * rng.standard_normal(f)).astype(np.float32)
    sc = (gamma * invstd).astype(np.float32)
    sh = (beta - mean * sc).astype(np.float32)
    wh = (rng.standard_normal((ncls, f)) / np.sqrt(f)).astype(np.float32)
    bh = (0.1 * rng.standard_normal(ncls)).astype(np.float32)
    tgt = rng.integers(0, ncls, (n, h, w))
    yd, scd, shd, whd, bhd, md, ivd = dev(y, dt), fdev(sc), fdev(sh), fdev(wh), fdev(bh), fdev(mean), fdev(invstd)
    tg = torch.from_numpy(tgt).cuda()
    hd = L.HeadDesc(edt, n, h, w, f, ncls)
    nb = lib.oct_head_blocks(C.byref(hd))
    w_ce, eps = 0.7, 1e-7
    # oracle forward on what the kernel reads
    z = y.astype(np.float64) * sc[None, :, None, None] + sh[None, :, None, None]
    a = np.maximum(z, 0)
    rlog = np.einsum("bchw,oc->bohw", a, wh.astype(np.float64)) + bh[None, :, None, None]
    rl, rce, rdice, cache = O.loss_head_fwd(rlog, tgt, w_ce, w_dice, eps)
    dc = None
    dprd = None
    if use_dprobs:
        dpr = rng.standard_normal((n, ncls, h, w)).astype(np.float32)
        dprd = fdev(dpr)
        rdl = O.softmax_bwd(cache[0], dpr.astype(np.float64))
    else:
        rdl = O.loss_head_bwd(cache, w_ce, w_dice, eps)
        if w_dice:     # the Dice coefficients come from the forward head pass, as in a training step
            part = torch.empty((nb, L.HEAD_LOSS_SLOTS), dtype=torch.float64, device="cuda")
            probs = torch.empty((n, ncls, h, w), device="cuda")
            L.check(lib.oct_head_forward(C.byref(hd), yd.data_ptr(), scd.data_ptr(), shd.data_ptr(), whd.data_ptr(),
                                         bhd.data_ptr(), tg.data_ptr(), probs.data_ptr(), None, None, part.data_ptr(), st))
            loss = torch.empty(3, device="cuda")
            dc = torch.zeros(2 * L.MAX_CLASSES, device="cuda")
            L.check(lib.oct_head_loss_finalize(C.byref(hd), part.data_ptr(), nb, w_ce, w_dice, eps, loss.data_ptr(), dc.data_ptr(), st))
    fused_dw = ncls <= 8
    dl = torch.full((n, h, w, ncls), float("nan"), dtype=tdt(dt), device="cuda") if (want_dl or not fused_dw) else None
    da = torch.full((n, h, w, f), float("nan"), dtype=tdt(dt), device="cuda")
    partials = torch.full((nb, 2, f), float("nan"), device="cuda")
    db = torch.zeros(ncls, device="cuda")
    dw = torch.zeros((ncls, f), device="cuda") if fused_dw else None
    ce_rows = torch.full((nb, L.HEAD_LOSS_SLOTS), float("nan"), dtype=torch.float64, device="cuda") if not use_dprobs else None
    L.check(lib.oct_head_backward_fused(
        C.byref(hd), yd.data_ptr(), scd.data_ptr(), shd.data_ptr(), md.data_ptr(), ivd.data_ptr(), whd.data_ptr(), bhd.data_ptr(),
        None if use_dprobs else tg.data_ptr(), L.ptr(dc), w_ce, L.ptr(dprd), L.ptr(dl), da.data_ptr(), partials.data_ptr(),
        db.data_ptr(), L.ptr(dw), L.ptr(ce_rows), st), "oct_head_backward_fused")
    torch.cuda.synchronize()
    rda = np.einsum("bchw,cf->bfhw", rdl, wh.astype(np.float64))
    tol = dict(scale_tol=3e-5) if dt == "f32" else {}
    close(host(da), rda, dt, "dA = W^T dlogits (unmasked)", **tol)
    if dl is not None:
        close(host(dl), rdl, dt, "dlogits", scale_tol=3e-5 if dt == "f32" else 1e-2)
    g = rda * (z > 0)
    xhat = (y.astype(np.float64) - mean[None, :, None, None]) * invstd[None, :, None, None]
    ps = partials.double().sum(0).cpu().numpy()
    close(ps[0], g.sum(axis=(0, 2, 3)), dt, "BN-backward partial sums: sum g", scale_tol=1e-4 if dt == "f32" else 6e-3)
    close(ps[1], (g * xhat).sum(axis=(0, 2, 3)), dt, "BN-backward partial sums: sum g*xhat", scale_tol=1e-4 if dt == "f32" else 6e-3)
    close(db.cpu().numpy(), rdl.sum(axis=(0, 2, 3)), dt, "db", scale_tol=1e-4 if dt == "f32" else 6e-3)
    if fused_dw:
        close(dw.cpu().numpy(), np.einsum("bchw,bfhw->cf", rdl, a), dt, "dW", scale_tol=1e-4 if dt == "f32" else 6e-3)
    if ce_rows is not None:
        loss = torch.empty(3, device="cuda")
        dc2 = torch.zeros(2 * L.MAX_CLASSES, device="cuda")
        L.check(lib.oct_head_loss_finalize(C.byref(hd), ce_rows.data_ptr(), nb, 1.0, 0.0, eps, loss.data_ptr(), dc2.data_ptr(), st))
        torch.cuda.synchronize()
        ce = O.loss_head_fwd(rlog, tgt, 1.0, 0.0, eps)[1]
        close(loss.cpu().numpy()[1:2], [ce], dt, "cross-entropy from the backward's loss rows", scale_tol=2e-5 if dt == "f32" else 2e-3)


def test_layout_roundtrip_and_sgd(env):
    L, E = env
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 3, 8, 16)).astype(np.float32)
    xd = fdev(x)
    for dtn, edt in (("f32", L.DT_F32), ("bf16", L.DT_BF16)):
        t = torch.empty((2, 8, 16, 3), dtype=tdt(dtn), device="cuda")
        back = torch.empty((2, 3, 8, 16), device="cuda")
        L.check(lib.oct_nchw_to_nhwc(edt, xd.data_ptr(), t.data_ptr(), 2, 3, 8, 16, st))
        L.check(lib.oct_nhwc_to_nchw(edt, t.data_ptr(), back.data_ptr(), 2, 3, 8, 16, st))
        torch.cuda.synchronize()
        assert np.array_equal(host(t), rnd(x, dtn))
        assert np.array_equal(back.cpu().numpy(), rnd(x, dtn))
    p0, g = rng.standard_normal(1000).astype(np.float32), rng.standard_normal(1000).astype(np.float32)
    p, gd, buf = fdev(p0), fdev(g), torch.zeros(1000, device="cuda")
    L.check(lib.oct_sgd_step(p.data_ptr(), gd.data_ptr(), buf.data_ptr(), 1000, 0.1, 0.9, 0.0, 1.0, 1, st))
    L.check(lib.oct_sgd_step(p.data_ptr(), gd.data_ptr(), buf.data_ptr(), 1000, 0.1, 0.9, 0.0, 1.0, 0, st))
    torch.cuda.synchronize()
    b1 = g.copy(); p1 = p0 - 0.1 * b1; b2 = 0.9 * b1 + g; p2 = p1 - 0.1 * b2
    np.testing.assert_allclose(p.cpu().numpy(), p2, rtol=1e-6, atol=1e-6)


def test_errors_are_reported_not_thrown(env):
    L, E = env
    d = L.ConvDesc(L.DT_BF16, 1, 8, 8, 4, 0, 4, 5, 0, 0, 0, 0, 0, 0)  # taps=5 is invalid
    a = L.ConvArgs()
    rc = L.lib().oct_conv_forward(C.byref(d), C.byref(a), None)
    assert rc == -22 and "taps" in L.last_error()
    with pytest.raises(L.OctError):
        L.check(rc, "oct_conv_forward")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("shape,with_skip", [((2, 8, 16, 32), True), ((1, 6, 10, 64), True), ((3, 4, 8, 16), False), ((2, 16, 32, 256), True)])
def test_pooled_bn_backward_without_stored_masked_gradient_is_bit_identical(env, dt, shape, with_skip):
    """oct_dact_bn_reduce(g = NULL) + oct_bn_bwd_apply_pool against the two-pass form that stores the routed, masked gradient g
    (oct_dact_bn_reduce(g) + oct_bn_bwd_apply(g)): equal partial sums, equal dY, bit for bit."""
    L, E = env
    lib = L.lib()
    n, h, w, c = shape
    eng = E.UNetEngine(1, 2, 4, dt)
    tdt = eng.tdt
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + c)
    y = torch.randn(n, h, w, c, generator=g).to(tdt).cuda()
    da = torch.randn(n, h, w, c, generator=g).to(tdt).cuda() if with_skip else None
    dpool = torch.randn(n, h // 2, w // 2, c, generator=g).to(tdt).cuda()
    scale = (torch.rand(c, generator=g) + 0.5).cuda(); shift = (torch.randn(c, generator=g) * 0.3).cuda()
    mean = (torch.randn(c, generator=g) * 0.1).cuda(); invstd = (torch.rand(c, generator=g) + 0.5).cuda()
    coef = torch.randn(3, c, generator=g).cuda()
    st = torch.cuda.current_stream().cuda_stream
    assert lib.oct_bn_bwd_apply_pool_ok(eng.dt, n, h, w, c) == 1
    nblk = lib.oct_dact_bn_reduce_blocks(n, h, w, c, 1)
    # two-pass reference form
    g1 = da.clone() if with_skip else torch.empty_like(y)
    p1 = torch.full((nblk, 2, c), float("nan"), device="cuda")
    L.check(lib.oct_dact_bn_reduce(eng.dt, L.ptr(g1 if with_skip else None), dpool.data_ptr(), y.data_ptr(), scale.data_ptr(),
                                   shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), g1.data_ptr(), p1.data_ptr(), n, h, w, c, st))
    L.check(lib.oct_bn_bwd_apply(eng.dt, g1.data_ptr(), y.data_ptr(), coef.data_ptr(), None, None, n * h * w, c, st))
    # reduce only + fused apply
    g2 = da.clone() if with_skip else torch.empty_like(y)
    p2 = torch.full((nblk, 2, c), float("nan"), device="cuda")
    L.check(lib.oct_dact_bn_reduce(eng.dt, L.ptr(g2 if with_skip else None), dpool.data_ptr(), y.data_ptr(), scale.data_ptr(),
                                   shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), None, p2.data_ptr(), n, h, w, c, st))
    if with_skip:
        assert torch.equal(g2, da)     # the reduce-only pass wrote nothing
    L.check(lib.oct_bn_bwd_apply_pool(eng.dt, L.ptr(g2 if with_skip else None), dpool.data_ptr(), y.data_ptr(), scale.data_ptr(),
                                      shift.data_ptr(), coef.data_ptr(), g2.data_ptr(), n, h, w, c, st))
    torch.cuda.synchronize()
    assert torch.equal(p1, p2), "partial sums"
    assert torch.equal(g1, g2), "dY"
