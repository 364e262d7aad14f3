"""Exact-arithmetic parity of the PRODUCTION bf16 kernels (igemm2.hip, wgrad2.hip, first.hip -- the kernels
bench.py times) against oracle/ref_cpu.py: BIT equality, not a tolerance.

Operands are chosen so that no rounding can occur anywhere in the kernel: activations are small integers,
weights are signed powers of two (or zero), BatchNorm scale/shift applied on load are powers of two / halves.
Every product is then exact, every partial sum is a multiple of 2^-3 below 2^21 (exact in fp32 in ANY
summation order, including the fp32 atomics of the weight gradients), and the only rounding left is the final
fp32 -> bf16 store, which is deterministic (round to nearest even) and reproduced on the oracle's float64
result.  A wrong tap, channel block, halo pixel, concat offset or tile predicate therefore shows up as a
non-zero difference, however small its numerical weight would be under a tolerance.
"""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from retinal_oct_image_segmentation_via_deep_learning_amd import _lib as L
    from retinal_oct_image_segmentation_via_deep_learning_amd import engine as E
    L.lib()
    return L, E


def to_bf16(a):
    return torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).float().numpy()


def dev(a_nchw):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a_nchw, np.float32).transpose(0, 2, 3, 1))).to("cuda", torch.bfloat16)


def host(t_nhwc):
    return t_nhwc.float().cpu().numpy().transpose(0, 3, 1, 2)


def fdev(a):
    return torch.from_numpy(np.asarray(a, np.float32)).cuda().contiguous()


def ints(rng, shape, lo=-4, hi=4):
    return rng.integers(lo, hi + 1, shape).astype(np.float32)


def pow2_weights(rng, shape, density=0.6):
    """0 or +-2^k, k in {-3..0}"""
    mag = np.exp2(rng.integers(-3, 1, shape)).astype(np.float32)
    sign = rng.choice([-1.0, 1.0], shape).astype(np.float32)
    keep = (rng.random(shape) < density).astype(np.float32)
    return mag * sign * keep


def exact_src(E, rng, n, h, w, c0, c1, xform):
    """integer sources; on-load transform max(x*s + b, 0) with s in {+-0.5, +-1, 2}, b in halves: exact in fp32 and
    exactly representable in bf16 (multiples of 0.5 below 16)"""
    parts = []
    x0 = ints(rng, (n, c0, h, w))
    bn0 = bn1 = None
    if xform:
        s0 = rng.choice([-1.0, -0.5, 0.5, 1.0, 2.0], c0).astype(np.float32)
        b0 = (rng.integers(-2, 3, c0) * 0.5).astype(np.float32)
        bn0 = E.BNState(fdev(s0), fdev(b0))
        parts.append(np.maximum(x0 * s0[None, :, None, None] + b0[None, :, None, None], 0))
    else:
        parts.append(x0)
    x1d = None
    if c1:
        x1 = ints(rng, (n, c1, h, w))
        s1 = rng.choice([-1.0, 0.5, 1.0, 2.0], c1).astype(np.float32)
        b1 = (rng.integers(-2, 3, c1) * 0.5).astype(np.float32)
        bn1 = E.BNState(fdev(s1), fdev(b1))
        parts.append(np.maximum(x1 * s1[None, :, None, None] + b1[None, :, None, None], 0))
        x1d = dev(x1)
    eff = np.concatenate(parts, axis=1)
    assert np.array_equal(to_bf16(eff), eff)
    return E.Src(dev(x0), c0, bn0, x1d, c1, bn1), eff.astype(np.float64)


def same(got, ref, what):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    bad = got != ref
    assert not bad.any(), (f"{what}: {int(bad.sum())} of {bad.size} elements differ, first at "
                           f"{np.unravel_index(bad.argmax(), bad.shape)}: got {got[bad][0]!r} want {ref[bad][0]!r}")


# every regular shape of tests/test_gpu_kernels.py CONV_SHAPES that the pipelined bf16 kernels take (channel counts
# multiples of 32; first layer: 1 -> 16/32/64), plus the interleaved-walk and ragged cases
PIPELINED = [
    (2, 16, 64, 32, 0, 32), (1, 24, 32, 32, 0, 64), (2, 8, 64, 64, 0, 32), (1, 16, 32, 32, 32, 32),
    (1, 8, 32, 64, 64, 64), (1, 16, 32, 64, 0, 128), (1, 8, 32, 128, 128, 256), (3, 8, 32, 96, 0, 192),
    (2, 32, 64, 64, 64, 128), (1, 16, 64, 128, 0, 256), (3, 16, 32, 32, 0, 128), (1, 48, 32, 96, 32, 128),
    (2, 31, 48, 32, 0, 64), (1, 62, 96, 32, 32, 32), (1, 12, 40, 64, 0, 128),
    (1, 32, 64, 256, 256, 256), (1, 16, 32, 512, 0, 512),          # the headline net's deepest shapes (K = 4608)
]
FIRST = [(2, 16, 64, 1, 0, 32), (1, 24, 40, 1, 0, 16), (1, 32, 32, 1, 0, 64)]
INTERLEAVED = [(16, 64, 256, 32, 0, 32), (4, 64, 512, 64, 0, 256), (4, 62, 530, 32, 0, 64), (4, 128, 512, 32, 32, 32)]


def _torch_ref_fprop(eff, wt):
    """large cases (the numpy im2col would need GBs): torch's float64 convolution on the host -- a different
    implementation of the same contraction, exact for these operands"""
    return torch.nn.functional.conv2d(torch.from_numpy(eff).double(), torch.from_numpy(wt).double(), padding=1).numpy()


@pytest.mark.parametrize("shape", PIPELINED + FIRST + INTERLEAVED)
@pytest.mark.parametrize("xform", [False, True])
def test_fprop_bit_exact(env, shape, xform):
    L, E = env
    n, h, w, c0, c1, cout = shape
    if c0 == 1 and xform:
        pytest.skip("the first layer reads the raw input")
    rng = np.random.default_rng(abs(hash((shape, xform))) % 2**32)
    eng = E.UNetEngine(1, 2, 4, "bf16")
    src, eff = exact_src(E, rng, n, h, w, c0, c1, xform)
    wt = pow2_weights(rng, (cout, c0 + c1, 3, 3), density=0.6 if c0 + c1 <= 128 else 0.25)
    wp = eng._pack("w", fdev(wt), L.PACK_CONV_FPROP, cout, c0 + c1)
    y = torch.full((n, h, w, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    part = torch.full((eng._stat_blocks(cout, n, h, w, src), 2, cout), float("nan"), dtype=torch.float32, device="cuda")
    eng._conv(src, wp, cout, 9, n, h, w, y, stats=part)
    torch.cuda.synchronize()
    big = n * h * w * cout * (c0 + c1) > 5e8
    ref = _torch_ref_fprop(eff, wt) if big else O.conv3x3_fwd(eff, wt.astype(np.float64))
    assert np.abs(ref).max() * 8 < 2 ** 22                      # the premise: fp32 holds every partial sum exactly
    same(host(y), to_bf16(ref), "fprop (bf16 store of the exact sum)")
    # BatchNorm partial sums are taken from the fp32 accumulators BEFORE the bf16 store
    s = part.double().sum(0).cpu().numpy()
    s1, s2 = ref.sum(axis=(0, 2, 3)), (ref ** 2).sum(axis=(0, 2, 3))
    if np.abs(ref).sum(axis=(0, 2, 3)).max() * 8 < 2 ** 24:      # then every per-lane / per-wave partial sum is exact too
        same(s[0], s1, "sum(y) from the fp32 accumulators")
    else:
        np.testing.assert_allclose(s[0], s1, rtol=1e-5, atol=1e-2)
    np.testing.assert_allclose(s[1], s2, rtol=2e-5)            # squares are summed in fp32: not exact by construction


@pytest.mark.parametrize("shape", PIPELINED + INTERLEAVED)
def test_dgrad_bit_exact(env, shape):
    L, E = env
    n, h, w, c0, c1, cout = shape
    cin = c0 + c1
    rng = np.random.default_rng(abs(hash(shape)) % 2**32 + 1)
    eng = E.UNetEngine(1, 2, 4, "bf16")
    wt = pow2_weights(rng, (cout, cin, 3, 3), density=0.6 if cout <= 128 else 0.25)
    dy = ints(rng, (n, cout, h, w))
    wp = eng._pack("w", fdev(wt), L.PACK_CONV_DGRAD, cout, cin)
    d0 = torch.full((n, h, w, c0), float("nan"), dtype=torch.bfloat16, device="cuda")
    d1 = torch.full((n, h, w, c1), float("nan"), dtype=torch.bfloat16, device="cuda") if c1 else None
    eng._conv(E.Src(dev(dy), cout), wp, cin, 9, n, h, w, d0, y1=d1, split=c0 if c1 else 0)
    torch.cuda.synchronize()
    if n * h * w * cout * cin > 5e8:
        dx = torch.nn.grad.conv2d_input((n, cin, h, w), torch.from_numpy(wt).double(), torch.from_numpy(dy).double(),
                                        padding=1).numpy()
    else:
        dx, _ = O.conv3x3_bwd(np.zeros((n, cin, h, w)), wt.astype(np.float64), dy.astype(np.float64))
    same(host(d0), to_bf16(dx[:, :c0]), "dgrad part 0")
    if c1:
        same(host(d1), to_bf16(dx[:, c0:]), "dgrad part 1 (virtual concat split)")


@pytest.mark.parametrize("shape", PIPELINED + FIRST + INTERLEAVED)
def test_wgrad_bit_exact(env, shape):
    """fp32 atomics across workgroups add dyadic rationals far below 2^24: any order gives the same bits"""
    L, E = env
    n, h, w, c0, c1, cout = shape
    cin = c0 + c1
    rng = np.random.default_rng(abs(hash(shape)) % 2**32 + 2)
    eng = E.UNetEngine(1, 2, 4, "bf16")
    src, eff = exact_src(E, rng, n, h, w, c0, c1, xform=(c0 != 1))
    dy = ints(rng, (n, cout, h, w), -2, 2) * (rng.random((n, cout, h, w)) < 0.5)
    dwp = eng._wgrad(src, dev(dy), cout, 9, n, h, w)
    grad = torch.full((cout, cin, 3, 3), float("nan"), dtype=torch.float32, device="cuda")
    eng._unpack(L.PACK_CONV_FPROP, dwp, grad, cout, cin, False)
    torch.cuda.synchronize()
    if n * h * w * cout * cin > 5e8:
        dw = torch.nn.grad.conv2d_weight(torch.from_numpy(eff).double(), (cout, cin, 3, 3),
                                         torch.from_numpy(dy.astype(np.float64)), padding=1).numpy()
    else:
        _, dw = O.conv3x3_bwd(eff, np.zeros((cout, cin, 3, 3)), dy.astype(np.float64), need_dx=False)
    assert np.abs(dw).max() * 2 < 2 ** 23
    same(grad.cpu().numpy(), dw, "wgrad")


DECONV = [(2, 16, 32, 64, 32), (1, 8, 64, 256, 128), (1, 8, 32, 512, 256), (1, 16, 64, 128, 64),
          # gemm1.hip (N % 256 == 0): several items per workgroup, an odd chunk count, both channel-block counts
          (8, 32, 128, 64, 256), (4, 16, 64, 96, 64), (2, 16, 64, 256, 64), (3, 24, 32, 128, 192)]


@pytest.mark.parametrize("shape", DECONV)
def test_deconv_fwd_dgrad_wgrad_bit_exact(env, shape):
    L, E = env
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(abs(hash(shape)) % 2**32 + 3)
    eng = E.UNetEngine(1, 2, 4, "bf16")
    src, eff = exact_src(E, rng, n, h, w, cin, 0, True)
    wt = pow2_weights(rng, (cin, cout, 2, 2), density=0.5)
    b = (rng.integers(-4, 5, cout) * 0.5).astype(np.float32)
    wd = fdev(wt)
    u = torch.full((n, 2 * h, 2 * w, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    eng._conv(src, eng._pack("u", wd, L.PACK_DECONV_FPROP, cout, cin), 4 * cout, 1, n, h, w, u, out_mode=L.OUT_D2S,
              bias=fdev(b))
    du = ints(rng, (n, cout, 2 * h, 2 * w), -2, 2)
    dud = dev(du)
    da = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device="cuda")
    eng._conv(E.Src(dud, cout), eng._pack("u", wd, L.PACK_DECONV_DGRAD, cout, cin), cin, 1, n, h, w, da, in_mode=L.IN_S2D)
    db = torch.zeros((cout,), dtype=torch.float32, device="cuda")
    dwp = eng._wgrad(src, dud, 4 * cout, 1, n, h, w, dy_mode=L.IN_S2D, dbias=db)
    grad = torch.full((cin, cout, 2, 2), float("nan"), dtype=torch.float32, device="cuda")
    eng._unpack(L.PACK_DECONV_FPROP, dwp, grad, cout, cin, False)
    torch.cuda.synchronize()
    wq = wt.astype(np.float64)
    same(host(u), to_bf16(O.deconv2x2_fwd(eff, wq, b.astype(np.float64))), "deconv fwd (+bias, depth-to-space store)")
    rda, rdw, rdb = O.deconv2x2_bwd(eff, wq, du.astype(np.float64))
    same(host(da), to_bf16(rda), "deconv dgrad (space-to-depth gather)")
    same(grad.cpu().numpy(), rdw, "deconv wgrad")
    same(db.cpu().numpy(), rdb, "deconv bias gradient (ones-fragment MFMA)")


@pytest.mark.parametrize("f", [16, 32, 64])
def test_first_layer_wgrad_with_fused_bn_backward_bit_exact(env, f):
    """first.hip applies dy = k0*[y*s+b > 0]*dA + k1*y + k2 on load (dY of layer 1 is never stored): with dyadic
    coefficients the bf16-rounded dy and the 9-tap accumulation are exact"""
    L, E = env
    n, h, w = 2, 16, 64
    rng = np.random.default_rng(f)
    eng = E.UNetEngine(1, 2, 4, "bf16")
    x = ints(rng, (n, 1, h, w), -3, 3)
    y = ints(rng, (n, f, h, w), -4, 4)
    dA = ints(rng, (n, f, h, w), -2, 2)
    sc = rng.choice([-1.0, 0.5, 1.0, 2.0], f).astype(np.float32)
    sh = (rng.integers(-2, 3, f) * 0.5 + 0.25).astype(np.float32)          # never exactly zero: no tie at the mask
    coef = np.stack([rng.choice([0.5, 1.0, 2.0], f), rng.choice([-0.25, 0.0, 0.25], f), rng.choice([-0.5, 0.0, 0.5], f)]).astype(np.float32)
    src = E.Src(dev(x), 1)
    dwp = eng._wgrad(src, dev(dA), f, 9, n, h, w, fused_apply=(dev(y), fdev(coef), fdev(sc), fdev(sh)))
    grad = torch.full((f, 1, 3, 3), float("nan"), dtype=torch.float32, device="cuda")
    eng._unpack(L.PACK_CONV_FPROP, dwp, grad, f, 1, False)
    torch.cuda.synchronize()
    mask = (y * sc[None, :, None, None] + sh[None, :, None, None]) > 0
    dy = coef[0][None, :, None, None] * (dA * mask) + coef[1][None, :, None, None] * y + coef[2][None, :, None, None]
    assert np.array_equal(to_bf16(dy), dy)
    _, dw = O.conv3x3_bwd(x.astype(np.float64), np.zeros((f, 1, 3, 3)), dy.astype(np.float64), need_dx=False)
    same(grad.cpu().numpy(), dw, "first-layer wgrad with fused BN-backward apply")


ONE_BY_ONE = [(2, 16, 64, 64, 0, 32), (1, 31, 48, 128, 0, 64), (1, 62, 96, 256, 0, 128), (2, 8, 32, 32, 32, 64), (1, 24, 40, 512, 0, 256)]


@pytest.mark.parametrize("shape", ONE_BY_ONE)
def test_conv1x1_plain_fprop_dgrad_wgrad_bit_exact(env, shape):
    """plain 1x1 convolutions (attention gates W_g / W_x, classifier heads) on the pipelined kernels: whole and ragged
    tiles, BatchNorm partial sums, virtual concat, data gradient, weight gradient with the bias gradient fused in"""
    L, E = env
    n, h, w, c0, c1, cout = shape
    cin = c0 + c1
    rng = np.random.default_rng(abs(hash(shape)) % 2**32 + 11)
    eng = E.UNetEngine(1, 2, 4, "bf16")
    src, eff = exact_src(E, rng, n, h, w, c0, c1, True)
    wt = pow2_weights(rng, (cout, cin, 1, 1), density=0.5)
    wd = fdev(wt)
    y = torch.full((n, h, w, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    part = torch.full((eng._stat_blocks(cout, n, h, w, src, taps=1), 2, cout), float("nan"), dtype=torch.float32, device="cuda")
    eng._conv(src, eng._pack("w1", wd, L.PACK_1X1_FPROP, cout, cin), cout, 1, n, h, w, y, stats=part)
    ref = np.einsum("bchw,oc->bohw", eff, wt[:, :, 0, 0].astype(np.float64))
    same(host(y), to_bf16(ref), "1x1 fprop")
    if np.abs(ref).sum(axis=(0, 2, 3)).max() * 8 < 2 ** 24:
        same(part.double().sum(0).cpu().numpy()[0], ref.sum(axis=(0, 2, 3)), "1x1 sum(y)")
    dy = ints(rng, (n, cout, h, w), -2, 2)
    d0 = torch.full((n, h, w, c0), float("nan"), dtype=torch.bfloat16, device="cuda")
    d1 = torch.full((n, h, w, c1), float("nan"), dtype=torch.bfloat16, device="cuda") if c1 else None
    eng._conv(E.Src(dev(dy), cout), eng._pack("w1", wd, L.PACK_1X1_DGRAD, cout, cin), cin, 1, n, h, w, d0, y1=d1, split=c0 if c1 else 0)
    dx = np.einsum("bohw,oc->bchw", dy.astype(np.float64), wt[:, :, 0, 0].astype(np.float64))
    same(host(d0), to_bf16(dx[:, :c0]), "1x1 dgrad part 0")
    if c1:
        same(host(d1), to_bf16(dx[:, c0:]), "1x1 dgrad part 1")
    db = torch.zeros(cout, dtype=torch.float32, device="cuda")
    dwp = eng._wgrad(src, dev(dy), cout, 1, n, h, w, dbias=db)
    grad = torch.full((cout, cin, 1, 1), float("nan"), dtype=torch.float32, device="cuda")
    eng._unpack(L.PACK_1X1_FPROP, dwp, grad, cout, cin, False)
    torch.cuda.synchronize()
    same(grad.cpu().numpy()[:, :, 0, 0], np.einsum("bohw,bchw->oc", dy.astype(np.float64), eff), "1x1 wgrad")
    same(db.cpu().numpy(), dy.astype(np.float64).sum(axis=(0, 2, 3)), "1x1 bias gradient (ones-fragment MFMA)")


@pytest.mark.parametrize("shape", [(2, 16, 64, 32, 0, 64), (1, 62, 96, 64, 64, 32)])
def test_conv3x3_wgrad_with_fused_bias_gradient_bit_exact(env, shape):
    """3x3 convolution with bias and no BatchNorm (conv_block.init_conv, common.py:9): db rides in the weight-gradient kernel"""
    L, E = env
    n, h, w, c0, c1, cout = shape
    rng = np.random.default_rng(abs(hash(shape)) % 2**32 + 12)
    eng = E.UNetEngine(1, 2, 4, "bf16")
    src, eff = exact_src(E, rng, n, h, w, c0, c1, False)
    dy = ints(rng, (n, cout, h, w), -2, 2)
    db = torch.zeros(cout, dtype=torch.float32, device="cuda")
    dwp = eng._wgrad(src, dev(dy), cout, 9, n, h, w, dbias=db)
    grad = torch.full((cout, c0 + c1, 3, 3), float("nan"), dtype=torch.float32, device="cuda")
    eng._unpack(L.PACK_CONV_FPROP, dwp, grad, cout, c0 + c1, False)
    torch.cuda.synchronize()
    _, dw = O.conv3x3_bwd(eff, np.zeros((cout, c0 + c1, 3, 3)), dy.astype(np.float64), need_dx=False)
    same(grad.cpu().numpy(), dw, "wgrad")
    same(db.cpu().numpy(), dy.astype(np.float64).sum(axis=(0, 2, 3)), "bias gradient")


# the 64 / 128-channel-block shapes run on the pipelined kernels (igemm2 TAPS = 21: three halo rows; wgrad2: three row-shifted
# launches), the others on the generic ones: several tile rows (interior tiles), ragged heights with 4 and 6 rows left,
# ragged widths, concat sources, whole tiles (LDS-DMA data gradient), and a height the pipelined kernels refuse (9 = 8 + 1)
KK73 = [(2, 16, 32, 32, 0, 32), (1, 14, 40, 64, 0, 64), (1, 16, 32, 64, 64, 64), (2, 9, 24, 32, 0, 96),
        (1, 40, 64, 64, 0, 64), (1, 30, 64, 64, 64, 64), (2, 24, 64, 128, 0, 128), (1, 20, 40, 64, 0, 64), (1, 17, 32, 64, 0, 64),
        (1, 5, 32, 96, 32, 192), (1, 3, 64, 64, 0, 64), (2, 11, 32, 64, 0, 128)]   # three channel blocks / chunks, images lower than the kernel, 3 rows left


@pytest.mark.parametrize("shape", KK73)
def test_conv7x3_fprop_dgrad_wgrad_bit_exact(env, shape):
    """ReLayNet's 7x3 convolution (ReLayNet_2017.py:155-160; padding (3, 1)) in bf16 on the (kh, kw) kernels: forward with the
    on-load transform and the BatchNorm sums, data gradient with the concat split, weight gradient in its three row groups --
    bit equality against torch's float64 convolution on the same exactly representable operands."""
    from retinal_oct_image_segmentation_via_deep_learning_amd import ops
    L, E = env
    n, h, w, c0, c1, cout = shape
    cin = c0 + c1
    rng = np.random.default_rng(abs(hash(shape)) % 2**32 + 73)
    eng = ops.kernels("bf16")
    src, eff = exact_src(E, rng, n, h, w, c0, c1, True)
    wt = pow2_weights(rng, (cout, cin, 7, 3), density=0.35)
    wd = fdev(wt)
    kk = dict(kh=7, kw=3)
    # forward
    wp = ops.packed(eng, wd, L.PACK_CONV_FPROP, cout, cin, cache=False, kk=(7, 3))
    y = torch.full((n, h, w, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    part = torch.full((eng._stat_blocks(cout, n, h, w, src, 21, **kk), 2, cout), float("nan"), dtype=torch.float32, device="cuda")
    eng._conv(src, wp, cout, 21, n, h, w, y, stats=part, **kk)
    ref = torch.nn.functional.conv2d(torch.from_numpy(eff).double(), torch.from_numpy(wt).double(), padding=(3, 1)).numpy()
    assert np.abs(ref).max() * 8 < 2 ** 22
    same(host(y), to_bf16(ref), "7x3 fprop")
    np.testing.assert_allclose(part.double().sum(0).cpu().numpy()[0], ref.sum(axis=(0, 2, 3)), rtol=1e-5, atol=1e-2)
    # data gradient (split over the two sources of a virtual concat)
    dy = ints(rng, (n, cout, h, w), -2, 2) * (rng.random((n, cout, h, w)) < 0.5)
    wpd = ops.packed(eng, wd, L.PACK_CONV_DGRAD, cout, cin, cache=False, kk=(7, 3))
    d0 = torch.full((n, h, w, c0), float("nan"), dtype=torch.bfloat16, device="cuda")
    d1 = torch.full((n, h, w, c1), float("nan"), dtype=torch.bfloat16, device="cuda") if c1 else None
    eng._conv(E.Src(dev(dy), cout), wpd, cin, 21, n, h, w, d0, y1=d1, split=c0 if c1 else 0, **kk)
    dx = torch.nn.grad.conv2d_input((n, cin, h, w), torch.from_numpy(wt).double(), torch.from_numpy(dy).double(), padding=(3, 1)).numpy()
    same(host(d0), to_bf16(dx[:, :c0]), "7x3 dgrad part 0")
    if c1:
        same(host(d1), to_bf16(dx[:, c0:]), "7x3 dgrad part 1")
    # weight gradient
    dwp = eng._wgrad(src, dev(dy), cout, 21, n, h, w, partials_ok=False, **kk)
    grad = torch.full((cout, cin, 7, 3), float("nan"), dtype=torch.float32, device="cuda")
    L.check(L.lib().oct_unpack_wgrad_kk(dwp.data_ptr(), grad.data_ptr(), cout, cin, 7, 3, 0, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    dw = torch.nn.grad.conv2d_weight(torch.from_numpy(eff).double(), (cout, cin, 7, 3), torch.from_numpy(dy.astype(np.float64)),
                                     padding=(3, 1)).numpy()
    assert np.abs(dw).max() * 2 < 2 ** 23
    same(grad.cpu().numpy(), dw, "7x3 wgrad")


@pytest.mark.parametrize("shape", [(2, 16, 64, 64), (1, 21, 32, 32), (3, 5, 96, 64), (1, 12, 40, 64)])
def test_conv7x3_first_layer_bit_exact(env, shape):
    """ReLayNet's first convolution, Conv2d(1 -> F, 7x3) (ReLayNet_2017.py:155-160 with in_channels = 1): forward (with the
    BatchNorm sums) and weight gradient on the matrix-pipe first-layer kernels (first.hip, KD = 7: 21 taps in two k16 steps) for
    W % 32 == 0, on the generic kernels otherwise (last shape) -- bit equality against torch's float64 convolution.  Heights
    below, at and above the seven tap rows."""
    from retinal_oct_image_segmentation_via_deep_learning_amd import ops
    L, E = env
    n, h, w, cout = shape
    rng = np.random.default_rng(abs(hash(shape)) % 2**32 + 731)
    eng = ops.kernels("bf16")
    src, eff = exact_src(E, rng, n, h, w, 1, 0, False)
    wt = pow2_weights(rng, (cout, 1, 7, 3), density=0.7)
    wd = fdev(wt)
    kk = dict(kh=7, kw=3)
    wp = ops.packed(eng, wd, L.PACK_CONV_FPROP, cout, 1, cache=False, kk=(7, 3))
    y = torch.full((n, h, w, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    part = torch.full((eng._stat_blocks(cout, n, h, w, src, 21, **kk), 2, cout), float("nan"), dtype=torch.float32, device="cuda")
    eng._conv(src, wp, cout, 21, n, h, w, y, stats=part, **kk)
    ref = torch.nn.functional.conv2d(torch.from_numpy(eff).double(), torch.from_numpy(wt).double(), padding=(3, 1)).numpy()
    same(host(y), to_bf16(ref), "7x3 first-layer fprop")
    np.testing.assert_allclose(part.double().sum(0).cpu().numpy()[0], ref.sum(axis=(0, 2, 3)), rtol=1e-5, atol=1e-2)
    np.testing.assert_allclose(part.double().sum(0).cpu().numpy()[1], (ref ** 2).sum(axis=(0, 2, 3)), rtol=1e-5, atol=1e-2)
    dy = ints(rng, (n, cout, h, w), -2, 2) * (rng.random((n, cout, h, w)) < 0.5)
    dwp = eng._wgrad(src, dev(dy), cout, 21, n, h, w, partials_ok=False, **kk)
    grad = torch.full((cout, 1, 7, 3), float("nan"), dtype=torch.float32, device="cuda")
    L.check(L.lib().oct_unpack_wgrad_kk(dwp.data_ptr(), grad.data_ptr(), cout, 1, 7, 3, 0, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    dw = torch.nn.grad.conv2d_weight(torch.from_numpy(eff).double(), (cout, 1, 7, 3), torch.from_numpy(dy.astype(np.float64)),
                                     padding=(3, 1)).numpy()
    assert np.abs(dw).max() * 2 < 2 ** 23
    same(grad.cpu().numpy(), dw, "7x3 first-layer wgrad")
