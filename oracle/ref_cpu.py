"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the U-Net training hot path.

This file is the checker, never the product: only tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg may import it.  The shipped path
(retinal_oct_image_segmentation_via_deep_learning_amd/) never imports anything from oracle/.

It restates, in plain numpy, the algorithm of the reference's U-Net path:

  * UNet.__init__/_block/forward  -> /root/reference/SOTAS/Lesions_Segment/YNet_2022.py:511-602
    (= SOTAS/Layers_Segment/YNet_2022:48-139, byte-identical class)
  * the arithmetic of every op the reference delegates to torch 2.10 (third-party, not vendored;
    the reference pins no version): Conv2d(3x3,p=1,bias=False), BatchNorm2d(train: batch stats,
    biased var to normalise, unbiased var for running_var, eps 1e-5, momentum 0.1), ReLU,
    MaxPool2d(2,2) (first max wins), ConvTranspose2d(k=2,s=2,bias), cat((dec, enc), 1),
    Conv2d(1x1,bias), Softmax2d  -- call sites YNet_2022.py:516-546, 578-599
  * the loss head the reference does not have: CE = nll_loss(log p) and
    soft Dice 1 - mean_c (2 I_c + eps)/(P_c + Y_c + eps); explicit hand-derived backward
  * torch.optim.SGD with momentum (first step: buf = grad)
  * Metrics/Region_based_metrics.py:3-61 and Metrics/ConfusionMatrix_based_metrics.py:4-63

Pinning: tests/test_oracle.py checks every function here against tests/golden/*.npz, which
tools/gen_golden.py produced by importing the reference itself in the build container
(forward probabilities / argmax, loss, every parameter gradient, BN buffers, a 3-step SGD
trajectory, metric known answers).  The reference holds no tests or golden vectors of its own.

Layout here is the reference's: NCHW, float64 by default (so that this oracle is at least as
accurate as the fp32 reference; agreement with the fp32 fixtures is ~1e-6).
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------------
# primitive ops (forward + hand-written backward)
# --------------------------------------------------------------------------------------------
def _im2col3x3(x):
    """x: (B,C,H,W) -> (B,H,W,C*9) with zero padding 1; column order (c, kh, kw)."""
    B, C, H, W = x.shape
    xp = np.zeros((B, C, H + 2, W + 2), dtype=x.dtype)
    xp[:, :, 1:-1, 1:-1] = x
    win = np.lib.stride_tricks.sliding_window_view(xp, (3, 3), axis=(2, 3))  # B,C,H,W,3,3
    return np.ascontiguousarray(win.transpose(0, 2, 3, 1, 4, 5)).reshape(B, H, W, C * 9)


def conv3x3_fwd(x, w, bias=None):
    """Conv2d k=3, stride 1, padding 1.  w: (Cout,Cin,3,3)."""
    B, C, H, W = x.shape
    col = _im2col3x3(x)
    y = col.reshape(-1, C * 9) @ w.reshape(w.shape[0], -1).T
    y = y.reshape(B, H, W, -1).transpose(0, 3, 1, 2)
    if bias is not None:
        y = y + bias[None, :, None, None]
    return np.ascontiguousarray(y)


def conv3x3_bwd(x, w, dy, need_dx=True):
    """Returns (dx, dw).  dgrad = conv of dy with the flipped, transposed filter."""
    B, C, H, W = x.shape
    Co = w.shape[0]
    col = _im2col3x3(x).reshape(-1, C * 9)
    dyf = dy.transpose(0, 2, 3, 1).reshape(-1, Co)
    dw = (dyf.T @ col).reshape(Co, C, 3, 3)
    dx = None
    if need_dx:
        wt = np.ascontiguousarray(w[:, :, ::-1, ::-1].transpose(1, 0, 2, 3))  # (Cin,Cout,3,3)
        dx = conv3x3_fwd(dy, wt)
    return dx, dw


def bn_train_fwd(y, gamma, beta, eps=BN_EPS):
    """Training-mode BatchNorm2d: batch mean, biased variance."""
    mean = y.mean(axis=(0, 2, 3))
    var = y.var(axis=(0, 2, 3))  # biased
    invstd = 1.0 / np.sqrt(var + eps)
    xhat = (y - mean[None, :, None, None]) * invstd[None, :, None, None]
    z = xhat * gamma[None, :, None, None] + beta[None, :, None, None]
    return z, mean, var, invstd, xhat


def bn_running_update(running_mean, running_var, mean, var_biased, n, momentum=BN_MOMENTUM):
    unbiased = var_biased * (n / max(n - 1, 1))
    rm = (1 - momentum) * running_mean + momentum * mean
    rv = (1 - momentum) * running_var + momentum * unbiased
    return rm, rv


def bn_eval_fwd(y, gamma, beta, running_mean, running_var, eps=BN_EPS):
    scale = gamma / np.sqrt(running_var + eps)
    shift = beta - running_mean * scale
    return y * scale[None, :, None, None] + shift[None, :, None, None]


def bn_train_bwd(dz, xhat, gamma, invstd):
    """Returns (dy, dgamma, dbeta)."""
    n = dz.shape[0] * dz.shape[2] * dz.shape[3]
    dbeta = dz.sum(axis=(0, 2, 3))
    dgamma = (dz * xhat).sum(axis=(0, 2, 3))
    a = (gamma * invstd)[None, :, None, None]
    dy = a * (dz - dbeta[None, :, None, None] / n - xhat * dgamma[None, :, None, None] / n)
    return dy, dgamma, dbeta


def maxpool2x2_fwd(a):
    """MaxPool2d(2,2).  Returns pooled and the flat argmax index (0..3, first max wins)."""
    B, C, H, W = a.shape
    v = a.reshape(B, C, H // 2, 2, W // 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(B, C, H // 2, W // 2, 4)
    idx = v.argmax(axis=-1)  # numpy argmax returns the first maximum, like ATen's max_pool2d
    return np.take_along_axis(v, idx[..., None], -1)[..., 0], idx


def maxpool2x2_bwd(dp, idx, shape):
    B, C, H, W = shape
    d = np.zeros((B, C, H // 2, W // 2, 4), dtype=dp.dtype)
    np.put_along_axis(d, idx[..., None], dp[..., None], -1)
    return d.reshape(B, C, H // 2, W // 2, 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(B, C, H, W)


def deconv2x2_fwd(a, w, b):
    """ConvTranspose2d k=2 s=2.  a: (B,Cin,H,W), w: (Cin,Cout,2,2), b: (Cout,)."""
    B, Ci, H, W = a.shape
    Co = w.shape[1]
    u = np.einsum("bihw,iokl->bohkwl", a, w).reshape(B, Co, 2 * H, 2 * W)
    return u + b[None, :, None, None]


def deconv2x2_bwd(a, w, du):
    B, Ci, H, W = a.shape
    Co = w.shape[1]
    du6 = du.reshape(B, Co, H, 2, W, 2)
    da = np.einsum("bohkwl,iokl->bihw", du6, w)
    dw = np.einsum("bihw,bohkwl->iokl", a, du6)
    db = du.sum(axis=(0, 2, 3))
    return da, dw, db


def softmax_c(logits):
    m = logits.max(axis=1, keepdims=True)
    e = np.exp(logits - m)
    return e / e.sum(axis=1, keepdims=True)


def loss_head_fwd(logits, target, w_ce=1.0, w_dice=0.0, dice_eps=1e-7):
    """Per-pixel CE (mean over pixels) + soft Dice over classes.  target: (B,H,W) ints."""
    B, C, H, W = logits.shape
    p = softmax_c(logits)
    m = logits.max(axis=1, keepdims=True)
    logp = logits - m - np.log(np.exp(logits - m).sum(axis=1, keepdims=True))
    onehot = (np.arange(C)[None, :, None, None] == target[:, None]).astype(logits.dtype)
    n = B * H * W
    ce = -(logp * onehot).sum() / n
    inter = (p * onehot).sum(axis=(0, 2, 3))
    psum = p.sum(axis=(0, 2, 3))
    ysum = onehot.sum(axis=(0, 2, 3))
    dice = 1.0 - ((2 * inter + dice_eps) / (psum + ysum + dice_eps)).mean()
    loss = w_ce * ce + w_dice * dice
    cache = (p, onehot, inter, psum, ysum, n)
    return loss, ce, dice, cache


def loss_head_bwd(cache, w_ce=1.0, w_dice=0.0, dice_eps=1e-7):
    """d loss / d logits."""
    p, onehot, inter, psum, ysum, n = cache
    C = p.shape[1]
    dlogits = w_ce * (p - onehot) / n
    if w_dice != 0.0:
        den = psum + ysum + dice_eps
        num = 2 * inter + dice_eps
        # d dice_c / d p = (2 y den - num) / den^2 ; loss_dice = 1 - mean_c dice_c
        dp = -(w_dice / C) * (2 * onehot * den[None, :, None, None] - num[None, :, None, None]) \
            / (den ** 2)[None, :, None, None]
        dlogits = dlogits + p * (dp - (p * dp).sum(axis=1, keepdims=True))
    return dlogits


def softmax_bwd(p, dp):
    return p * (dp - (p * dp).sum(axis=1, keepdims=True))


# --------------------------------------------------------------------------------------------
# the U-Net (reference: YNet_2022.py:509-602)
# --------------------------------------------------------------------------------------------
BLOCKS_ENC = ["enc1", "enc2", "enc3", "enc4"]


def _block_names(level):
    """(module attribute, layer prefix) for the nine conv blocks in forward order."""
    return {
        "enc1": ("encoder1", "enc1"), "enc2": ("encoder2", "enc2"), "enc3": ("encoder3", "enc3"),
        "enc4": ("encoder4", "enc4"), "bott": ("bottleneck", "bottleneck"),
        "dec4": ("decoder4", "dec4"), "dec3": ("decoder3", "dec3"), "dec2": ("decoder2", "dec2"),
        "dec1": ("decoder1", "dec1"),
    }[level]


def round_bf16(a):
    """float -> nearest bfloat16 (ties to even) -> back, on the float32 bit pattern (NaN-free inputs)."""
    u = np.ascontiguousarray(np.asarray(a, np.float32)).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).astype(np.float64)


class OracleUNet:
    """State is a dict with the reference's state_dict keys (numpy arrays).

    storage=None restates the reference (every tensor in `dtype`).  storage="bf16" additionally rounds to
    bfloat16 exactly where the HIP production path STORES a tensor in bf16 -- network input, packed conv /
    deconv weights, raw conv outputs y (the BatchNorm statistics are taken before that rounding, from the fp32
    accumulators), the activated tensor a conv stages, pooled tensors, deconv outputs, dlogits, every activation
    gradient dA / dY -- and nowhere else (parameters, statistics, weight gradients and the loss stay wide).  The
    bf16 kernels are compared with THIS model, which removes the 1 %-of-ReLU-masks-flip noise that a comparison
    with the unrounded reference carries."""

    def __init__(self, state, dtype=np.float64, storage=None):
        self.dtype = dtype
        assert storage in (None, "bf16")
        if storage == "bf16":
            # rounded values come back in `dtype`: with dtype=float32 the contractions accumulate in fp32 like the
            # MFMA accumulators do (in numpy's blocked order, not the kernel's)
            self.q = (lambda a: round_bf16(a).astype(dtype)) if dtype != np.float64 else round_bf16
        else:
            self.q = lambda a: a
        self.trace = None                  # set to a dict to keep the stored conv outputs / gradients per layer
        self.s = {k: (np.array(v, dtype=dtype) if np.issubdtype(np.asarray(v).dtype, np.floating)
                      else np.array(v)) for k, v in state.items()}
        self.mom = {}

    # -- helpers --------------------------------------------------------------------------
    def _conv_bn_relu(self, x, mod, pre, i, train, cache):
        q = self.q
        w = q(self.s[f"{mod}.{pre}conv{i}.weight"])
        gk, bk = f"{mod}.{pre}norm{i}.weight", f"{mod}.{pre}norm{i}.bias"
        rmk, rvk = f"{mod}.{pre}norm{i}.running_mean", f"{mod}.{pre}norm{i}.running_var"
        nbk = f"{mod}.{pre}norm{i}.num_batches_tracked"
        yacc = conv3x3_fwd(x, w)           # the accumulators: statistics come from these
        y = q(yacc)                        # what is stored and re-read
        if train:
            _, mean, var, invstd, _ = bn_train_fwd(yacc, self.s[gk], self.s[bk])
            xhat = (y - mean[None, :, None, None]) * invstd[None, :, None, None]
            z = xhat * self.s[gk][None, :, None, None] + self.s[bk][None, :, None, None]
            n = y.shape[0] * y.shape[2] * y.shape[3]
            self.s[rmk], self.s[rvk] = bn_running_update(self.s[rmk], self.s[rvk], mean, var, n)
            self.s[nbk] = self.s[nbk] + 1
        else:
            z = bn_eval_fwd(y, self.s[gk], self.s[bk], self.s[rmk], self.s[rvk])
            xhat = invstd = None
        a = np.maximum(z, 0)
        cache.append((f"{mod}.{pre}conv{i}.weight", gk, bk, x, xhat, invstd, z))
        if self.trace is not None:
            self.trace["y:" + f"{mod}.{pre}conv{i}.weight"] = y
        self._a_wide = a                   # the head reads the activation before it is rounded for staging
        return q(a)

    def _block(self, x, level, train, caches):
        mod, pre = _block_names(level)
        c = []
        a1 = self._conv_bn_relu(x, mod, pre, 1, train, c)
        a2 = self._conv_bn_relu(a1, mod, pre, 2, train, c)
        caches[level] = c
        return a2

    # -- forward ---------------------------------------------------------------------------
    def forward(self, x, train=True):
        x = np.asarray(x, dtype=self.dtype)
        if x.shape[2] % 16 or x.shape[3] % 16:
            # reference: torch.cat raises at YNet_2022.py:557
            raise RuntimeError("Sizes of tensors must match except in dimension 1")
        caches = {}
        skips, pools = {}, {}
        h = self.q(x)
        for lv in BLOCKS_ENC:
            a = self._block(h, lv, train, caches)
            skips[lv] = a
            # the pooling kernel compares the activations BEFORE they are rounded for storage (two neighbours that
            # share a bf16 bucket do not tie), and the backward routing re-derives the same winner from y
            h, idx = maxpool2x2_fwd(self._a_wide)
            h = self.q(h)
            pools[lv] = (idx, a.shape)
        h = self._block(h, "bott", train, caches)
        ups = {}
        for k, lv in zip([4, 3, 2, 1], ["dec4", "dec3", "dec2", "dec1"]):
            w, b = self.s[f"upconv{k}.weight"], self.s[f"upconv{k}.bias"]
            ups[k] = h
            u = self.q(deconv2x2_fwd(h, self.q(w), b))
            cat = np.concatenate([u, skips[f"enc{k}"]], axis=1)  # decoder channels first (:557)
            h = self._block(cat, lv, train, caches)
        wh, bh = self.s["conv.weight"], self.s["conv.bias"]
        logits = np.einsum("bchw,oc->bohw", self._a_wide, wh[:, :, 0, 0]) + bh[None, :, None, None]
        self._cache = (caches, pools, ups, h)
        self.logits = logits
        return softmax_c(logits)

    # -- backward from d loss / d logits ----------------------------------------------------
    def backward(self, dlogits):
        caches, pools, ups, hlast = self._cache
        q = self.q
        g = {}
        wh = self.s["conv.weight"]
        dlogits = q(dlogits)
        g["conv.weight"] = np.einsum("bohw,bchw->oc", dlogits, hlast)[:, :, None, None]
        g["conv.bias"] = dlogits.sum(axis=(0, 2, 3))
        da = q(np.einsum("bohw,oc->bchw", dlogits, wh[:, :, 0, 0]))

        def block_bwd(level, da, need_dx=True):
            for (wk, gk, bk, xin, xhat, invstd, z) in reversed(caches[level]):
                dz = q(da * (z > 0))
                dy, dgam, dbet = bn_train_bwd(dz, xhat, self.s[gk], invstd)
                dy = q(dy)
                if self.trace is not None:
                    self.trace["g:" + wk], self.trace["dy:" + wk] = dz, dy
                g[gk], g[bk] = dgam, dbet
                first = wk.endswith("conv1.weight")
                dx, dw = conv3x3_bwd(xin, q(self.s[wk]), dy, need_dx=(need_dx or not first))
                g[wk] = dw
                da = q(dx) if dx is not None else None
            return da

        dskip = {}
        for k, lv in zip([1, 2, 3, 4], ["dec1", "dec2", "dec3", "dec4"]):
            dcat = block_bwd(lv, da)
            co = self.s[f"upconv{k}.weight"].shape[1]
            du, dskip[k] = dcat[:, :co], dcat[:, co:]
            da, dw, db = deconv2x2_bwd(ups[k], q(self.s[f"upconv{k}.weight"]), du)
            da = q(da)
            g[f"upconv{k}.weight"], g[f"upconv{k}.bias"] = dw, db
        dp = block_bwd("bott", da)
        for k in [4, 3, 2, 1]:
            idx, shp = pools[f"enc{k}"]
            da = maxpool2x2_bwd(dp, idx, shp) + dskip[k]
            dp = block_bwd(f"enc{k}", da, need_dx=(k != 1))
        return g

    # -- one training step -----------------------------------------------------------------
    def loss_and_grads(self, x, target, w_ce=1.0, w_dice=0.0, dice_eps=1e-7):
        probs = self.forward(x, train=True)
        loss, ce, dice, cache = loss_head_fwd(self.logits, np.asarray(target), w_ce, w_dice, dice_eps)
        grads = self.backward(loss_head_bwd(cache, w_ce, w_dice, dice_eps))
        return probs, (loss, ce, dice), grads

    def sgd_step(self, grads, lr, momentum=0.0):
        """torch.optim.SGD semantics (dampening 0, no nesterov, no weight decay)."""
        for k, gr in grads.items():
            if momentum != 0.0:
                if k not in self.mom:
                    self.mom[k] = np.array(gr, copy=True)
                else:
                    self.mom[k] = momentum * self.mom[k] + gr
                gr = self.mom[k]
            self.s[k] = self.s[k] - lr * gr


class OracleBioUNet:
    """`UNet` of SOTAS/Layers_Segment/BioNet_2020.py:24-75 restated on the primitives above: four
    bias-conv blocks 64/128/256/512 with three poolings (:55-59), ConvTranspose + `cat([enc, dec])`
    + block three times (:61-73), 1x1 `final` giving raw logits (:75).

    Pinned by tests/golden/bionet_unet_*.npz, produced by the reference's own class
    (tools/gen_golden_bionet.py), and cross-checked against an independent torch.nn restatement
    (oracle/torch_unet.TorchBioUNet) in tests/test_oracle.py."""

    ENC = ["enc1", "enc2", "enc3", "enc4"]
    DEC = [("up4", "dec4"), ("up3", "dec3"), ("up2", "dec2")]

    def __init__(self, state, dtype=np.float64):
        self.dtype = dtype
        self.s = {k: (np.array(v, dtype=dtype) if np.issubdtype(np.asarray(v).dtype, np.floating)
                      else np.array(v)) for k, v in state.items()}

    def _block(self, x, mod, train, caches):
        c = []
        for ci, ni in ((0, 1), (3, 4)):
            wk, cbk = f"{mod}.{ci}.weight", f"{mod}.{ci}.bias"
            gk, bk = f"{mod}.{ni}.weight", f"{mod}.{ni}.bias"
            rmk, rvk, nbk = f"{mod}.{ni}.running_mean", f"{mod}.{ni}.running_var", f"{mod}.{ni}.num_batches_tracked"
            y = conv3x3_fwd(x, self.s[wk], self.s[cbk])
            if train:
                z, mean, var, invstd, xhat = bn_train_fwd(y, self.s[gk], self.s[bk])
                n = y.shape[0] * y.shape[2] * y.shape[3]
                self.s[rmk], self.s[rvk] = bn_running_update(self.s[rmk], self.s[rvk], mean, var, n)
                self.s[nbk] = self.s[nbk] + 1
            else:
                z = bn_eval_fwd(y, self.s[gk], self.s[bk], self.s[rmk], self.s[rvk])
                xhat = invstd = None
            c.append((wk, cbk, gk, bk, x, xhat, invstd, z))
            x = np.maximum(z, 0)
        caches[mod] = c
        return x

    def forward(self, x, train=True):
        x = np.asarray(x, dtype=self.dtype)
        if x.shape[2] % 8 or x.shape[3] % 8:
            raise RuntimeError("Sizes of tensors must match except in dimension 1")  # torch.cat, :64
        caches, skips, pools = {}, [], []
        h = x
        for i, mod in enumerate(self.ENC):
            h = self._block(h, mod, train, caches)
            if i < 3:
                skips.append(h)
                a = h
                h, idx = maxpool2x2_fwd(a)
                pools.append((idx, a.shape))
        ups = []
        for di, (up, mod) in enumerate(self.DEC):
            ups.append(h)
            u = deconv2x2_fwd(h, self.s[f"{up}.weight"], self.s[f"{up}.bias"])
            h = self._block(np.concatenate([skips[2 - di], u], axis=1), mod, train, caches)  # encoder first (:64)
        logits = np.einsum("bchw,oc->bohw", h, self.s["final.weight"][:, :, 0, 0]) + self.s["final.bias"][None, :, None, None]
        self._cache = (caches, pools, ups, h)
        self.logits = logits
        return logits

    def backward(self, dlogits):
        caches, pools, ups, hlast = self._cache
        g = {}
        g["final.weight"] = np.einsum("bohw,bchw->oc", dlogits, hlast)[:, :, None, None]
        g["final.bias"] = dlogits.sum(axis=(0, 2, 3))
        da = np.einsum("bohw,oc->bchw", dlogits, self.s["final.weight"][:, :, 0, 0])

        def block_bwd(mod, da, need_dx=True):
            for j, (wk, cbk, gk, bk, xin, xhat, invstd, z) in enumerate(reversed(caches[mod])):
                dz = da * (z > 0)
                dy, g[gk], g[bk] = bn_train_bwd(dz, xhat, self.s[gk], invstd)
                g[cbk] = dy.sum(axis=(0, 2, 3))          # analytically zero (BN removes the mean)
                da, g[wk] = conv3x3_bwd(xin, self.s[wk], dy, need_dx=(need_dx or j == 0))
            return da

        dskip = [None] * 3
        for di in (2, 1, 0):
            up, mod = self.DEC[di]
            dcat = block_bwd(mod, da)
            cs = dcat.shape[1] - self.s[f"{up}.weight"].shape[1]
            dskip[2 - di], du = dcat[:, :cs], dcat[:, cs:]
            da, g[f"{up}.weight"], g[f"{up}.bias"] = deconv2x2_bwd(ups[di], self.s[f"{up}.weight"], du)
        dp = block_bwd("enc4", da)
        for i in (2, 1, 0):
            idx, shp = pools[i]
            dp = block_bwd(self.ENC[i], maxpool2x2_bwd(dp, idx, shp) + dskip[i], need_dx=(i != 0))
        return g

    def loss_and_grads(self, x, target, w_ce=1.0, w_dice=0.0, dice_eps=1e-7):
        logits = self.forward(x, train=True)
        loss, ce, dice, cache = loss_head_fwd(logits, np.asarray(target), w_ce, w_dice, dice_eps)
        return logits, (loss, ce, dice), self.backward(loss_head_bwd(cache, w_ce, w_dice, dice_eps))


# --------------------------------------------------------------------------------------------
# Metrics (reference: Metrics/Region_based_metrics.py, Metrics/ConfusionMatrix_based_metrics.py)
# --------------------------------------------------------------------------------------------
def confusion_sums(y_true, y_pred):
    """The six sums the reference's formulas are built from, with numpy's dtype semantics:
    products and `1 - y` are evaluated in the input dtype (so uint8 wraps), sums of integer
    arrays are exact, sums of float arrays are returned as float64 here."""
    yt, yp = np.asarray(y_true), np.asarray(y_pred)
    if yt.dtype == np.bool_:
        prod = lambda a, b: np.logical_and(a, b)  # noqa: E731  numpy: bool*bool = and
        one_minus = lambda a: 1 - a  # noqa: E731  -> int64
    else:
        prod = lambda a, b: a * b  # noqa: E731
        one_minus = lambda a: 1 - a  # noqa: E731
    acc = np.float64 if np.issubdtype(yt.dtype, np.floating) else None
    s = lambda a: np.sum(a, dtype=acc) if acc else np.sum(a)  # noqa: E731
    nt, npd = one_minus(yt), one_minus(yp)
    return {
        "tp": s(prod(yt, yp)), "t": s(yt), "p": s(yp),
        "tn": s(nt * npd), "fp": s(nt * yp), "fn": s(yt * npd), "n": int(np.prod(yt.shape)),
    }


def dice_coefficient(y_true, y_pred):  # Region_based_metrics.py:3-16
    c = confusion_sums(y_true, y_pred)
    return (2.0 * c["tp"]) / (c["t"] + c["p"] + 1e-7)


def iou_score(y_true, y_pred):  # :18-31
    c = confusion_sums(y_true, y_pred)
    return c["tp"] / (c["t"] + c["p"] - c["tp"] + 1e-7)


def region_precision(y_true, y_pred):  # :33-46
    c = confusion_sums(y_true, y_pred)
    return c["tp"] / (c["p"] + 1e-7)


def region_recall(y_true, y_pred):  # :48-61
    c = confusion_sums(y_true, y_pred)
    return c["tp"] / (c["t"] + 1e-7)


def accuracy(y_true, y_pred):  # ConfusionMatrix_based_metrics.py:4-18 (no eps)
    c = confusion_sums(y_true, y_pred)
    return (c["tp"] + c["tn"]) / c["n"]


def sensitivity(y_true, y_pred):  # :20-33
    c = confusion_sums(y_true, y_pred)
    return c["tp"] / (c["tp"] + c["fn"] + 1e-7)


def cm_precision(y_true, y_pred):  # :35-48
    c = confusion_sums(y_true, y_pred)
    return c["tp"] / (c["tp"] + c["fp"] + 1e-7)


def specificity(y_true, y_pred):  # :50-63
    c = confusion_sums(y_true, y_pred)
    return c["tn"] / (c["tn"] + c["fp"] + 1e-7)


def mean_squared_error(y_true, y_pred):  # PixelError_based_metrics.py:3-19
    d = np.asarray(y_true).astype(float) - np.asarray(y_pred).astype(float)
    return np.mean(d ** 2)


def root_mean_squared_error(y_true, y_pred):  # :21-37
    return np.sqrt(mean_squared_error(y_true, y_pred))


def thickness_difference(y_true, y_pred):  # Biomarker_based_metrics.py:3-21 (numpy dtype semantics kept)
    return np.mean(np.abs(np.sum(np.asarray(y_true), axis=0) - np.sum(np.asarray(y_pred), axis=0)))


def vascularity_index(y_true, y_pred):  # :23-38
    yt, yp = np.asarray(y_true), np.asarray(y_pred)
    return np.abs(np.sum(yt) / yt.size - np.sum(yp) / yp.size)


METRIC_FUNCS = {
    "region.dice_coefficient": dice_coefficient, "region.iou_score": iou_score,
    "region.precision": region_precision, "region.recall": region_recall,
    "cm.accuracy": accuracy, "cm.sensitivity": sensitivity,
    "cm.precision": cm_precision, "cm.specificity": specificity,
    "pixel.mean_squared_error": mean_squared_error, "pixel.root_mean_squared_error": root_mean_squared_error,
    "bio.thickness_difference": thickness_difference, "bio.vascularity_index": vascularity_index,
}
