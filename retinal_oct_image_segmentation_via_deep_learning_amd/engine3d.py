"""Kernel schedule of the volumetric (3-D) U-Net -- BASELINE configs[4] (cfg5), SURVEY.md §8 f4.

The reference has no 3-D network (only the unused `ffc3d` flag, YNet_2022.py:161,194): the network is defined by
analogy with its 2-D `UNet` (YNet_2022.py:509-602) -- Conv3d(3x3x3, p=1, bias=False) + BatchNorm3d + ReLU twice per
block, MaxPool3d(2), ConvTranspose3d(k=2, s=2, bias), cat((dec, enc), 1), Conv3d(1x1x1) + channel softmax -- and its
oracle is stock torch.nn on the CPU (oracle/torch_unet3d.py).  PARITY IS UNPINNED BY THE REFERENCE.

Layout: a volume batch (B, C, D, H, W) lives in HBM as NDHWC = B*D channels-last images, so everything per-voxel
(BatchNorm statistics and backward, ReLU, the loss head, layout conversion) is the 2-D path's kernel on n = B*D
images.  What changes is the contraction:
  Conv3d           implicit GEMM with K = 3 * Cin per (kh, kw) tap: depth tap kd reads slice d + kd - 1 of the same
                   volume (zero outside) -- the "virtual concat" of three depth-shifted sources (OctConvDesc.depth)
  its dW           three launches, one per depth tap, input slice shifted by kd - 1 (OctWgradDesc.in_img_shift)
  ConvTranspose3d  two depth-to-space launches (kd = 0, 1) writing slice 2d + kd; dA in one launch with
                   K = 8 * Cout gathered from slices 2d + kd; dW per kd with dY gathered from slice 2d + kd
  MaxPool3d(2)     2x2 pooling inside each slice (fused with BN + ReLU), then a pairwise maximum over slices
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .engine import (BN_EPS, BN_MOMENTUM, BNState, BlockSpec, ConvKeys, ConvRec, Ctx, NetSpec, Src, UNetEngine, _stream,
                     ynet_unet_spec)


class UNet3DEngine(UNetEngine):
    def __init__(self, in_channels: int, out_channels: int, features: int = 32, dtype: str = "bf16"):
        super().__init__(in_channels, out_channels, features, dtype, spec=ynet_unet_spec(in_channels, out_channels, features))
        self.supports_frozen_bwd = False   # (no reference counterpart; eval-mode forward is inference only)

    # ---- forward ------------------------------------------------------------------------------------
    def _conv_bn3(self, P, keys: ConvKeys, src: Src, cout, n, h, w, depth, train) -> ConvRec:
        wkey, nk = keys.w, keys.bn
        dev = P[wkey].device
        wp = self._pack(wkey, P[wkey], L.PACK_CONV3D_FPROP, cout, src.channels)
        y = self._act(n, h, w, cout, dev)
        scale = torch.empty(cout, dtype=torch.float32, device=dev)
        shift = torch.empty_like(scale)
        if train:
            nblk = self._stat_blocks(cout, n, h, w, src, depth=depth)
            partials = torch.empty((nblk, 2, cout), dtype=torch.float32, device=dev)
            self._conv(src, wp, cout, 9, n, h, w, y, stats=partials, depth=depth)
            mean, invstd = torch.empty_like(scale), torch.empty_like(scale)
            L.check(L.lib().oct_bn_finalize(
                partials.data_ptr(), nblk, cout, float(n * h * w), P[nk + ".weight"].data_ptr(),
                P[nk + ".bias"].data_ptr(), BN_EPS, BN_MOMENTUM, P[nk + ".running_mean"].data_ptr(),
                P[nk + ".running_var"].data_ptr(), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                shift.data_ptr(), None, _stream()), "oct_bn_finalize")
            self._nbt.append(P[nk + ".num_batches_tracked"])   # bumped by ONE multi-tensor launch at the end of forward
            bn = BNState(scale, shift, mean, invstd)
        else:
            self._conv(src, wp, cout, 9, n, h, w, y, depth=depth)
            L.check(L.lib().oct_bn_eval_coeffs(
                cout, P[nk + ".weight"].data_ptr(), P[nk + ".bias"].data_ptr(), P[nk + ".running_mean"].data_ptr(),
                P[nk + ".running_var"].data_ptr(), BN_EPS, scale.data_ptr(), shift.data_ptr(), None, _stream()),
                "oct_bn_eval_coeffs")
            bn = BNState(scale, shift)
        rec = ConvRec(wkey, nk + ".weight", nk + ".bias", None, src, y, bn, cout, n, h, w)
        rec.depth = depth
        return rec

    def _block3(self, P, blk: BlockSpec, src, n, h, w, depth, train, ctx):
        r1 = self._conv_bn3(P, blk.c1, src, blk.cout, n, h, w, depth, train)
        r2 = self._conv_bn3(P, blk.c2, Src(r1.y, blk.cout, r1.bn), blk.cout, n, h, w, depth, train)
        ctx.convs[blk.name] = [r1, r2]
        return r2

    def forward(self, P: dict, x: torch.Tensor, train: bool, target: torch.Tensor | None = None,
                loss_cfg=(1.0, 0.0, 1e-7), want_probs=True, want_argmax=False, want_logits=False, defer_loss=False):
        """x: (B, Cin, D, H, W).  Returns (ctx, probs (B,C,D,H,W) | None, argmax (B,D,H,W) | None, logits | None)."""
        lib = L.lib()
        if x.dim() != 5 or x.shape[1] != self.cin:
            raise RuntimeError(f"expected input (B,{self.cin},D,H,W), got {tuple(x.shape)}")
        b, _, dd, h, w = x.shape
        sp = self.spec
        if dd % sp.divisor or h % sp.divisor or w % sp.divisor:
            raise RuntimeError(f"Sizes of tensors must match except in dimension 1. Volume {dd}x{h}x{w} is not divisible by "
                               f"{sp.divisor} ({len(sp.enc) - 1} 2x2x2 poolings followed by as many 2x up-samplings)")
        dev = x.device
        if dev.type != "cuda":
            raise L.OctError("the HIP path needs a device tensor (there is no CPU fallback)")
        xf = x.detach().to(torch.float32).contiguous()
        ctx = Ctx(n=b, h=dd * h, w=w, loss_cfg=tuple(loss_cfg))      # head geometry: (B, D*H, W) == (B, D, H, W) flattened
        ctx.vol = (b, dd, h, w)
        ctx.pool2 = {}
        self._nbt = []
        n = b * dd
        xt = self._act(n, h, w, self.cin, dev)
        L.check(lib.oct_nchw_to_nhwc(self.dt, xf.data_ptr(), xt.data_ptr(), b, self.cin, dd * h, w, _stream()), "oct_nchw_to_nhwc")
        src = Src(xt, self.cin)
        hh, ww, dl = h, w, dd
        skips = []
        prev = None
        for li, blk in enumerate(sp.enc):
            prev = self._block3(P, blk, src, b * dl, hh, ww, dl, train, ctx)
            if li == len(sp.enc) - 1:
                break
            skips.append(prev)
            p2 = self._act(b * dl, hh // 2, ww // 2, blk.cout, dev)
            L.check(lib.oct_bn_relu_pool_fwd(self.dt, prev.y.data_ptr(), prev.bn.scale.data_ptr(), prev.bn.shift.data_ptr(),
                                             p2.data_ptr(), b * dl, hh, ww, blk.cout, _stream()), "oct_bn_relu_pool_fwd")
            pooled = self._act(b * dl // 2, hh // 2, ww // 2, blk.cout, dev)
            L.check(lib.oct_depth_pool_fwd(self.dt, p2.data_ptr(), pooled.data_ptr(), b * dl // 2,
                                           (hh // 2) * (ww // 2) * blk.cout, _stream()), "oct_depth_pool_fwd")
            ctx.pool2[blk.name] = p2
            hh //= 2
            ww //= 2
            dl //= 2
            src = Src(pooled, blk.cout)
        for di, ((wkey, bkey, cout_d), blk) in enumerate(zip(sp.ups, sp.dec)):
            cin_d = prev.cout
            u = self._act(2 * b * dl, hh * 2, ww * 2, cout_d, dev)
            for kdi in (0, 1):      # ConvTranspose3d(k2, s2): slice 2d + kdi of the output from slice d of the input
                wp = self._pack(f"{wkey}#{kdi}", P[wkey], L.PACK_DECONV3D_FPROP, cout_d, cin_d, kdi)
                self._conv(Src(prev.y, cin_d, prev.bn), wp, 4 * cout_d, 1, b * dl, hh, ww, u, out_mode=L.OUT_D2S,
                           bias=P[bkey], oimg=(2, kdi))
            ctx.ups[di] = (prev, u)
            hh *= 2
            ww *= 2
            dl *= 2
            sk = skips[len(skips) - 1 - di]
            src = Src(u, cout_d, None, sk.y, sk.cout, sk.bn)       # cat((dec, enc), 1)
            prev = self._block3(P, blk, src, b * dl, hh, ww, dl, train, ctx)
        ctx.head_in = prev
        if self._nbt:       # BatchNorm step counters: one multi-tensor launch instead of one per layer
            torch._foreach_add_(self._nbt, 1)
            self._nbt = []
        hd = L.HeadDesc(self.dt, b, dd * h, w, self.f, self.ncls)
        probs = torch.empty((b, self.ncls, dd, h, w), dtype=torch.float32, device=dev) if want_probs else None
        amax = torch.empty((b, dd, h, w), dtype=torch.int64, device=dev) if want_argmax else None
        logits = torch.empty((b, self.ncls, dd, h, w), dtype=torch.float32, device=dev) if want_logits else None
        partials = None
        if target is not None:
            if target.shape != (b, dd, h, w):
                raise RuntimeError(f"target must be (B,D,H,W)={b, dd, h, w}, got {tuple(target.shape)}")
            target = target.to(device=dev, dtype=torch.int64).contiguous()
            nb = lib.oct_head_blocks(C.byref(hd))
            partials = torch.empty((nb, L.HEAD_LOSS_SLOTS), dtype=torch.float64, device=dev)
            if (defer_loss and loss_cfg[1] == 0.0 and self.f == 32 and self.ncls <= 8
                    and not (want_probs or want_argmax or want_logits)):
                ctx.target = target
                ctx.loss = torch.empty(3, dtype=torch.float32, device=dev)
                ctx.dice_coef = torch.zeros(2 * L.MAX_CLASSES, dtype=torch.float32, device=dev)
                ctx.loss_partials = partials
                return ctx, None, None, None
        L.check(lib.oct_head_forward(C.byref(hd), prev.y.data_ptr(), prev.bn.scale.data_ptr(), prev.bn.shift.data_ptr(),
                                     P[sp.head_w].data_ptr(), P[sp.head_b].data_ptr(), L.ptr(target), L.ptr(probs),
                                     L.ptr(amax), L.ptr(logits), L.ptr(partials), _stream()), "oct_head_forward")
        if target is not None:
            w_ce, w_dice, eps = loss_cfg
            ctx.target = target
            ctx.loss = torch.empty(3, dtype=torch.float32, device=dev)
            ctx.dice_coef = torch.zeros(2 * L.MAX_CLASSES, dtype=torch.float32, device=dev)
            L.check(lib.oct_head_loss_finalize(C.byref(hd), partials.data_ptr(), partials.shape[0], w_ce, w_dice, eps,
                                               ctx.loss.data_ptr(), ctx.dice_coef.data_ptr(), _stream()),
                    "oct_head_loss_finalize")
        return ctx, probs, amax, logits

    # ---- backward -----------------------------------------------------------------------------------
    def _conv_backward(self, rec: ConvRec, dy, G, accumulate, need_dx=True):
        """Conv3d: dW from three depth-tap launches, dA from one depth-tap GEMM with the point-reflected filter."""
        n, h, w, depth = rec.n, rec.h, rec.w, rec.depth
        src = rec.src
        cin = src.channels
        slab = self._dwp_take(27 * rec.cout * cin, dy.device).view(3, 9, rec.cout, cin)
        qd = L.WgradDesc(self.dt, n, h, w, src.c0, src.c1, rec.cout, 9, 0, 0, L.IN_PLAIN, 0, 0, depth, L.IMG_SHIFT_ALL, 0, 0, 0)
        if src.bn0 is None and src.c1 == 0 and not self.deterministic and L.lib().oct_conv_wgrad_all_depth_taps_ok(C.byref(qd)):
            # first layer (Cin = 1): the matrix-pipe kernel takes all 27 taps in one pass over dY
            self._wgrad(src, dy, rec.cout, 9, n, h, w, depth=depth, in_shift=L.IMG_SHIFT_ALL, dwp=slab)
        else:
            for kdi in range(3):
                self._wgrad(src, dy, rec.cout, 9, n, h, w, depth=depth, in_shift=kdi - 1, dwp=slab[kdi])
        L.check(L.lib().oct_unpack_wgrad3d(L.PACK_CONV3D_FPROP, slab.data_ptr(), G[rec.wkey].data_ptr(), rec.cout, cin, 0,
                                           int(accumulate), _stream()), "oct_unpack_wgrad3d")
        if not need_dx:
            return None, None
        wp = self._pack(rec.wkey, self._P[rec.wkey], L.PACK_CONV3D_DGRAD, rec.cout, cin)
        d0 = self._act(n, h, w, src.c0, dy.device)
        d1 = self._act(n, h, w, src.c1, dy.device) if src.c1 else None
        self._conv(Src(dy, rec.cout), wp, cin, 9, n, h, w, d0, y1=d1, split=src.c0 if src.c1 else 0, depth=depth)
        return d0, d1

    def _block_backward(self, name, da, dpool, G, accumulate, need_dx=True, partials=None):
        r1, r2 = self._ctx.convs[name]
        dy2 = self._bn_backward(r2, da, dpool, G, accumulate, partials=partials)
        da1, _ = self._conv_backward(r2, dy2, G, accumulate)
        dy1 = self._bn_backward(r1, da1, None, G, accumulate)
        return self._conv_backward(r1, dy1, G, accumulate, need_dx=need_dx)

    def backward(self, P: dict, ctx: Ctx, G: dict, dprobs: torch.Tensor | None = None, accumulate=False,
                 dlogits: torch.Tensor | None = None, stage_hook=None):
        lib = L.lib()
        sp = self.spec
        self._P, self._ctx = P, ctx
        b, dd, h, w = ctx.vol
        f, ncls = self.f, self.ncls
        rec = ctx.head_in
        dev = rec.y.device
        self._arena_begin(dev)
        hd = L.HeadDesc(self.dt, b, dd * h, w, f, ncls)
        nimg = b * dd
        if dprobs is not None:
            dprobs = dprobs.to(torch.float32).contiguous()
            tgt, dc, w_ce = None, None, 0.0
        else:
            if ctx.target is None:
                raise RuntimeError("backward without an output gradient needs forward(target=...)")
            tgt, dc, w_ce = ctx.target, ctx.dice_coef, ctx.loss_cfg[0]
        hw_, hb = P[sp.head_w], P[sp.head_b]
        bgrad, wgrad_t = G[sp.head_b], G[sp.head_w]
        head_partials = None
        if f == 32 and ncls <= 8:
            if not accumulate:
                bgrad.zero_()
                wgrad_t.zero_()
            da = self._act(nimg, h, w, f, dev)
            nb = lib.oct_head_blocks(C.byref(hd))
            head_partials = torch.empty((nb, 2, f), dtype=torch.float32, device=dev)
            L.check(lib.oct_head_backward_fused(
                C.byref(hd), rec.y.data_ptr(), rec.bn.scale.data_ptr(), rec.bn.shift.data_ptr(), rec.bn.mean.data_ptr(),
                rec.bn.invstd.data_ptr(), hw_.data_ptr(), hb.data_ptr(), L.ptr(tgt), L.ptr(dc), w_ce, L.ptr(dprobs), None,
                da.data_ptr(), head_partials.data_ptr(), bgrad.data_ptr(), wgrad_t.data_ptr(),
                L.ptr(ctx.loss_partials) if dprobs is None else None, _stream()), "oct_head_backward_fused")
            if ctx.loss_partials is not None and dprobs is None:
                w_ce_, w_dice_, eps_ = ctx.loss_cfg
                L.check(lib.oct_head_loss_finalize(C.byref(hd), ctx.loss_partials.data_ptr(), ctx.loss_partials.shape[0],
                                                   w_ce_, w_dice_, eps_, ctx.loss.data_ptr(), ctx.dice_coef.data_ptr(),
                                                   _stream()), "oct_head_loss_finalize")
        else:
            dl = self._act(nimg, h, w, ncls, dev)
            L.check(lib.oct_head_dlogits(C.byref(hd), rec.y.data_ptr(), rec.bn.scale.data_ptr(), rec.bn.shift.data_ptr(),
                                         hw_.data_ptr(), hb.data_ptr(), L.ptr(tgt), L.ptr(dc), w_ce, L.ptr(dprobs),
                                         dl.data_ptr(), _stream()), "oct_head_dlogits")
            L.check(lib.oct_channel_sum(self.dt, dl.data_ptr(), bgrad.data_ptr(), nimg * h * w, ncls, int(accumulate), _stream()),
                    "oct_channel_sum")
            wp = self._pack(sp.head_w, hw_, L.PACK_1X1_DGRAD, ncls, f)
            da = self._act(nimg, h, w, f, dev)
            self._conv(Src(dl, ncls), wp, f, 1, nimg, h, w, da)
            dwp = self._wgrad(Src(rec.y, f, rec.bn), dl, ncls, 1, nimg, h, w)
            self._unpack(L.PACK_1X1_FPROP, dwp, wgrad_t, ncls, f, accumulate)
        nd = len(sp.dec)
        dskip = [None] * nd
        for di in range(nd - 1, -1, -1):
            du, dskip[di] = self._block_backward(sp.dec[di].name, da, None, G, accumulate,
                                                 partials=head_partials if di == nd - 1 else None)
            wkey, bkey, cout_d = sp.ups[di]
            prev, _ = ctx.ups[di]
            cin_d, nl, hl, wl = prev.cout, prev.n, prev.h, prev.w
            bg = G[bkey]
            if not accumulate:
                bg.zero_()
            for kdi in (0, 1):
                dwp = self._wgrad(Src(prev.y, cin_d, prev.bn), du, 4 * cout_d, 1, nl, hl, wl, dy_mode=L.IN_S2D, dbias=bg,
                                  dy_img=(2, kdi), partials_ok=False)
                L.check(lib.oct_unpack_wgrad3d(L.PACK_DECONV3D_FPROP, dwp.data_ptr(), G[wkey].data_ptr(), cout_d, cin_d, kdi,
                                               int(accumulate), _stream()), "oct_unpack_wgrad3d")
            wp = self._pack(wkey, P[wkey], L.PACK_DECONV3D_DGRAD, cout_d, cin_d)
            da = self._act(nl, hl, wl, cin_d, dev)
            self._conv(Src(du, cout_d), wp, cin_d, 1, nl, hl, wl, da, in_mode=L.IN_S2D, depth=prev.depth)
            self._stage_done(stage_hook, nd - 1 - di)
        dpool, _ = self._block_backward(sp.enc[-1].name, da, None, G, accumulate)
        self._stage_done(stage_hook, nd)
        for li in range(len(sp.enc) - 2, -1, -1):
            blk = sp.enc[li]
            p2 = ctx.pool2[blk.name]
            dp2 = torch.empty_like(p2)
            L.check(lib.oct_depth_pool_bwd(self.dt, p2.data_ptr(), dpool.data_ptr(), dp2.data_ptr(), p2.shape[0] // 2,
                                           p2.shape[1] * p2.shape[2] * p2.shape[3], _stream()), "oct_depth_pool_bwd")
            dpool, _ = self._block_backward(blk.name, dskip[nd - 1 - li], dp2, G, accumulate, need_dx=(li != 0))
            if li:
                self._stage_done(stage_hook, nd + len(sp.enc) - 1 - li)
        self._arena_end(dev)
        self._stage_done(stage_hook, nd + len(sp.enc) - 1)
        self._P = self._ctx = None
        return G
