"""Kernel schedule of the U-Net training path on one MI355X.

Host-side plumbing only: tensors come from PyTorch's caching allocator, every arithmetic step is
a call into liboct_hip.so (include/oct_hip.h) on torch's current HIP stream.  The network it
runs is the reference's UNet (SOTAS/Lesions_Segment/YNet_2022.py:509-602): four encoder blocks,
bottleneck, four decoder blocks with ConvTranspose2d up-sampling and (dec, enc) concatenation,
1x1 head + channel softmax.

Data layout in HBM
  activations : NHWC, bf16 (production) or fp32 (parity mode); only RAW conv outputs are stored.
                BatchNorm-apply + ReLU are never materialised -- the consumer conv applies
                a = max(y*scale+shift, 0) while staging its input tile; torch.cat is virtual
                (the consumer reads two base pointers); max-pool output is the one extra tensor.
  parameters  : fp32, torch layout (state_dict compatible); re-packed into MFMA fragment order
                (activation dtype) whenever they change.
  gradients   : activation gradients in the activation dtype, parameter gradients fp32.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import torch

from . import _lib as L

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

@dataclass
class ConvKeys:
    """state_dict keys of one conv + BatchNorm pair"""
    w: str
    bn: str                 # prefix of .weight/.bias/.running_mean/.running_var/.num_batches_tracked
    b: str | None = None    # conv bias (BioNet / MGUNet style blocks); None for bias=False


@dataclass
class BlockSpec:
    name: str
    c1: ConvKeys
    c2: ConvKeys
    cout: int


@dataclass
class NetSpec:
    """Encoder-decoder topology shared by the reference's U-Nets.  `enc` ends with the bottleneck
    (no pooling after it); `ups[i]` / `dec[i]` run in decode order and consume the skip of
    enc[len(enc)-2-i]."""
    cin: int
    ncls: int
    enc: list
    ups: list               # (weight key, bias key, cout)
    dec: list
    head_w: str
    head_b: str
    dec_first: bool         # torch.cat((dec, enc), 1) [YNet_2022.py:557] vs cat([enc, dec]) [BioNet_2020.py:64]
    softmax_out: bool       # UNet of YNet_2022 returns probabilities, BioNet's returns logits

    @property
    def divisor(self) -> int:
        return 1 << (len(self.enc) - 1)

    @property
    def head_feat(self) -> int:
        return self.dec[-1].cout


def ynet_unet_spec(cin: int, ncls: int, f: int) -> NetSpec:
    """UNet of SOTAS/{Lesions,Layers}_Segment/YNet_2022 (reference :509-602)."""
    def blk(mod, pre, cout):
        return BlockSpec(pre, ConvKeys(f"{mod}.{pre}conv1.weight", f"{mod}.{pre}norm1"),
                         ConvKeys(f"{mod}.{pre}conv2.weight", f"{mod}.{pre}norm2"), cout)
    enc = [blk(f"encoder{i + 1}", f"enc{i + 1}", f << i) for i in range(4)] + [blk("bottleneck", "bottleneck", f * 16)]
    ups = [(f"upconv{k}.weight", f"upconv{k}.bias", f << (k - 1)) for k in (4, 3, 2, 1)]
    dec = [blk(f"decoder{k}", f"dec{k}", f << (k - 1)) for k in (4, 3, 2, 1)]
    return NetSpec(cin, ncls, enc, ups, dec, "conv.weight", "conv.bias", dec_first=True, softmax_out=True)


def bionet_unet_spec(cin: int, ncls: int) -> NetSpec:
    """UNet of SOTAS/Layers_Segment/BioNet_2020.py:24-75: three poolings, bias convs, cat([enc, dec]), logits."""
    def blk(mod, cout):
        return BlockSpec(mod, ConvKeys(f"{mod}.0.weight", f"{mod}.1", f"{mod}.0.bias"),
                         ConvKeys(f"{mod}.3.weight", f"{mod}.4", f"{mod}.3.bias"), cout)
    enc = [blk("enc1", 64), blk("enc2", 128), blk("enc3", 256), blk("enc4", 512)]
    ups = [(f"up{k}.weight", f"up{k}.bias", c) for k, c in ((4, 256), (3, 128), (2, 64))]
    dec = [blk("dec4", 256), blk("dec3", 128), blk("dec2", 64)]
    return NetSpec(cin, ncls, enc, ups, dec, "final.weight", "final.bias", dec_first=False, softmax_out=False)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


@dataclass
class Src:
    """Input of a conv: up to two NHWC tensors (virtual concat) with optional BN+ReLU on load."""
    x0: torch.Tensor
    c0: int
    bn0: "BNState | None" = None
    x1: torch.Tensor | None = None
    c1: int = 0
    bn1: "BNState | None" = None

    @property
    def channels(self) -> int:
        return self.c0 + self.c1


@dataclass
class BNState:
    scale: torch.Tensor
    shift: torch.Tensor
    mean: torch.Tensor | None = None
    invstd: torch.Tensor | None = None
    relu: bool = True   # False: the consumer applies scale*x + shift without the ReLU (OCT_XF_AFFINE: a deferred bias add)
    frozen: bool = False   # eval-mode forward: mean / invstd are the RUNNING statistics; backward is a per-channel affine


def _xf(bn) -> int:
    return L.XF_NONE if bn is None else (L.XF_AFFINE_RELU if bn.relu else L.XF_AFFINE)


@dataclass
class ConvRec:
    """What backward needs about one conv+BN(+ReLU) layer."""
    wkey: str
    gkey: str
    bkey: str
    cbkey: str | None      # conv bias key (its gradient is exactly zero in front of a train-mode BN)
    src: Src
    y: torch.Tensor
    bn: BNState
    cout: int
    n: int
    h: int
    w: int


@dataclass
class Ctx:
    n: int = 0
    h: int = 0
    w: int = 0
    convs: dict = field(default_factory=dict)   # block name -> [ConvRec conv1, ConvRec conv2]
    ups: dict = field(default_factory=dict)     # decode index -> (input ConvRec, u tensor)
    head_in: ConvRec | None = None
    target: torch.Tensor | None = None
    dice_coef: torch.Tensor | None = None
    loss: torch.Tensor | None = None
    loss_cfg: tuple = (1.0, 0.0, 1e-7)
    loss_partials: torch.Tensor | None = None   # set when the loss was deferred to the fused head backward


class UNetEngine:
    def __init__(self, in_channels: int, out_channels: int, features: int = 32, dtype: str = "bf16",
                 spec: NetSpec | None = None):
        if out_channels > L.MAX_CLASSES:
            raise L.OctError(f"out_channels={out_channels} exceeds the head kernel's limit {L.MAX_CLASSES}")
        self.spec = spec if spec is not None else ynet_unet_spec(in_channels, out_channels, features)
        self.cin, self.ncls, self.f = self.spec.cin, self.spec.ncls, self.spec.head_feat
        self.set_dtype(dtype)
        self._packed = {}  # (key, mode) -> (version, tensor)
        self._arena, self._arena_on, self._arena_off, self._arena_short = None, False, 0, False
        self._unpack_jobs = []
        self._nbt = []
        self._frozen_bwd = False
        self.supports_frozen_bwd = True   # backward through an eval-mode forward (BatchNorm on running statistics)
        self._consts = {}  # (value, n, device) -> constant fp32 vector (never written)
        # OCT_ROWDOT=0: one-output-channel 1x1 convolutions stay on the padded MFMA kernels (A/B switch, parity tests)
        self.rowdot_off = os.environ.get("OCT_ROWDOT", "1") == "0"
        # deterministic = True: weight gradients through the two-stage reduction (OctWgradDesc.partials: per-workgroup slabs
        # summed in order) instead of fp32 atomics -- bit-identical gradients from run to run (OCT_DETERMINISTIC=1 sets it)
        self.deterministic = os.environ.get("OCT_DETERMINISTIC", "0") == "1"
        self.debug = None  # set to a dict to capture intermediate gradients (tests / probes)
        self.prof = None   # set to a list: (kind, start_event, end_event) around every MFMA launch
        self.prof_labels = None   # set to a list next to `prof`: (taps, cin, cout, n, h, w, in/dy mode, out mode) per launch

    def set_dtype(self, dtype: str) -> None:
        if dtype not in ("bf16", "f32"):
            raise ValueError("dtype must be 'bf16' or 'f32'")
        self.dtype = dtype
        self.dt = L.DT_BF16 if dtype == "bf16" else L.DT_F32
        self.tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
        self._packed = {}
        self._pack_plan = {}

    # ---- small helpers --------------------------------------------------------------------------
    def _prof_begin(self):
        if self.prof is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()  # torch's current stream == the stream the kernel is launched on
        return ev

    def _prof_end(self, ev, kind, label=None):
        if ev is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            self.prof.append((kind, ev, end))
            if self.prof_labels is not None:
                self.prof_labels.append(label)

    def _const(self, value: float, n: int, dev):
        key = (float(value), int(n), str(dev))
        t = self._consts.get(key)
        if t is None:
            t = self._consts[key] = torch.full((n,), float(value), dtype=torch.float32, device=dev)
        return t

    def _act(self, n, h, w, c, dev):
        return torch.empty((n, h, w, c), dtype=self.tdt, device=dev)

    def _pack(self, key: str, wt: torch.Tensor, mode: int, cout: int, cin: int, kdi: int = 0) -> torch.Tensor:
        # packed copies are reused until the parameter changes: torch bumps _version on in-place
        # updates, our own raw-pointer optimizer bumps L.param_generation
        ver = (wt._version, L.param_generation[0])
        hit = self._packed.get((key, mode))
        if hit is not None and hit[0] == ver and hit[2] == wt.data_ptr():
            return hit[1]
        self._pack_plan[(key, mode)] = (cout, cin)
        rows, taps, kch = {
            L.PACK_CONV_FPROP: (cout, 9, cin), L.PACK_CONV_DGRAD: (cin, 9, cout),
            L.PACK_DECONV_FPROP: (4 * cout, 1, cin), L.PACK_DECONV_DGRAD: (cin, 1, 4 * cout),
            L.PACK_1X1_DGRAD: (cin, 1, cout), L.PACK_1X1_FPROP: (cout, 1, cin),
            L.PACK_CONV3D_FPROP: (cout, 9, 3 * cin), L.PACK_CONV3D_DGRAD: (cin, 9, 3 * cout),
            L.PACK_DECONV3D_FPROP: (4 * cout, 1, cin), L.PACK_DECONV3D_DGRAD: (cin, 1, 8 * cout)}[mode]
        elems = L.lib().oct_packed_weight_elems(rows, taps, kch)
        out = hit[1] if hit is not None else torch.empty(elems, dtype=self.tdt, device=wt.device)
        if mode >= L.PACK_CONV3D_FPROP:
            self._pack_plan.pop((key, mode), None)      # the batched re-pack launch knows the 2-D modes only
            L.check(L.lib().oct_pack_weights3d(mode, self.dt, wt.data_ptr(), out.data_ptr(), cout, cin, kdi, _stream()),
                    "oct_pack_weights3d")
        else:
            L.check(L.lib().oct_pack_weights(mode, self.dt, wt.data_ptr(), out.data_ptr(), cout, cin, _stream()),
                    "oct_pack_weights")
        self._packed[(key, mode)] = (ver, out, wt.data_ptr())
        return out

    def _prepack(self, P: dict) -> None:
        """Refresh, in ONE launch, every packed weight a previous step used and the optimizer has since
        changed (the per-layer `_pack` calls of this step then all hit the cache)."""
        jobs, news = [], []
        for (key, mode), (cout, cin) in self._pack_plan.items():
            wt = P.get(key) if isinstance(key, str) else None
            hit = self._packed.get((key, mode))
            if wt is None or hit is None or hit[2] != wt.data_ptr():
                continue
            ver = (wt._version, L.param_generation[0])
            if hit[0] != ver:
                jobs.append(L.PackJob(mode, cout, cin, 0, wt.data_ptr(), hit[1].data_ptr()))
                news.append(((key, mode), (ver, hit[1], hit[2])))
        if jobs:
            arr = (L.PackJob * len(jobs))(*jobs)
            L.check(L.lib().oct_pack_weights_batch(self.dt, len(jobs), arr, _stream()), "oct_pack_weights_batch")
            self._packed.update(news)

    def _conv(self, src: Src, wpacked, cout, taps, n, h, w, y0, *, y1=None, split=0, in_mode=L.IN_PLAIN,
              out_mode=L.OUT_PLAIN, bias=None, stats=None, kh=0, kw=0, depth=0, oimg=(0, 0)):
        """taps 9 / 1 = 3x3 / 1x1; any other kernel passes (kh, kw) and taps = kh*kw (7x3: ReLayNet).
        depth = D > 0: the n images are volumes of D slices and the GEMM gains depth taps (oct_hip.h, OctConvDesc)."""
        d = L.ConvDesc(self.dt, n, h, w, src.c0, src.c1, cout, taps,
                       _xf(src.bn0), _xf(src.bn1),
                       in_mode, out_mode, split, 1 if stats is not None else 0, kh, kw, depth, oimg[0], oimg[1])
        a = L.ConvArgs(L.ptr(src.x0), L.ptr(src.x1),
                       L.ptr(src.bn0.scale) if src.bn0 else None, L.ptr(src.bn0.shift) if src.bn0 else None,
                       L.ptr(src.bn1.scale) if src.bn1 else None, L.ptr(src.bn1.shift) if src.bn1 else None,
                       L.ptr(wpacked), L.ptr(bias), L.ptr(y0), L.ptr(y1), L.ptr(stats))
        ev = self._prof_begin()
        L.check(L.lib().oct_conv_forward(C.byref(d), C.byref(a), _stream()), "oct_conv_forward")
        self._prof_end(ev, "igemm", (taps, src.channels, cout, n, h, w, in_mode, out_mode))

    def _stat_blocks(self, cout, n, h, w, src: Src, taps=9, kh=0, kw=0, depth=0):
        """rows of the partial-statistics buffer the conv with this exact descriptor will write"""
        d = L.ConvDesc(self.dt, n, h, w, src.c0, src.c1, cout, taps,
                       _xf(src.bn0), _xf(src.bn1), 0, 0, 0, 1, kh, kw, depth, 0, 0)
        return L.lib().oct_conv_stat_blocks(C.byref(d))

    def _wgrad(self, src: Src, dy, cout, taps, n, h, w, dy_mode=L.IN_PLAIN, dbias=None, fused_apply=None, kh=0, kw=0,
               depth=0, in_shift=0, dy_img=(0, 0), dwp=None, partials_ok=True):
        """fused_apply = (y, coef, scale, shift): `dy` holds dA and the kernel applies BN backward on load.
        depth / in_shift / dy_img: one depth tap of a 3-D weight gradient (oct_hip.h, OctWgradDesc); dwp: write into this
        (zeroed) slab instead of taking a new one.  partials_ok: the caller unpacks through `_unpack` (OctUnpackJob.nparts
        sums the per-workgroup slabs of deterministic mode); callers of the single-slab unpack entry points
        (oct_unpack_wgrad_kk, oct_unpack_wgrad3d) pass False and stay on atomics."""
        ktot = src.channels
        parts = 1 if (self.deterministic and partials_ok and dwp is None and depth == 0) else 0
        d = L.WgradDesc(self.dt, n, h, w, src.c0, src.c1, cout, taps,
                        _xf(src.bn0), _xf(src.bn1), dy_mode, kh, kw, depth, in_shift,
                        dy_img[0], dy_img[1], parts)
        bias_parts = None
        if parts:
            nparts = L.lib().oct_conv_wgrad_partials(C.byref(d))
            dwp = torch.empty((nparts, taps, cout, ktot), dtype=torch.float32, device=dy.device)   # every slab is written whole
            dwp._oct_nparts = nparts
            if dbias is not None:
                bias_parts = torch.empty((nparts, cout), dtype=torch.float32, device=dy.device)
        elif dwp is None:
            dwp = self._dwp_take(taps * cout * ktot, dy.device).view(taps, cout, ktot)
        a = L.WgradArgs(L.ptr(src.x0), L.ptr(src.x1),
                        L.ptr(src.bn0.scale) if src.bn0 else None, L.ptr(src.bn0.shift) if src.bn0 else None,
                        L.ptr(src.bn1.scale) if src.bn1 else None, L.ptr(src.bn1.shift) if src.bn1 else None,
                        L.ptr(dy), L.ptr(dwp), L.ptr(dbias), *([L.ptr(t) for t in fused_apply] if fused_apply
                                                                 else [None, None, None, None]), L.ptr(bias_parts))
        ev = self._prof_begin()
        L.check(L.lib().oct_conv_wgrad(C.byref(d), C.byref(a), _stream()), "oct_conv_wgrad")
        self._prof_end(ev, "wgrad", (taps, ktot, cout, n, h, w, dy_mode, 0))
        if bias_parts is not None:     # the caller zeroed dbias (atomics contract): add the ordered sum to it
            L.check(L.lib().oct_reduce_bias_partials(bias_parts.data_ptr(), nparts, cout,
                                                     cout // 4 if dy_mode == L.IN_S2D else cout, dbias.data_ptr(), 1,
                                                     _stream()), "oct_reduce_bias_partials")
        return dwp

    def _dwp_take(self, numel: int, dev) -> torch.Tensor:
        """Zeroed fp32 scratch for one weight-gradient launch (the kernels accumulate with atomics).  Inside
        backward() the buffers of a step come out of one arena that is cleared with a single fill."""
        if self._arena_on:
            lo = self._arena_off
            self._arena_off = lo + (numel + 63) // 64 * 64
            if self._arena is not None and self._arena_off <= self._arena.numel() and self._arena.device == dev:
                return self._arena[lo:lo + numel]
            self._arena_short = True
        return torch.zeros(numel, dtype=torch.float32, device=dev)

    def _arena_begin(self, dev):
        self._arena_on, self._arena_off, self._arena_short = True, 0, False
        self._unpack_jobs = []
        if self._arena is not None and self._arena.device == dev:
            self._arena.zero_()

    def _flush_unpack(self):
        """One launch that brings every weight gradient produced so far into torch layout."""
        if self._unpack_jobs:
            arr = (L.UnpackJob * len(self._unpack_jobs))(*[j[0] for j in self._unpack_jobs])
            L.check(L.lib().oct_unpack_wgrad_batch(len(self._unpack_jobs), arr, _stream()), "oct_unpack_wgrad_batch")
            self._unpack_jobs = []

    def _arena_end(self, dev):
        self._flush_unpack()
        if self._arena_short or self._arena is None:   # first step (or a larger batch): size it for the next one
            self._arena = torch.empty(self._arena_off, dtype=torch.float32, device=dev)
        self._arena_on = False

    def _unpack(self, mode, dwp, grad, cout, cin, accumulate):
        job = L.UnpackJob(mode, cout, cin, int(accumulate), getattr(dwp, "_oct_nparts", 1), 0, dwp.data_ptr(), grad.data_ptr())
        if self._arena_on:   # inside backward(): all gradients are unpacked by one launch at the end
            self._unpack_jobs.append((job, dwp, grad))
            return
        arr = (L.UnpackJob * 1)(job)
        L.check(L.lib().oct_unpack_wgrad_batch(1, arr, _stream()), "oct_unpack_wgrad_batch")

    # ---- forward --------------------------------------------------------------------------------
    def _conv_bn(self, P, keys: ConvKeys, src: Src, cout, n, h, w, train: bool) -> ConvRec:
        wkey, nk = keys.w, keys.bn
        wt = P[wkey]
        dev = wt.device
        cbias = L.ptr(P[keys.b]) if keys.b else None
        wp = self._pack(wkey, wt, L.PACK_CONV_FPROP, cout, src.channels)
        y = self._act(n, h, w, cout, dev)
        scale = torch.empty(cout, dtype=torch.float32, device=dev)
        shift = torch.empty_like(scale)
        if train:
            nblk = self._stat_blocks(cout, n, h, w, src)
            partials = torch.empty((nblk, 2, cout), dtype=torch.float32, device=dev)
            self._conv(src, wp, cout, 9, n, h, w, y, stats=partials)
            mean = torch.empty_like(scale)
            invstd = torch.empty_like(scale)
            L.check(L.lib().oct_bn_finalize(
                partials.data_ptr(), nblk, cout, float(n * h * w), P[nk + ".weight"].data_ptr(),
                P[nk + ".bias"].data_ptr(), BN_EPS, BN_MOMENTUM, P[nk + ".running_mean"].data_ptr(),
                P[nk + ".running_var"].data_ptr(), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                shift.data_ptr(), cbias, _stream()), "oct_bn_finalize")
            self._nbt.append(P[nk + ".num_batches_tracked"])   # bumped by ONE multi-tensor launch at the end of forward
            bn = BNState(scale, shift, mean, invstd)
        else:
            self._conv(src, wp, cout, 9, n, h, w, y)
            L.check(L.lib().oct_bn_eval_coeffs(
                cout, P[nk + ".weight"].data_ptr(), P[nk + ".bias"].data_ptr(),
                P[nk + ".running_mean"].data_ptr(), P[nk + ".running_var"].data_ptr(), BN_EPS,
                scale.data_ptr(), shift.data_ptr(), cbias, _stream()), "oct_bn_eval_coeffs")
            if self._frozen_bwd:
                # backward through frozen statistics (fine-tuning with model.eval()): xhat = (y + conv_bias - running_mean)
                # * rsqrt(running_var + eps) for d(gamma); the data gradient is dz * gamma * invstd, no statistics terms
                invstd = torch.rsqrt(P[nk + ".running_var"] + BN_EPS)
                mean = P[nk + ".running_mean"] - P[keys.b] if keys.b else P[nk + ".running_mean"]
                bn = BNState(scale, shift, mean, invstd, frozen=True)
            else:
                bn = BNState(scale, shift)
        return ConvRec(wkey, nk + ".weight", nk + ".bias", keys.b, src, y, bn, cout, n, h, w)

    def _block(self, P, blk: BlockSpec, src, n, h, w, train, ctx):
        r1 = self._conv_bn(P, blk.c1, src, blk.cout, n, h, w, train)
        r2 = self._conv_bn(P, blk.c2, Src(r1.y, blk.cout, r1.bn), blk.cout, n, h, w, train)
        ctx.convs[blk.name] = [r1, r2]
        return r2

    def forward(self, P: dict, x: torch.Tensor, train: bool, target: torch.Tensor | None = None,
                loss_cfg=(1.0, 0.0, 1e-7), want_probs=True, want_argmax=False, want_logits=False, defer_loss=False,
                frozen_bwd=False):
        """P: name -> fp32 device tensor with the reference's state_dict keys.
        Returns (ctx, probs|None, argmax|None, logits|None).
        defer_loss: the caller will run backward() right away and wants no output but the loss; when the
        loss has no Dice term and the fused head backward applies (32 head features, <= 8 classes) the
        forward head pass is skipped and backward() computes the cross-entropy while it is there."""
        lib = L.lib()
        if x.dim() != 4 or x.shape[1] != self.cin:
            raise RuntimeError(f"expected input (B,{self.cin},H,W), got {tuple(x.shape)}")
        n, _, h, w = x.shape
        sp = self.spec
        if h % sp.divisor or w % sp.divisor:
            # same failure the reference hits at torch.cat (YNet_2022.py:557, BioNet_2020.py:64)
            raise RuntimeError(
                f"Sizes of tensors must match except in dimension 1. Input {h}x{w} is not divisible by {sp.divisor} "
                f"({len(sp.enc) - 1} 2x2 poolings followed by as many 2x up-samplings)")
        dev = x.device
        if dev.type != "cuda":
            raise L.OctError("the HIP path needs a device tensor (there is no CPU fallback)")
        xf = x.detach().to(torch.float32).contiguous()
        if train or frozen_bwd:
            self._prepack(P)
        self._frozen_bwd = bool(frozen_bwd) and not train
        ctx = Ctx(n=n, h=h, w=w, loss_cfg=tuple(loss_cfg))
        self._nbt = []
        xt = self._act(n, h, w, self.cin, dev)
        L.check(lib.oct_nchw_to_nhwc(self.dt, xf.data_ptr(), xt.data_ptr(), n, self.cin, h, w, _stream()),
                "oct_nchw_to_nhwc")
        src = Src(xt, self.cin)
        hh, ww = h, w
        skips = []
        prev = None
        for li, blk in enumerate(sp.enc):
            prev = self._block(P, blk, src, n, hh, ww, train, ctx)
            if li == len(sp.enc) - 1:
                break  # bottleneck: no pooling
            skips.append(prev)
            pooled = self._act(n, hh // 2, ww // 2, blk.cout, dev)
            L.check(lib.oct_bn_relu_pool_fwd(self.dt, prev.y.data_ptr(), prev.bn.scale.data_ptr(),
                                             prev.bn.shift.data_ptr(), pooled.data_ptr(), n, hh, ww, blk.cout,
                                             _stream()), "oct_bn_relu_pool_fwd")
            hh //= 2
            ww //= 2
            src = Src(pooled, blk.cout)
        for di, ((wkey, bkey, cout_d), blk) in enumerate(zip(sp.ups, sp.dec)):
            cin_d = prev.cout
            wp = self._pack(wkey, P[wkey], L.PACK_DECONV_FPROP, cout_d, cin_d)
            u = self._act(n, hh * 2, ww * 2, cout_d, dev)
            self._conv(Src(prev.y, cin_d, prev.bn), wp, 4 * cout_d, 1, n, hh, ww, u, out_mode=L.OUT_D2S, bias=P[bkey])
            ctx.ups[di] = (prev, u)
            hh *= 2
            ww *= 2
            sk = skips[len(skips) - 1 - di]
            src = (Src(u, cout_d, None, sk.y, sk.cout, sk.bn) if sp.dec_first
                   else Src(sk.y, sk.cout, sk.bn, u, cout_d, None))
            prev = self._block(P, blk, src, n, hh, ww, train, ctx)
        ctx.head_in = prev
        if self._nbt:       # BatchNorm step counters: one multi-tensor launch instead of one per layer
            torch._foreach_add_(self._nbt, 1)
            self._nbt = []
        self._frozen_bwd = False
        hd = L.HeadDesc(self.dt, n, h, w, self.f, self.ncls)
        probs = torch.empty((n, self.ncls, h, w), dtype=torch.float32, device=dev) if want_probs else None
        amax = torch.empty((n, h, w), dtype=torch.int64, device=dev) if want_argmax else None
        logits = torch.empty((n, self.ncls, h, w), dtype=torch.float32, device=dev) if want_logits else None
        partials = None
        if target is not None:
            if target.shape != (n, h, w):
                raise RuntimeError(f"target must be (B,H,W)={n, h, w}, got {tuple(target.shape)}")
            target = target.to(device=dev, dtype=torch.int64).contiguous()
            nb = lib.oct_head_blocks(C.byref(hd))
            partials = torch.empty((nb, L.HEAD_LOSS_SLOTS), dtype=torch.float64, device=dev)
            if (defer_loss and loss_cfg[1] == 0.0 and self.f == 32 and self.ncls <= 8
                    and not (want_probs or want_argmax or want_logits)):
                ctx.target = target
                ctx.loss = torch.empty(3, dtype=torch.float32, device=dev)
                ctx.dice_coef = torch.zeros(2 * L.MAX_CLASSES, dtype=torch.float32, device=dev)
                ctx.loss_partials = partials      # filled and finalised by backward()
                return ctx, None, None, None
        L.check(lib.oct_head_forward(C.byref(hd), prev.y.data_ptr(), prev.bn.scale.data_ptr(),
                                     prev.bn.shift.data_ptr(), P[sp.head_w].data_ptr(),
                                     P[sp.head_b].data_ptr(), L.ptr(target), L.ptr(probs), L.ptr(amax),
                                     L.ptr(logits), L.ptr(partials), _stream()), "oct_head_forward")
        if target is not None:
            w_ce, w_dice, eps = loss_cfg
            ctx.target = target
            ctx.loss = torch.empty(3, dtype=torch.float32, device=dev)
            ctx.dice_coef = torch.zeros(2 * L.MAX_CLASSES, dtype=torch.float32, device=dev)
            L.check(lib.oct_head_loss_finalize(C.byref(hd), partials.data_ptr(), partials.shape[0], w_ce, w_dice,
                                               eps, ctx.loss.data_ptr(), ctx.dice_coef.data_ptr(), _stream()),
                    "oct_head_loss_finalize")
        return ctx, probs, amax, logits

    # ---- backward -------------------------------------------------------------------------------
    def _bn_backward(self, rec: ConvRec, da, dpool, G, accumulate, partials=None, defer_apply=False):
        """da (and/or pooled gradient) wrt relu(bn(y)) -> dy in place; BN parameter grads.
        partials: reduction already done by the producer of da (fused head backward)."""
        lib = L.lib()
        n, h, w, c = rec.n, rec.h, rec.w, rec.cout
        dev = rec.y.device
        pooled = dpool is not None
        g = da if da is not None else self._act(n, h, w, c, dev)
        # pooled layers: reduce only as well when the library can re-derive the routed, masked gradient in the apply pass
        pool_fused = pooled and partials is None and self.debug is None and bool(lib.oct_bn_bwd_apply_pool_ok(self.dt, n, h, w, c))
        if partials is not None:
            nblk = partials.shape[0]
        else:
            nblk = lib.oct_dact_bn_reduce_blocks(n, h, w, c, 1 if pooled else 0)
            partials = torch.empty((nblk, 2, c), dtype=torch.float32, device=dev)
            # non-pooled layers: reduce only (no masked copy is written); the apply pass re-derives the mask
            L.check(lib.oct_dact_bn_reduce(self.dt, L.ptr(da), L.ptr(dpool), rec.y.data_ptr(), rec.bn.scale.data_ptr(),
                                           rec.bn.shift.data_ptr(), rec.bn.mean.data_ptr(), rec.bn.invstd.data_ptr(),
                                           g.data_ptr() if (pooled and not pool_fused) else None, partials.data_ptr(),
                                           n, h, w, c, _stream()),
                    "oct_dact_bn_reduce")
        coef = torch.empty((3, c), dtype=torch.float32, device=dev)
        L.check(lib.oct_bn_bwd_finalize(partials.data_ptr(), nblk, c, float(n * h * w), self._P[rec.gkey].data_ptr(),
                                        rec.bn.mean.data_ptr(), rec.bn.invstd.data_ptr(), G[rec.gkey].data_ptr(),
                                        G[rec.bkey].data_ptr(), coef.data_ptr(), int(accumulate), _stream()),
                "oct_bn_bwd_finalize")
        if self.debug is not None:
            self.debug["g:" + rec.wkey] = g.float().clone()
        if rec.bn.frozen:
            # running statistics do not depend on the batch: dy = k0 * dz (k0 = gamma * invstd), and a convolution bias in
            # front of the BatchNorm gets d(bias) = sum dy = k0 * sum dz
            coef[1:].zero_()
            if rec.cbkey:
                gb = coef[0] * partials[:, 0, :].sum(0)
                G[rec.cbkey].add_(gb) if accumulate else G[rec.cbkey].copy_(gb)
        if defer_apply and not pooled:
            return g, coef   # the consumer (first-layer wgrad) applies dy = k0*mask*dA + k1*y + k2 on load
        if pool_fused:
            L.check(lib.oct_bn_bwd_apply_pool(self.dt, L.ptr(da), dpool.data_ptr(), rec.y.data_ptr(), rec.bn.scale.data_ptr(),
                                              rec.bn.shift.data_ptr(), coef.data_ptr(), g.data_ptr(), n, h, w, c, _stream()),
                    "oct_bn_bwd_apply_pool")
            return g
        L.check(lib.oct_bn_bwd_apply(self.dt, g.data_ptr(), rec.y.data_ptr(), coef.data_ptr(),
                                     None if pooled else rec.bn.scale.data_ptr(),
                                     None if pooled else rec.bn.shift.data_ptr(), n * h * w, c, _stream()),
                "oct_bn_bwd_apply")
        if self.debug is not None:
            self.debug["dy:" + rec.wkey] = g.float().clone()
            self.debug["y:" + rec.wkey] = rec.y.float().clone()
            self.debug["coef:" + rec.wkey] = coef.clone()
        return g

    def _conv_backward(self, rec: ConvRec, dy, G, accumulate, need_dx=True):
        """dW of the conv (into G) and, if needed, the gradient(s) wrt its (virtually concatenated) input."""
        n, h, w = rec.n, rec.h, rec.w
        src = rec.src
        cin = src.channels
        dwp = self._wgrad(src, dy, rec.cout, 9, n, h, w)
        self._unpack(L.PACK_CONV_FPROP, dwp, G[rec.wkey], rec.cout, cin, accumulate)
        if not need_dx:
            return None, None
        wp = self._pack(rec.wkey, self._P[rec.wkey], L.PACK_CONV_DGRAD, rec.cout, cin)
        d0 = self._act(n, h, w, src.c0, dy.device)
        d1 = self._act(n, h, w, src.c1, dy.device) if src.c1 else None
        self._conv(Src(dy, rec.cout), wp, cin, 9, n, h, w, d0, y1=d1, split=src.c0 if src.c1 else 0)
        if self.debug is not None:
            # everything a checker needs to redo THIS layer from the tensors the kernels actually saw (teacher forcing)
            def bnc(b):
                return None if b is None else (b.scale.clone(), b.shift.clone())
            self.debug["layer:" + rec.wkey] = dict(
                x0=src.x0.float().clone(), bn0=bnc(src.bn0), x1=None if src.x1 is None else src.x1.float().clone(),
                bn1=bnc(src.bn1), dy=dy.float().clone(), d0=d0.float().clone(), d1=None if d1 is None else d1.float().clone(),
                y=rec.y.float().clone())
        return d0, d1

    def _block_backward(self, name, da, dpool, G, accumulate, need_dx=True, partials=None):
        r1, r2 = self._ctx.convs[name]
        for r in (r1, r2):
            if r.cbkey and not accumulate and not r.bn.frozen:
                G[r.cbkey].zero_()   # a bias in front of a train-mode BatchNorm cancels in (y - mean): zero gradient
        dy2 = self._bn_backward(r2, da, dpool, G, accumulate, partials=partials)
        da1, _ = self._conv_backward(r2, dy2, G, accumulate)
        first_fused = False
        if not need_dx and r1.src.c1 == 0 and r1.src.bn0 is None:
            # the library decides (shape, dtype, OCT_DISABLE_V2): a host-side copy of that rule would drift
            qd = L.WgradDesc(self.dt, r1.n, r1.h, r1.w, r1.src.c0, 0, r1.cout, 9, L.XF_NONE, L.XF_NONE, L.IN_PLAIN)
            first_fused = bool(L.lib().oct_conv_wgrad_fused_apply_ok(C.byref(qd)))
        if first_fused:
            # first layer: no data gradient, so dY1 has a single consumer -- the weight-gradient kernel
            # applies the BN backward itself and the dY1 tensor is never written
            da1, coef = self._bn_backward(r1, da1, None, G, accumulate, defer_apply=True)
            dwp = self._wgrad(r1.src, da1, r1.cout, 9, r1.n, r1.h, r1.w,
                              fused_apply=(r1.y, coef, r1.bn.scale, r1.bn.shift))
            self._unpack(L.PACK_CONV_FPROP, dwp, G[r1.wkey], r1.cout, 1, accumulate)
            return None, None
        dy1 = self._bn_backward(r1, da1, None, G, accumulate)
        return self._conv_backward(r1, dy1, G, accumulate, need_dx=need_dx)

    def backward_stages(self):
        """[(stage name, [parameter keys])] in the order backward() FINISHES their gradients: last decoder
        block (+ head, + its up-convolution) first, first encoder block last.  This is the reverse of the
        reference's construction order (YNet_2022.py:511-546, BioNet_2020.py:24-43), which is what lets a
        data-parallel caller ship contiguous tail slices of a flat gradient buffer while backward continues."""
        sp = self.spec

        def blk_keys(b: BlockSpec):
            ks = []
            for c in (b.c1, b.c2):
                ks += [c.w, c.bn + ".weight", c.bn + ".bias"] + ([c.b] if c.b else [])
            return ks
        nd = len(sp.dec)
        stages = []
        for di in range(nd - 1, -1, -1):
            ks = ([sp.head_w, sp.head_b] if di == nd - 1 else []) + blk_keys(sp.dec[di]) + list(sp.ups[di][:2])
            stages.append((f"dec{di}", ks))
        for li in range(len(sp.enc) - 1, -1, -1):
            stages.append((f"enc{li}", blk_keys(sp.enc[li])))
        return stages

    def _stage_done(self, hook, idx):
        if hook is not None and idx in hook.flush_stages:
            self._flush_unpack()        # the stage's weight gradients must be in torch layout before they leave
            hook.stage_done(idx)

    def backward(self, P: dict, ctx: Ctx, G: dict, dprobs: torch.Tensor | None = None, accumulate=False,
                 dlogits: torch.Tensor | None = None, stage_hook=None):
        """Fills G (name -> fp32 grad tensor, torch layout) for every parameter.
        dprobs: gradient wrt the softmax output; dlogits: gradient wrt the logits (BioNet-style nets);
        neither: gradient of the fused loss recorded by forward(target=...).
        stage_hook (data parallel): object with `flush_stages` (indices into backward_stages()) and
        `stage_done(idx)`, called on the launch stream's host thread as soon as every gradient of those
        stages has been enqueued."""
        lib = L.lib()
        sp = self.spec
        self._P, self._ctx = P, ctx
        n, h, w, f, ncls = ctx.n, ctx.h, ctx.w, self.f, self.ncls
        rec = ctx.head_in
        dev = rec.y.device
        self._arena_begin(dev)
        hd = L.HeadDesc(self.dt, n, h, w, f, ncls)
        if dprobs is not None:
            dprobs = dprobs.to(torch.float32).contiguous()
            tgt, dc, w_ce = None, None, 0.0
        elif dlogits is None:
            if ctx.target is None:
                raise RuntimeError("backward without an output gradient needs forward(target=...)")
            tgt, dc, w_ce = ctx.target, ctx.dice_coef, ctx.loss_cfg[0]
        hsrc = Src(rec.y, f, rec.bn)
        head_partials = None
        hw, hb = P[sp.head_w], P[sp.head_b]
        bgrad, wgrad_t = G[sp.head_b], G[sp.head_w]
        fused_dw = False
        dl = None
        if f == 32 and dlogits is None:
            # fused: dlogits + dA = W^T dlogits + bias / weight gradients + BN-backward partial sums in one pass over y
            fused_dw = ncls <= 8
            if not accumulate:
                bgrad.zero_()
                if fused_dw:
                    wgrad_t.zero_()
            if not fused_dw:
                dl = self._act(n, h, w, ncls, dev)
            da = self._act(n, h, w, f, dev)
            nb = lib.oct_head_blocks(C.byref(hd))
            head_partials = torch.empty((nb, 2, f), dtype=torch.float32, device=dev)
            L.check(lib.oct_head_backward_fused(
                C.byref(hd), rec.y.data_ptr(), rec.bn.scale.data_ptr(), rec.bn.shift.data_ptr(),
                rec.bn.mean.data_ptr(), rec.bn.invstd.data_ptr(), hw.data_ptr(),
                hb.data_ptr(), L.ptr(tgt), L.ptr(dc), w_ce, L.ptr(dprobs), L.ptr(dl), da.data_ptr(),
                head_partials.data_ptr(), bgrad.data_ptr(), wgrad_t.data_ptr() if fused_dw else None,
                L.ptr(ctx.loss_partials) if dprobs is None else None, _stream()),
                "oct_head_backward_fused")
            if ctx.loss_partials is not None and dprobs is None:
                w_ce_, w_dice_, eps_ = ctx.loss_cfg
                L.check(lib.oct_head_loss_finalize(C.byref(hd), ctx.loss_partials.data_ptr(), ctx.loss_partials.shape[0],
                                                   w_ce_, w_dice_, eps_, ctx.loss.data_ptr(), ctx.dice_coef.data_ptr(),
                                                   _stream()), "oct_head_loss_finalize")
        else:
            dl = self._act(n, h, w, ncls, dev)
            if dlogits is not None:
                dlf = dlogits.to(torch.float32).contiguous()
                L.check(lib.oct_nchw_to_nhwc(self.dt, dlf.data_ptr(), dl.data_ptr(), n, ncls, h, w, _stream()),
                        "oct_nchw_to_nhwc")
            else:
                L.check(lib.oct_head_dlogits(C.byref(hd), rec.y.data_ptr(), rec.bn.scale.data_ptr(),
                                             rec.bn.shift.data_ptr(), hw.data_ptr(), hb.data_ptr(), L.ptr(tgt),
                                             L.ptr(dc), w_ce, L.ptr(dprobs), dl.data_ptr(), _stream()),
                        "oct_head_dlogits")
            L.check(lib.oct_channel_sum(self.dt, dl.data_ptr(), bgrad.data_ptr(), n * h * w, ncls, int(accumulate),
                                        _stream()), "oct_channel_sum")
            wp = self._pack(sp.head_w, hw, L.PACK_1X1_DGRAD, ncls, f)
            da = self._act(n, h, w, f, dev)
            self._conv(Src(dl, ncls), wp, f, 1, n, h, w, da)
        if not fused_dw:
            # head weight gradient through the generic 1x1 wgrad
            dwp = self._wgrad(hsrc, dl, ncls, 1, n, h, w)
            self._unpack(L.PACK_1X1_FPROP, dwp, wgrad_t, ncls, f, accumulate)
        nd = len(sp.dec)
        dskip = [None] * nd
        for di in range(nd - 1, -1, -1):   # last decoder block first
            d0, d1 = self._block_backward(sp.dec[di].name, da, None, G, accumulate,
                                          partials=head_partials if di == nd - 1 else None)
            du, dskip[di] = (d0, d1) if sp.dec_first else (d1, d0)
            wkey, bkey, cout_d = sp.ups[di]
            prev, u = ctx.ups[di]
            cin_d = prev.cout
            hl, wl = prev.h, prev.w
            bgrad = G[bkey]
            if not accumulate:
                bgrad.zero_()
            dwp = self._wgrad(Src(prev.y, cin_d, prev.bn), du, 4 * cout_d, 1, n, hl, wl, dy_mode=L.IN_S2D,
                              dbias=bgrad)
            self._unpack(L.PACK_DECONV_FPROP, dwp, G[wkey], cout_d, cin_d, accumulate)
            wp = self._pack(wkey, P[wkey], L.PACK_DECONV_DGRAD, cout_d, cin_d)
            da = self._act(n, hl, wl, cin_d, dev)
            self._conv(Src(du, cout_d), wp, cin_d, 1, n, hl, wl, da, in_mode=L.IN_S2D)
            if self.debug is not None:   # teacher forcing of the transposed convolution (tests): what its three kernels read and wrote
                self.debug["up:" + wkey] = dict(x=prev.y.float().clone(), bn=(prev.bn.scale.clone(), prev.bn.shift.clone()),
                                                u=u.float().clone(), du=du.float().clone(), da=da.float().clone(), bkey=bkey)
            self._stage_done(stage_hook, nd - 1 - di)
        dpool, _ = self._block_backward(sp.enc[-1].name, da, None, G, accumulate)
        self._stage_done(stage_hook, nd)
        for li in range(len(sp.enc) - 2, -1, -1):
            # the skip of enc[li] was consumed by decode step nd-1-li
            dpool, _ = self._block_backward(sp.enc[li].name, dskip[nd - 1 - li], dpool, G, accumulate,
                                            need_dx=(li != 0))
            if li:
                self._stage_done(stage_hook, nd + len(sp.enc) - 1 - li)
        self._arena_end(dev)
        self._stage_done(stage_hook, nd + len(sp.enc) - 1)   # after the last unpack launch
        self._P = self._ctx = None
        return G
