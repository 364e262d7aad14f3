"""Single-node data parallelism: one process per GPU, minibatch sharded across ranks, the gradient
exchanged as a few contiguous BUCKETS of one flat buffer on RCCL over xGMI (torch.distributed
backend "nccl" is RCCL on ROCm), each bucket launched on a side stream as soon as backward has
finished it.

The reference has no distributed code at all; semantics are those of stock DDP:
per-rank BatchNorm statistics (no SyncBN), gradients averaged over ranks, parameters AND buffers
broadcast from rank 0 at construction, BN running buffers rank-local afterwards.

Why buckets are slices: backward finishes gradients in the reverse of the reference's construction
order (head, decoder1, upconv1, ... bottleneck, encoder4 ... encoder1 -- `UNetEngine.backward_stages`),
and `optim.FlatParams` lays parameters out in construction order, so "everything backward has
finished so far" is always a contiguous TAIL of the flat gradient buffer.  The payload is tiny
(7.76 M fp32 = 31 MB for UNet(1,8)) against >= 20 ms of compute: the point of bucketing is not
bandwidth but hiding all of it except the last, smallest bucket (encoder1-3: 1.1 MB) under the
full-resolution encoder backward.  xGMI is point-to-point, a ring all-reduce is latency-bound at
these sizes, so there are few buckets (4 by default), not many.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract).  Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        # bounded rendezvous: a rank that never arrives (died before init) must fail the others, not hang them
        import datetime
        tmo = float(os.environ.get("OCT_RDZV_TIMEOUT_S", "300"))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=tmo))
    return rank, world, local


def barrier(local: int | None = None):
    """dist.barrier() that names this rank's device (RCCL otherwise guesses it from the rank)."""
    if not dist.is_initialized():
        return
    if dist.get_backend() == "nccl" and local is not None:
        dist.barrier(device_ids=[local])
    else:
        dist.barrier()


def shard_batch(global_batch: int, rank: int, world: int):
    """Contiguous per-rank slice [lo, hi) of a global minibatch (ragged tails go to the first ranks)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def plan_buckets(stage_spans, total: int, cap_bytes: int = 3 << 20):
    """stage_spans: [(lo, hi)] of every backward stage inside the flat gradient buffer, in the order
    backward finishes them.  Returns [(last_stage_index, lo, hi)]: consecutive stages are merged until a
    bucket holds >= cap_bytes; the final bucket takes whatever is left and ends at offset 0.
    Falls back to ONE bucket (the whole buffer after the last stage) when the finished stages do not form
    a contiguous tail of the buffer -- correctness never depends on the layout assumption."""
    n = len(stage_spans)
    edge = total
    tails = []
    for lo, hi in stage_spans:
        if hi != edge or lo > hi:
            return [(n - 1, 0, total)]
        edge = lo
        tails.append(lo)
    if edge != 0:
        return [(n - 1, 0, total)]
    buckets, hi = [], total
    for i, lo in enumerate(tails):
        if ((hi - lo) * 4 >= cap_bytes or i == n - 1) and hi > lo:
            buckets.append((i, lo, hi))
            hi = lo
    return buckets


class GradAllReducer:
    """Sums one flat gradient buffer over all ranks, bucket by bucket, on a side stream.

    As `stage_hook` of `UNetEngine.backward` it launches bucket k the moment backward has enqueued the
    last gradient kernel of the bucket's final stage (the side stream waits for an event recorded on the
    compute stream at that point, so the exchange runs under the remaining backward kernels).  Without a
    plan it is the one-bucket reducer: `start()` after backward, `finish()` before the optimizer."""

    def __init__(self, flat_grad: torch.Tensor, world: int | None = None, buckets=None, always_communicate=False):
        self.flat = flat_grad
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        # highest priority: every compute kernel of the step is a persistent one-workgroup-per-CU launch, and a collective
        # queued at default priority would have to wait for CUs behind whole conv kernels instead of slipping in between
        # their workgroups
        self.stream = torch.cuda.Stream(priority=-1) if flat_grad.is_cuda else None
        self.buckets = list(buckets) if buckets else [(0, 0, flat_grad.numel())]
        self.flush_stages = {b[0] for b in self.buckets}
        # a 1-rank group has nothing to exchange; `always_communicate` still issues the collectives (tests)
        self.active = self.world > 1 or (always_communicate and dist.is_initialized())
        if not self.active:
            self.flush_stages = set()
        self._by_stage = {b[0]: k for k, b in enumerate(self.buckets)}
        self._launched = [False] * len(self.buckets)
        self.launch_log = []        # (bucket index, lo, hi) in launch order, for tests / the bench line

    def begin_step(self):
        """Forget the launches of a step that never reached finish() (forward_backward raised after some stage hooks had
        fired): without this the next step would skip those buckets and the ranks would fall out of step."""
        self._launched = [False] * len(self.buckets)

    # ---- stage hook protocol (UNetEngine.backward) ----------------------------------------------------
    def stage_done(self, idx: int):
        k = self._by_stage.get(idx)
        if k is not None:
            self._launch(k)

    def _launch(self, k: int):
        if not self.active or self._launched[k]:
            return
        _, lo, hi = self.buckets[k]
        self._launched[k] = True
        self.launch_log.append((k, lo, hi))
        piece = self.flat[lo:hi]
        if self.stream is None:
            dist.all_reduce(piece, op=dist.ReduceOp.SUM)
            return
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            dist.all_reduce(piece, op=dist.ReduceOp.SUM)

    # ---- one-shot protocol ------------------------------------------------------------------------------
    def start(self):
        """Enqueue every bucket not launched yet, after everything already queued on the compute stream."""
        for k in range(len(self.buckets)):
            self._launch(k)

    def finish(self) -> float:
        """Make the compute stream wait for the exchange; returns the scale the optimizer applies."""
        self.start()
        if self.active and self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        self._launched = [False] * len(self.buckets)
        return 1.0 / self.world


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0):
    """Same initial weights on every rank (what DDP does at construction)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat_params, src=src)
        from . import _lib as L
        L.param_generation[0] += 1


def broadcast_buffers(module: torch.nn.Module, src: int = 0):
    """BN running statistics and step counters from rank 0, once (stock DDP does the same at
    construction); they stay rank-local afterwards.  Coalesced: one broadcast per dtype."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    groups = {}
    for b in module.buffers():
        groups.setdefault(b.dtype, []).append(b)
    for bufs in groups.values():
        flat = torch.cat([b.reshape(-1) for b in bufs])
        dist.broadcast(flat, src=src)
        off = 0
        for b in bufs:
            n = b.numel()
            b.copy_(flat[off:off + n].view(b.shape))
            off += n


def bucket_plan_for(model, layout, cap_bytes: int = 3 << 20):
    """Buckets of `layout` (optim.FlatParams of model.named_parameters()) along the model's backward stages."""
    stages = model._engine.backward_stages()
    spans = []
    for _, keys in stages:
        los, his = zip(*(layout.span(k) for k in keys)) if keys else ((), ())
        spans.append((min(los), max(his)) if keys else (0, 0))
    return plan_buckets(spans, layout.total, cap_bytes)


class DataParallelTrainer:
    """model.forward_backward on the local shard (gradient buckets leave while backward runs) -> fused SGD.

    use_graph: the ~150 kernel launches of forward + loss + backward are recorded once into a HIP
    graph (torch.cuda.CUDAGraph, capture on the stream the C ABI launches on) and replayed per
    step, which removes the host launch gaps between the short kernels of the deep levels.  The
    exchange (then ONE bucket after the replay) and the one-kernel optimizer step stay outside the
    graph.  Capture happens on the first step() after `graph_warmup` eager steps (the optimizer's
    first-step flag and the weight re-packing that follows every update must already be in their
    steady state)."""

    def __init__(self, model, lr=0.01, momentum=0.9, weight_decay=0.0, w_ce=1.0, w_dice=0.0, use_graph=False,
                 graph_warmup=2, bucket_cap_bytes=3 << 20, always_communicate=False):
        from .optim import FusedSGD
        self.model = model
        self.opt = FusedSGD(list(model.named_parameters()), lr=lr, momentum=momentum, weight_decay=weight_decay)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        broadcast_parameters(self.opt.flat_p)
        broadcast_buffers(model)
        buckets = None if use_graph else bucket_plan_for(model, self.opt.layout, bucket_cap_bytes)
        self.reducer = GradAllReducer(self.opt.flat_g, self.world, buckets, always_communicate=always_communicate)
        self.w_ce, self.w_dice = w_ce, w_dice
        self.use_graph, self.graph_warmup = use_graph, graph_warmup
        self.graph = None
        self.graph_error = None
        self._eager_steps = 0
        self._sx = self._st = self._loss = None

    def _capture(self, x, target):
        from . import _lib as L
        self._sx, self._st = x.clone(), target.clone()
        L.param_generation[0] += 1          # the recorded forward must contain the weight packing kernels
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                self._loss = self.model.forward_backward(self._sx, self._st, self.w_ce, self.w_dice)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = g

    def step(self, x, target):
        if self.use_graph and self.graph is None and self.graph_error is None and self._eager_steps >= self.graph_warmup:
            try:
                self._capture(x, target)
            except Exception as e:  # capture is an optimisation: report and keep training eagerly
                self.graph_error = f"{type(e).__name__}: {e}"
                self.graph = None
        if self.graph is not None:
            if x is not self._sx and x.data_ptr() != self._sx.data_ptr():
                self._sx.copy_(x, non_blocking=True)
            if target is not self._st and target.data_ptr() != self._st.data_ptr():
                self._st.copy_(target, non_blocking=True)
            self.graph.replay()
            loss = self._loss
        else:
            hook = self.reducer if (self.reducer.active and not self.use_graph) else None
            self.reducer.begin_step()
            loss = self.model.forward_backward(x, target, self.w_ce, self.w_dice, stage_hook=hook)
            self._eager_steps += 1
        scale = self.reducer.finish()       # launches whatever backward did not, then joins the streams
        self.opt.step(grad_scale=scale)
        return loss

    def global_metric_counts(self, sums):
        """Sum per-rank confusion counts [tp,t,p,tn,fp,fn] over ranks (48-byte all-reduce)."""
        t = torch.tensor(sums, dtype=torch.int64, device=self.opt.flat_p.device)
        if self.world > 1:
            dist.all_reduce(t)
        return t.tolist()
