"""Single-node data parallelism: one process per GPU, minibatch sharded across ranks, ONE gradient
all-reduce per step on RCCL over xGMI (torch.distributed backend "nccl" is RCCL on ROCm).

The reference has no distributed code at all; semantics are those of stock DDP:
per-rank BatchNorm statistics (no SyncBN), gradients averaged over ranks, BN running buffers stay
rank-local.  The payload is tiny (7.76 M fp32 = 31 MB for UNet(1,8)) against >= 20 ms of compute,
so the exchange is latency-bound: all gradients live in one flat buffer (FusedSGD) and go out as a
single all-reduce on a side stream; the optimizer kernel waits on its event.  On CPU (gloo) the
same code path is exercised by tests/test_ddp_cpu.py with a stand-in step function.
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
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(global_batch: int, rank: int, world: int):
    """Contiguous per-rank slice [lo, hi) of a global minibatch (ragged tails go to the first ranks)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradAllReducer:
    """Averages one flat gradient buffer over all ranks, overlapped on a side stream."""

    def __init__(self, flat_grad: torch.Tensor, world: int | None = None):
        self.flat = flat_grad
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.stream = torch.cuda.Stream() if flat_grad.is_cuda else None

    def start(self):
        """Enqueue the all-reduce after everything already queued on the compute stream."""
        if self.world == 1:
            return
        if self.stream is None:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            return
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)

    def finish(self) -> float:
        """Make the compute stream wait for the exchange; returns the scale the optimizer applies."""
        if self.world > 1 and self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        return 1.0 / self.world


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0):
    """Same initial weights on every rank (what DDP does at construction)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat_params, src=src)
        from . import _lib as L
        L.param_generation[0] += 1


class DataParallelTrainer:
    """model.forward_backward on the local shard -> gradient all-reduce -> fused SGD.

    use_graph: the ~150 kernel launches of forward + loss + backward are recorded once into a HIP
    graph (torch.cuda.CUDAGraph, capture on the stream the C ABI launches on) and replayed per
    step, which removes the host launch gaps between the short kernels of the deep levels.  The
    exchange and the one-kernel optimizer step stay outside the graph.  Capture happens on the
    first step() after `graph_warmup` eager steps (the optimizer's first-step flag and the
    weight re-packing that follows every update must already be in their steady state)."""

    def __init__(self, model, lr=0.01, momentum=0.9, weight_decay=0.0, w_ce=1.0, w_dice=0.0, use_graph=False,
                 graph_warmup=2):
        from .optim import FusedSGD
        self.model = model
        self.opt = FusedSGD(model.parameters(), lr=lr, momentum=momentum, weight_decay=weight_decay)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        broadcast_parameters(self.opt.flat_p)
        self.reducer = GradAllReducer(self.opt.flat_g, self.world)
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
            loss = self.model.forward_backward(x, target, self.w_ce, self.w_dice)
            self._eager_steps += 1
        self.reducer.start()
        scale = self.reducer.finish()
        self.opt.step(grad_scale=scale)
        return loss

    def global_metric_counts(self, sums):
        """Sum per-rank confusion counts [tp,t,p,tn,fp,fn] over ranks (32-byte all-reduce)."""
        t = torch.tensor(sums, dtype=torch.int64, device=self.opt.flat_p.device)
        if self.world > 1:
            dist.all_reduce(t)
        return t.tolist()
