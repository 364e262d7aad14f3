"""CPU, world_size 2, gloo: the data-parallel plumbing (shard -> gradient all-reduce -> identical
update on every rank) with a stand-in step, since the HIP kernels need a GPU.  What is checked is
exactly what the N>1 bench path relies on: contiguous sharding, SUM all-reduce + 1/world scaling,
parameter broadcast, and that 2-rank training equals 1-rank training on the concatenated batch for
a model without batch statistics (BN statistics stay rank-local by design, like stock DDP)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from retinal_oct_image_segmentation_via_deep_learning_amd import ddp
    r, w, _ = ddp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)  # deliberately different initial weights per rank
    flat_p = torch.randn(1000)
    ddp.broadcast_parameters(flat_p)  # now rank 0's
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8, 1000, generator=g)  # the "global batch": per-sample gradient = x_i * <x_i, p>
    lo, hi = ddp.shard_batch(8, rank, world)
    flat_g = torch.zeros(1000)
    red = ddp.GradAllReducer(flat_g, world)
    losses = []
    for _ in range(3):
        xs = X[lo:hi]
        flat_g.copy_((xs * (xs @ flat_p)[:, None]).sum(0) / (hi - lo))  # mean over the local shard
        red.start()
        scale = red.finish()
        flat_p -= 0.01 * scale * flat_g
        losses.append(float(0.5 * ((X @ flat_p) ** 2).mean()))
    counts = torch.tensor([rank + 1, 10 * (rank + 1), 3, 4, 5, 6], dtype=torch.int64)
    dist.all_reduce(counts)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), p=flat_p.numpy(), losses=np.array(losses), counts=counts.numpy())
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["p"], r1["p"])  # identical parameters on every rank
    assert r0["counts"].tolist() == [3, 30, 6, 8, 10, 12]
    # single-process reference on the whole batch, starting from rank 0's weights
    torch.manual_seed(100)
    p = torch.randn(1000)
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8, 1000, generator=g)
    for _ in range(3):
        grad = (X * (X @ p)[:, None]).sum(0) / 8
        p -= 0.01 * grad
    np.testing.assert_allclose(r0["p"], p.numpy(), rtol=1e-5, atol=1e-6)
    assert r0["losses"][-1] < r0["losses"][0]


def test_shard_batch_is_contiguous_and_ragged_safe():
    from retinal_oct_image_segmentation_via_deep_learning_amd.ddp import shard_batch
    for gb, world in ((256, 8), (32, 1), (10, 4), (3, 8)):
        spans = [shard_batch(gb, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == gb
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
