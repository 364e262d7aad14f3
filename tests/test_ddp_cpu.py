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


# --------------------------------------------------------------------------------------------------------
# bucketed exchange on the REAL flat-parameter layout (optim.FlatParams of the UNet's named_parameters),
# driven the way UNetEngine.backward drives its stage hook, world_size 2, gloo
# --------------------------------------------------------------------------------------------------------
def _standin_grad(flat_p, shard_seed):
    """A gradient that depends on the parameters and on the rank's shard (so averaging matters)."""
    g = torch.Generator().manual_seed(shard_seed)
    return 0.5 * flat_p + torch.randn(flat_p.numel(), generator=g)


def _bucket_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from retinal_oct_image_segmentation_via_deep_learning_amd import UNet, ddp
    from retinal_oct_image_segmentation_via_deep_learning_amd.optim import FlatParams
    ddp.init_from_env("gloo")
    torch.manual_seed(100 + rank)
    model = UNet(1, 3, init_features=4)                      # parameter containers only: nothing runs on CPU
    with torch.no_grad():
        model.encoder1.enc1norm1.running_mean.fill_(float(rank + 5))   # rank 1 must end up with rank 0's 5.0
    layout = FlatParams(model.named_parameters())
    ddp.broadcast_parameters(layout.flat_p)
    ddp.broadcast_buffers(model)
    buckets = ddp.bucket_plan_for(model, layout, cap_bytes=4 << 10)    # small cap: several buckets on this tiny net
    red = ddp.GradAllReducer(layout.flat_g, world, buckets)
    nstages = len(model._engine.backward_stages())
    buf = torch.zeros_like(layout.flat_p)
    lr, mom = 0.05, 0.9
    for step in range(3):
        layout.flat_g.copy_(_standin_grad(layout.flat_p, 1000 * step + rank))
        for idx in range(nstages):                           # the hook protocol of UNetEngine.backward
            if idx in red.flush_stages:
                red.stage_done(idx)
        scale = red.finish()
        buf.mul_(mom).add_(layout.flat_g * scale)            # torch.optim.SGD(momentum) on the flat buffers
        layout.flat_p.sub_(lr * buf)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), p=layout.flat_p.numpy(),
             rm=model.encoder1.enc1norm1.running_mean.numpy(), nb=len(buckets),
             order=np.array([k for k, _, _ in red.launch_log]), w=model.conv.weight.detach().numpy())
    dist.destroy_process_group()


def test_bucketed_reducer_on_flat_unet_layout_two_ranks(tmp_path):
    world = 2
    mp.spawn(_bucket_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["p"], r1["p"])
    assert r1["rm"].tolist() == [5.0] * 4                    # buffers broadcast from rank 0
    nb = int(r0["nb"])
    assert nb >= 3
    assert r0["order"].tolist() == list(range(nb)) * 3       # every bucket once per step, in backward order
    # single process: mean of the two shard gradients
    sys.path.insert(0, ROOT)
    from retinal_oct_image_segmentation_via_deep_learning_amd import UNet
    from retinal_oct_image_segmentation_via_deep_learning_amd.optim import FlatParams
    torch.manual_seed(100)
    model = UNet(1, 3, init_features=4)
    layout = FlatParams(model.named_parameters())
    buf = torch.zeros_like(layout.flat_p)
    for step in range(3):
        g = 0.5 * (_standin_grad(layout.flat_p, 1000 * step) + _standin_grad(layout.flat_p, 1000 * step + 1))
        buf.mul_(0.9).add_(g)
        layout.flat_p.sub_(0.05 * buf)
    np.testing.assert_allclose(r0["p"], layout.flat_p.numpy(), rtol=1e-5, atol=1e-6)
    # parameters are views of the flat buffer: the module sees the update
    np.testing.assert_allclose(r0["w"], model.conv.weight.detach().numpy(), rtol=1e-5, atol=1e-6)


def test_bucket_plan_tiles_the_flat_buffer_in_backward_order():
    from retinal_oct_image_segmentation_via_deep_learning_amd import UNet, ddp
    from retinal_oct_image_segmentation_via_deep_learning_amd.optim import FlatParams
    from retinal_oct_image_segmentation_via_deep_learning_amd.unet import BioUNet
    for model in (UNet(1, 8, init_features=32), BioUNet(1, 2)):
        layout = FlatParams(model.named_parameters())
        stages = model._engine.backward_stages()
        keys = [k for _, ks in stages for k in ks]
        assert sorted(keys) == sorted(n for n, _ in model.named_parameters())   # every parameter in exactly one stage
        buckets = ddp.bucket_plan_for(model, layout)
        assert buckets[0][2] == layout.total and buckets[-1][1] == 0
        assert all(a[1] == b[2] for a, b in zip(buckets, buckets[1:]))         # contiguous, tail first
        assert [b[0] for b in buckets] == sorted(b[0] for b in buckets) and buckets[-1][0] == len(stages) - 1
    # the headline net: decoder / bottleneck / encoder4 / encoder1-3, and only ~1 MB is left for the end
    model = UNet(1, 8, init_features=32)
    layout = FlatParams(model.named_parameters())
    b = ddp.bucket_plan_for(model, layout)
    assert len(b) == 4 and 4 * (b[-1][2] - b[-1][1]) < 1.5 * (1 << 20)
    # a layout whose finished stages are NOT a tail of the buffer falls back to one bucket after the last stage
    assert ddp.plan_buckets([(0, 10), (10, 20)], 20) == [(1, 0, 20)]
    assert ddp.plan_buckets([(10, 20), (0, 10)], 20, cap_bytes=1) == [(0, 10, 20), (1, 0, 10)]


def _run_bench(extra, env=None):
    import subprocess
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=e, capture_output=True,
                          text=True, timeout=300)


def test_bench_spawns_its_own_ranks_dry_run_gloo():
    """`bench.py --gpus 2` without a launcher: the script starts two rank processes itself, they rendezvous,
    broadcast, run the bucketed reducer, barrier, MAX-reduce the time, and rank 0 prints ONE JSON line."""
    import json
    res = _run_bench(["--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "2", "--warmup", "1", "--features", "8"])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] and out["allreduce_ok"] and out["params_identical_after_broadcast"]
    assert out["config"]["parallelism"] == "dp2" and out["backend"] == "gloo"
    assert out["bucket_launch_order"] == list(range(len(out["buckets"])))


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    res = _run_bench(["--gpus", "2", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0"})
    assert res.returncode == 2 and "WORLD_SIZE=1" in res.stderr
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_bench_under_torchrun_contract_dry_run():
    """The driver's form: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2"""
    import json
    import subprocess
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run",
                          "--steps", "1", "--warmup", "0", "--features", "8"], env=e, capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["allreduce_ok"]


def test_bench_eight_ranks_dry_run_gloo():
    """The N = 8 case the driver runs on an 8-GPU node, rehearsed on CPU: eight self-spawned ranks, one JSON line that
    states the backend and the world size the process group reports."""
    import json
    res = _run_bench(["--gpus", "8", "--backend", "gloo", "--dry-run", "--steps", "2", "--warmup", "1", "--features", "8"])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["world_size_seen"] == 8 and out["backend"] == "gloo"
    assert out["allreduce_ok"] and out["params_identical_after_broadcast"] and out["config"]["parallelism"] == "dp8"


def test_a_rank_that_dies_before_rendezvous_fails_the_launch_quickly():
    """The parent terminates the surviving ranks (exactly the children it started) as soon as one exits non-zero; the
    survivors' rendezvous itself is bounded by OCT_RDZV_TIMEOUT_S."""
    import time
    t0 = time.time()
    res = _run_bench(["--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "1", "--warmup", "0", "--features", "8"],
                     env={"OCT_BENCH_FAIL_RANK": "1", "OCT_RDZV_TIMEOUT_S": "30"})
    assert res.returncode != 0 and time.time() - t0 < 90
    assert "rank 1 exited with code 3" in res.stderr
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
