"""Data-parallel step on the real kernels.  A one-GPU box cannot host two RCCL ranks (RCCL refuses two
ranks on one device), so the 2-rank case runs the SAME DataParallelTrainer over gloo with device tensors
(both processes on cuda:0), and RCCL itself is exercised as a 1-rank group with the collectives forced on
(`always_communicate`): bucket launches on the side stream, event hand-over, optimizer wait."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

F, C, H, W, B = 8, 4, 32, 64, 4   # per-rank batch 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _data():
    g = torch.Generator().manual_seed(11)
    return torch.randn(B, 1, H, W, generator=g), torch.randint(0, C, (B, H, W), generator=g)


def _worker(rank, world, port, backend, out_dir, always):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch.distributed as dist
    from retinal_oct_image_segmentation_via_deep_learning_amd import UNet, ddp
    torch.cuda.set_device(0)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    torch.manual_seed(50 + rank)                      # broadcast must overwrite rank 1's weights
    model = UNet(1, C, init_features=F, compute_dtype="f32").cuda().train()
    tr = ddp.DataParallelTrainer(model, lr=0.05, momentum=0.9, bucket_cap_bytes=8 << 10, always_communicate=always)
    x, t = _data()
    lo, hi = ddp.shard_batch(B, rank, world)
    xs, ts = x[lo:hi].cuda(), t[lo:hi].cuda()
    losses = []
    for _ in range(3):
        losses.append(tr.step(xs, ts)[0].item())
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"{backend}{rank}.npz"), p=tr.opt.flat_p.cpu().numpy(), losses=np.array(losses),
             nb=len(tr.reducer.buckets), launches=len(tr.reducer.launch_log),
             rm=model.encoder1.enc1norm1.running_mean.cpu().numpy())
    dist.destroy_process_group()


def _single_process_reference(world):
    """The same arithmetic in one process: per-shard forward_backward (rank-local BN statistics, like stock DDP),
    gradients averaged over the shards, one fused SGD step."""
    from retinal_oct_image_segmentation_via_deep_learning_amd import UNet, ddp
    from retinal_oct_image_segmentation_via_deep_learning_amd.optim import FusedSGD
    torch.manual_seed(50)
    model = UNet(1, C, init_features=F, compute_dtype="f32").cuda().train()
    opt = FusedSGD(list(model.named_parameters()), lr=0.05, momentum=0.9)
    x, t = _data()
    for _ in range(3):
        acc = torch.zeros_like(opt.flat_g)
        for r in range(world):
            lo, hi = ddp.shard_batch(B, r, world)
            model.forward_backward(x[lo:hi].cuda(), t[lo:hi].cuda())
            acc += opt.flat_g
        opt.flat_g.copy_(acc)
        opt.step(grad_scale=1.0 / world)
    torch.cuda.synchronize()
    return opt.flat_p.cpu().numpy()


def test_two_rank_trainer_over_gloo_matches_one_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), "gloo", str(tmp_path), False), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "gloo0.npz"), np.load(tmp_path / "gloo1.npz")
    np.testing.assert_array_equal(r0["p"], r1["p"])          # ranks stay in lock-step
    assert int(r0["nb"]) >= 3 and int(r0["launches"]) == 3 * int(r0["nb"])
    ref = _single_process_reference(world)
    # weight gradients are summed with fp32 atomics (order varies run to run): fp32 round-off, not bit equality
    assert np.abs(r0["p"] - ref).max() <= 3e-4 * np.abs(ref).max()
    assert r0["losses"][-1] < r0["losses"][0]


def test_rccl_group_runs_the_bucketed_exchange(tmp_path):
    """RCCL initialises on this box and the bucketed side-stream exchange of a 1-rank group leaves the
    training trajectory untouched (sum over one rank, scale 1)."""
    mp.spawn(_worker, args=(1, _free_port(), "nccl", str(tmp_path), True), nprocs=1, join=True)
    r = np.load(tmp_path / "nccl0.npz")
    assert int(r["launches"]) == 3 * int(r["nb"]) and int(r["nb"]) >= 3
    from retinal_oct_image_segmentation_via_deep_learning_amd import UNet
    from retinal_oct_image_segmentation_via_deep_learning_amd.optim import FusedSGD
    torch.manual_seed(50)
    model = UNet(1, C, init_features=F, compute_dtype="f32").cuda().train()
    opt = FusedSGD(list(model.named_parameters()), lr=0.05, momentum=0.9)
    x, t = _data()
    for _ in range(3):
        model.forward_backward(x.cuda(), t.cuda())
        opt.step()
    torch.cuda.synchronize()
    ref = opt.flat_p.cpu().numpy()
    assert np.abs(r["p"] - ref).max() <= 3e-4 * np.abs(ref).max()    # fp32 atomics: summation order varies


def test_bench_two_ranks_real_kernels_one_json_line():
    """`bench.py --gpus 2` end to end with the real kernels: two self-spawned ranks (gloo, sharing the box's one GPU --
    RCCL refuses that), broadcasts, bucketed exchange under backward, barrier + MAX-reduced time, and exactly ONE JSON
    line on stdout that says n_gpus 2 and a global batch of 2 x the per-rank one."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--batch", "2", "--height", "64", "--width", "128", "--no-cpu-baseline"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and np.isfinite(out["loss"]) and len(out["config"]["grad_buckets_bytes"]) >= 1
