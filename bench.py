#!/usr/bin/env python3
"""Headline benchmark: B-scans/sec of one U-Net training step (forward + CE loss + backward +
gradient all-reduce + SGD) -- BASELINE.json configs[1]: SOTAS/Layers_Segment UNet(1, 8),
512x1024, bf16, batch 32 per MI355X, synthetic data, random-init weights.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0.  `roofline` is measured live: every MFMA conv launch of one extra
(untimed) step is bracketed with HIP events on the launch stream; `cpu_baseline` times the oracle's
stock-torch port (oracle/torch_unet.py) on the host cores for a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_BSCAN_TRAIN = 578.009e9   # 3*F_fwd - dgrad(enc1conv1), UNet(1,8) @ 512x1024
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def train_flops(features, classes, h, w, in_ch=1):
    """Algorithmic conv FLOPs of one B-scan's training step (same rule as SURVEY.md App. A)."""
    f = features
    fwd = 0.0
    first = 2.0 * h * w * 9 * in_ch * f
    hh, ww, cin = h, w, in_ch
    for lvl in range(4):
        c = f << lvl
        fwd += 2.0 * hh * ww * 9 * (cin * c + c * c)
        cin = c
        hh //= 2
        ww //= 2
    fwd += 2.0 * hh * ww * 9 * (cin * 16 * f + 16 * f * 16 * f)
    c = 16 * f
    for lvl in range(4):
        fwd += 2.0 * hh * ww * c * (c // 2) * 4          # ConvTranspose2d k2 s2
        hh *= 2
        ww *= 2
        c //= 2
        fwd += 2.0 * hh * ww * 9 * (2 * c * c + c * c)
    fwd += 2.0 * hh * ww * f * classes
    return 3.0 * fwd - first


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--classes", type=int, default=8)
    ap.add_argument("--features", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay forward+loss+backward as one HIP graph (no gain at batch 32: the queue never runs dry)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from retinal_oct_image_segmentation_via_deep_learning_amd import UNet, ddp

    rank, world, local = ddp.init_from_env("nccl")
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    torch.manual_seed(0)
    model = UNet(1, args.classes, init_features=args.features, compute_dtype="bf16").to(dev).train()
    trainer = ddp.DataParallelTrainer(model, lr=0.01, momentum=0.9, use_graph=args.graph,
                                      graph_warmup=max(1, min(2, args.warmup - 1)))
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(args.batch, 1, args.height, args.width, generator=g).to(dev)
    t = torch.randint(0, args.classes, (args.batch, args.height, args.width), generator=g).to(dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(x, t)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(x, t)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    ms_per_step = elapsed / args.steps * 1e3
    value = args.batch * world * args.steps / elapsed

    # ---- roofline of the MFMA conv stack: one extra step with events around every conv launch ----
    flops_bscan = train_flops(args.features, args.classes, args.height, args.width)
    graph_used, trainer.graph = trainer.graph is not None, None     # the measuring step below runs eagerly
    trainer.use_graph = False
    model._engine.prof = []
    trainer.step(x, t)
    torch.cuda.synchronize()
    prof = model._engine.prof
    model._engine.prof = None
    conv_ms = sum(s.elapsed_time(e) for _, s, e in prof)
    kinds = {}
    for k, s, e in prof:
        kinds.setdefault(k, [0, 0.0])
        kinds[k][0] += 1
        kinds[k][1] += s.elapsed_time(e)
    achieved = flops_bscan * args.batch / (conv_ms * 1e-3) / 1e12
    # HBM bytes per conv launch: PMC counters cannot be read from inside this process, so the figure is
    # the one tools/traffic_report.py derived from two rocprofv3 --pmc passes over this same command
    # (profiles/*_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE of the conv kernels per step), when committed
    traffic = None
    tfile = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_v11_traffic.json")
    if os.path.exists(tfile) and (args.batch, args.height, args.width, args.features) == (32, 512, 1024, 32):
        with open(tfile) as fh:
            conv = json.load(fh).get("conv", {})
        if conv:
            traffic = round((conv["read_GB_per_step"] + conv["write_GB_per_step"]) * 1e9 / max(len(prof), 1))
    roofline = {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
        "kernel": "igemm_kernel + wgrad_kernel (conv stack, %d launches/step)" % len(prof),
        "avg_launch_ms": round(conv_ms / max(len(prof), 1), 4),
        "flop_per_launch": flops_bscan * args.batch / max(len(prof), 1),
        "conv_ms_per_step": round(conv_ms, 3),
        "by_kernel_ms": {k: [v[0], round(v[1], 3)] for k, v in kinds.items()},
        "whole_step_frac": round(value / world * flops_bscan / 1e12 / PEAK_BF16_TFLOPS, 4),
    }

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import torch_unet
        cores = os.cpu_count() or 1
        threads = min(cores, 16)
        bs, best, th = torch_unet.time_train_steps(1, args.height, args.width, args.classes, args.features,
                                                   iters=3, threads=threads)
        cpu_baseline = {"value": round(bs, 3), "unit": "B-scans/s", "cores": th, "kind": "port",
                        "sample": f"oracle/torch_unet.py (stock torch fp32 port of the reference UNet), batch 1 "
                                  f"x {args.height}x{args.width}, fwd+loss+bwd+SGD, best of 3 after 1 warm-up "
                                  f"({best:.2f} s/iter)"}

    if rank == 0:
        out = {
            "metric": "B-scans/sec (train fwd+bwd) at batch 32, 512x1024", "value": round(value, 2),
            "unit": "B-scans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"SOTAS/Layers_Segment UNet(1,{args.classes},init_features={args.features}) "
                                   f"train step, {args.height}x{args.width}, batch {args.batch}/GPU "
                                   f"(BASELINE configs[1]{'/[2] data-parallel' if world > 1 else ''})",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "step": "fwd + CE loss + bwd + grad all-reduce + SGD(momentum)",
                       "hip_graph": graph_used, "hip_graph_error": trainer.graph_error},
            "loss": float(loss[0].item()),
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
