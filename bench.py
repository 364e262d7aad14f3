#!/usr/bin/env python3
"""Headline benchmark: B-scans/sec of one U-Net training step (forward + CE loss + backward +
bucketed gradient all-reduce + SGD) -- BASELINE.json configs[1]: SOTAS/Layers_Segment UNet(1, 8),
512x1024, bf16, batch 32 per MI355X, synthetic data, random-init weights.

  python bench.py --gpus N --steps K --warmup W

N > 1 runs one process per GPU.  Either the caller launches the ranks (torch.distributed.run: RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or -- when WORLD_SIZE is absent -- this
script starts N rank processes itself BEFORE anything touches the GPU and relays rank 0's JSON line.
A WORLD_SIZE that contradicts --gpus is an error (exit 2), never a relabelled single-GPU number.

Prints ONE JSON line on rank 0.  `roofline` is measured live: every MFMA conv launch of one extra
(untimed) step is bracketed with HIP events on the launch stream; `cpu_baseline` times the oracle's
stock-torch port (oracle/torch_unet.py) on the host cores for a bounded sample (rank 0, N=1 only).

  --backend gloo --dry-run   no kernels, CPU tensors: rendezvous, parameter/buffer broadcast, the
                             bucketed reducer on the real flat-parameter layout, barrier, MAX-reduce and
                             the JSON line (tests/test_ddp_cpu.py runs it with --gpus 2).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_BSCAN_TRAIN = 578.009e9   # 3*F_fwd - dgrad(enc1conv1), UNet(1,8) @ 512x1024
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
TRAFFIC_FILE = os.path.join("profiles", "r03_traffic.json")


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


def bionet_train_flops(classes, h, w, in_ch=1):
    """Algorithmic conv FLOPs of one B-scan's training step of BioNet_2020.UNet (three poolings, widths 64-512,
    BioNet_2020.py:24-75): 3 x forward minus the data gradient of the first convolution, as train_flops()."""
    first = 2.0 * h * w * 9 * in_ch * 64
    fwd, hh, ww, cin = 0.0, h, w, in_ch
    for c in (64, 128, 256, 512):
        fwd += 2.0 * hh * ww * 9 * (cin * c + c * c)
        cin = c
        if c != 512:
            hh //= 2
            ww //= 2
    c = 512
    for _ in range(3):
        fwd += 2.0 * hh * ww * c * (c // 2) * 4          # ConvTranspose2d k2 s2
        hh *= 2
        ww *= 2
        c //= 2
        fwd += 2.0 * hh * ww * 9 * (2 * c * c + c * c)
    fwd += 2.0 * hh * ww * 64 * classes
    return 3.0 * fwd - first


def csrc_digest():
    """sha256 over the kernel sources: profiles/*_traffic.json records it, so a traffic figure measured on
    other kernels than the ones being timed is recognised as stale."""
    d = os.path.join(ROOT, "retinal_oct_image_segmentation_via_deep_learning_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")) or name == "Makefile":
            with open(os.path.join(d, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
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
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe-inclusive side measurement (N = 1)")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the fp32 parity-mode side measurement (N = 1)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"))
    ap.add_argument("--dry-run", action="store_true", help="no kernels: exercise the multi-rank plumbing on CPU tensors")
    ap.add_argument("--bucket-mb", type=float, default=3.0, help="gradient bucket threshold (MB)")
    ap.add_argument("--config", default="cfg2", choices=("cfg2", "cfg1", "cfg4", "cfg5", "relaynet", "mgunet2"),
                    help="cfg2 (default, the headline line the driver runs): Layers_Segment UNet(1,8) 512x1024 batch 32; "
                         "cfg1: BioNet_2020.UNet(1,2) 256x256 batch 4 (add --graph: one hipGraph replay per step); "
                         "cfg4: attention-gated AttU_Net(1,3) 496x768 batch 16; cfg5: volumetric UNet3D 64x512x512 batch 4 "
                         "(BASELINE configs[0] / [3] / [4]; single GPU; --batch/--height/--width/--depth shrink them); "
                         "relaynet: ReLayNet_2017.ReLayNet(1,10) 496x768 batch 16; mgunet2: MGUNet_2021.MGUNet_2(1,11) 496x768 "
                         "batch 16 (SURVEY 8(f) block families: throughput evidence)")
    ap.add_argument("--depth", type=int, default=64, help="cfg5: slices per volume")
    ap.add_argument("--graph", action="store_true",
                    help="replay forward+loss+backward as one HIP graph (no gain at batch 32: the queue never runs dry)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------
# launcher: the parent never imports torch.cuda / never touches the GPU
# ---------------------------------------------------------------------------------------------------
def launch_ranks(args) -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, text=rank == 0))

    def relay(stream):      # ONE JSON line on stdout: anything else rank 0 (or a library under it) prints goes to stderr
        for line in stream:
            (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()
    pump = threading.Thread(target=relay, args=(procs[0].stdout,), daemon=True)
    pump.start()
    rc = 0
    pending = set(range(args.gpus))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                for o in pending:
                    procs[o].terminate()          # exactly the children started above
        time.sleep(0.05)
    pump.join(timeout=10)
    return rc


# ---------------------------------------------------------------------------------------------------
# dry run (CPU): everything of the N > 1 path except the kernels
# ---------------------------------------------------------------------------------------------------
def dry_run(args, rank, world):
    import torch
    import torch.distributed as dist
    from retinal_oct_image_segmentation_via_deep_learning_amd import UNet, ddp
    from retinal_oct_image_segmentation_via_deep_learning_amd.optim import FlatParams

    torch.manual_seed(100 + rank)       # different on purpose: the broadcast must make them rank 0's
    model = UNet(1, args.classes, init_features=args.features)
    layout = FlatParams(model.named_parameters())
    ddp.broadcast_parameters(layout.flat_p)
    ddp.broadcast_buffers(model)
    buckets = ddp.bucket_plan_for(model, layout, int(args.bucket_mb * (1 << 20)))
    red = ddp.GradAllReducer(layout.flat_g, world, buckets)
    nstages = len(model._engine.backward_stages())
    ddp.barrier()
    t0 = time.perf_counter()
    ok = True
    for step in range(args.warmup + args.steps):
        if step == args.warmup:
            ddp.barrier()
            t0 = time.perf_counter()
        layout.flat_g.fill_(float(rank + 1 + step))
        for idx in range(nstages):      # what UNetEngine.backward does with a stage hook
            if idx in red.flush_stages:
                red.stage_done(idx)
        scale = red.finish()
        expect = sum(r + 1 + step for r in range(world))
        ok = ok and bool((layout.flat_g == expect).all()) and scale == 1.0 / world
    ddp.barrier()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    csum = torch.tensor([float(layout.flat_p.double().sum())], dtype=torch.float64)
    same = True
    if world > 1:
        lo, hi = csum.clone(), csum.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        same = bool(lo.item() == hi.item())
    if rank == 0:
        print(json.dumps({
            "metric": "dry-run (no kernels)", "value": 0.0, "unit": "B-scans/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(tt.item() / max(args.steps, 1) * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "dry-run",
            "dry_run": True, "backend": dist.get_backend() if world > 1 else None,
            "world_size_seen": dist.get_world_size() if world > 1 else 1,
            "allreduce_ok": ok, "params_identical_after_broadcast": same,
            "buckets": [[lo, hi] for _, lo, hi in red.buckets],
            "bucket_launch_order": [k for k, _, _ in red.launch_log[:len(red.buckets)]],
            "config": {"workload": "multi-rank plumbing of the UNet(1,%d,%d) step" % (args.classes, args.features),
                       "global_batch": args.batch * world, "parallelism": f"dp{world}"}}))
    if world > 1:
        dist.destroy_process_group()
    return 0 if (ok and same) else 1


# ---------------------------------------------------------------------------------------------------
def cpu_baseline(args):
    """Reference path on the host cores: cfg1 always, the cfg2 shape at batch 1.  Threads = all
    physical cores of this box's share; when that is more than 16 the run is repeated with 16 threads (a GPU box
    hands one GPU a 16-CPU share of a much larger host, and oversubscribed threads run slower) and the faster of
    the two is the baseline -- both are stated."""
    from oracle import torch_unet
    cores = torch_unet.physical_cores()
    tries = [cores] + ([16] if cores > 16 else [])
    c1 = max((torch_unet.time_train_steps(4, 256, 256, classes=2, iters=6, threads=n, model="bionet", budget_s=4.0)
              for n in tries), key=lambda r: r["bscans_per_s_min"])
    runs = [torch_unet.time_train_steps(1, args.height, args.width, args.classes, args.features, iters=5, threads=n,
                                        budget_s=8.0) for n in tries]
    c2 = max(runs, key=lambda r: r["bscans_per_s_min"])
    others = "; ".join(f"{r['threads']} threads: min {r['s_per_iter_min']:.2f} s/iter" for r in runs if r is not c2)
    return {
        "value": round(c2["bscans_per_s_min"], 3), "unit": "B-scans/s", "cores": c2["threads"], "kind": "port",
        "sample": f"oracle/torch_unet.py (stock torch fp32 port of the reference UNet), batch 1 x {args.height}x"
                  f"{args.width}, fwd+loss+bwd+SGD, {c2['iters']} timed iterations after 1 warm-up on {c2['threads']} "
                  f"threads ({cores} physical cores visible): min {c2['s_per_iter_min']:.2f} s/iter, median "
                  f"{c2['s_per_iter_median']:.2f} s/iter" + (f" [{others}]" if others else ""),
        "median": round(c2["bscans_per_s_median"], 3),
        "cfg1": {"value": round(c1["bscans_per_s_min"], 2), "median": round(c1["bscans_per_s_median"], 2),
                 "unit": "B-scans/s", "cores": c1["threads"],
                 "sample": f"BioNet_2020 UNet(1,2) port, batch 4 x 256x256 (BASELINE configs[0]), {c1['iters']} timed "
                           f"iterations: min {c1['s_per_iter_min'] * 1e3:.0f} ms/iter, median "
                           f"{c1['s_per_iter_median'] * 1e3:.0f} ms/iter"},
    }


def traffic_from_profile(args, launches):
    """HBM bytes per conv launch.  PMC counters cannot be read from inside this process: the figure comes from
    two separate rocprofv3 --pmc passes over this same command (tools/traffic.sh -> tools/traffic_report.py).
    It is reported only when that file was produced from THESE kernel sources and this workload."""
    path = os.path.join(ROOT, TRAFFIC_FILE)
    if not os.path.exists(path):
        return None, f"{TRAFFIC_FILE} missing"
    with open(path) as fh:
        doc = json.load(fh)
    if (args.batch, args.height, args.width, args.features) != (32, 512, 1024, 32):
        return None, "not the profiled workload"
    if doc.get("csrc_digest") != csrc_digest():
        print(f"bench.py: {TRAFFIC_FILE} was measured on other kernel sources (digest {doc.get('csrc_digest')} != "
              f"{csrc_digest()}): roofline.traffic is withheld -- re-run tools/traffic.sh", file=sys.stderr)
        return None, f"{TRAFFIC_FILE} is stale (kernel sources changed since it was measured)"
    conv = doc.get("conv", {})
    if not conv:
        return None, f"{TRAFFIC_FILE} has no conv entry"
    total = (conv["read_GB_per_step"] + conv["write_GB_per_step"]) * 1e9
    return round(total / max(launches, 1)), f"{TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, csrc {doc['csrc_digest']})"


def attunet_train_flops(h, w, in_ch=1, classes=3, ch=(64, 128, 256, 512, 1024)):
    """Algorithmic conv FLOPs of one B-scan's AttU_Net training step (SD_Layer_Net/unet.py:76-150, common.py:6-91):
    2*MAC of every Conv2d, x3 for fwd + dgrad + wgrad, minus the data gradient of the very first conv (SURVEY App. A rule)."""
    def c3(hh, ww, ci, co):
        return 2.0 * hh * ww * 9 * ci * co

    def c1(hh, ww, ci, co):
        return 2.0 * hh * ww * ci * co
    fwd, hh, ww, ci = 0.0, h, w, in_ch
    sizes = []
    for lvl, co in enumerate(ch):                      # conv_block: init_conv + two convs
        if lvl:
            hh //= 2
            ww //= 2
        fwd += c3(hh, ww, ci, co) + 2 * c3(hh, ww, co, co)
        sizes.append((hh, ww))
        ci = co
    first = c3(h, w, in_ch, ch[0])
    for lvl in range(len(ch) - 1, 0, -1):              # Up (bilinear + conv3x3), Att (three 1x1), Up_conv (conv_block on the concat)
        hh, ww = sizes[lvl - 1]
        co, fint = ch[lvl - 1], ch[lvl - 1] // 2
        fwd += c3(hh, ww, ch[lvl], co)
        fwd += 2 * c1(hh, ww, co, fint) + c1(hh, ww, fint, 1)
        fwd += c3(hh, ww, 2 * co, co) + 2 * c3(hh, ww, co, co)
    fwd += c1(h, w, ch[0], classes)
    return 3.0 * fwd - first


def unet3d_train_flops(features, classes, d, h, w, in_ch=1):
    """The same rule for the volumetric U-Net (27 taps per Conv3d, 8 per ConvTranspose3d)."""
    f = features
    fwd = 0.0
    first = 2.0 * d * h * w * 27 * in_ch * f
    vox, cin = d * h * w, in_ch
    for lvl in range(4):
        c = f << lvl
        fwd += 2.0 * vox * 27 * (cin * c + c * c)
        cin = c
        vox //= 8
    fwd += 2.0 * vox * 27 * (cin * 16 * f + 16 * f * 16 * f)
    c = 16 * f
    for lvl in range(4):
        fwd += 2.0 * vox * c * (c // 2) * 8
        vox *= 8
        c //= 2
        fwd += 2.0 * vox * 27 * (2 * c * c + c * c)
    fwd += 2.0 * vox * f * classes
    return 3.0 * fwd - first


def side_config(args) -> int:
    """cfg1 / cfg4 / cfg5 (BASELINE configs[0] / [3] / [4]): single GPU, same JSON contract; not the line the driver parses."""
    import torch
    import torch.nn.functional as F
    from retinal_oct_image_segmentation_via_deep_learning_amd.optim import FusedSGD
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1234)
    if args.config == "cfg1":
        from retinal_oct_image_segmentation_via_deep_learning_amd import ddp
        from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.BioNet_2020 import UNet as BioUNet
        b = args.batch if args.batch != 32 else 4
        h, w = (args.height, args.width) if (args.height, args.width) != (512, 1024) else (256, 256)
        model = BioUNet(1, 2).to(dev).train()
        x = torch.randn(b, 1, h, w, generator=g).to(dev)
        t = torch.randint(0, 2, (b, h, w), generator=g).to(dev)
        trainer = ddp.DataParallelTrainer(model, lr=0.01, momentum=0.9, use_graph=args.graph,
                                          graph_warmup=max(1, min(2, args.warmup - 1)))
        flops = bionet_train_flops(2, h, w)
        unit, workload = "B-scans/s", (f"SOTAS/Layers_Segment BioNet_2020.UNet(1,2) train step, {h}x{w}, batch {b} "
                                      f"(BASELINE configs[0], the reference's own CPU-runnable case)")
        step_desc = ("fwd + CE loss + bwd + fused SGD(momentum), all on liboct_hip.so"
                     + ("; the step is ONE hipGraph replay" if args.graph else ""))

        def step():
            return trainer.step(x, t)[0]
    elif args.config == "cfg4":
        from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Lesions_Segment.SD_Layer_Net import AttU_Net
        b = args.batch if args.batch != 32 else 16
        h, w = (args.height, args.width) if (args.height, args.width) != (512, 1024) else (496, 768)
        model = AttU_Net(1, 3).to(dev).train()
        x = torch.randn(b, 1, h, w, generator=g).to(dev)
        t = torch.randint(0, 3, (b, h, w), generator=g).to(dev)
        opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
        flops = attunet_train_flops(h, w)
        unit, workload = "B-scans/s", f"SOTAS AttU_Net(1,3) (SD_Layer_Net/unet.py:76-150) train step, {h}x{w}, batch {b} (BASELINE configs[3])"
        step_desc = "fwd + torch cross_entropy on the logits + bwd (autograd over the HIP ops layer) + torch SGD(momentum)"

        def step():
            opt.zero_grad(set_to_none=True)
            loss = F.cross_entropy(model(x), t)
            loss.backward()
            opt.step()
            return loss
    elif args.config in ("relaynet", "mgunet2"):
        b = args.batch if args.batch != 32 else 16
        h, w = (args.height, args.width) if (args.height, args.width) != (512, 1024) else (496, 768)
        if args.config == "relaynet":
            from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Lesions_Segment.ReLayNet_2017 import ReLayNet
            ncls = 10
            model = ReLayNet(1, ncls).to(dev).train()
            name = "SOTAS/Lesions_Segment ReLayNet_2017.ReLayNet(1,10) (7x3 convolutions, 64 filters, pool-with-indices / unpool)"
            # conv FLOPs of one forward: 3 encoders + bottleneck + 3 decoders, 7x3 taps, 64 filters; classifier 1x1
            hw = [(h >> i, w >> i) for i in range(4)]
            fwd = 2.0 * 21 * (hw[0][0] * hw[0][1] * 1 * 64 + hw[1][0] * hw[1][1] * 64 * 64 + hw[2][0] * hw[2][1] * 64 * 64
                              + hw[3][0] * hw[3][1] * 64 * 64 + hw[2][0] * hw[2][1] * 128 * 64 + hw[1][0] * hw[1][1] * 128 * 64
                              + hw[0][0] * hw[0][1] * 128 * 64) + 2.0 * h * w * 64 * ncls
            flops = 3.0 * fwd - 2.0 * 21 * h * w * 64
        else:
            from retinal_oct_image_segmentation_via_deep_learning_amd.SOTAS.Layers_Segment.MGUNet_2021 import MGUNet_2
            ncls = 11
            model = MGUNet_2(1, ncls).to(dev).train()
            name = "SOTAS/Layers_Segment MGUNet_2021.MGUNet_2(1,11) (feature_scale 4, multi-scale graph reasoning at the bottleneck)"
            flops = 0.0
        x = torch.randn(b, 1, h, w, generator=g).to(dev)
        t = torch.randint(0, ncls, (b, h, w), generator=g).to(dev)
        opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
        unit, workload = "B-scans/s", f"{name} train step, {h}x{w}, batch {b}"
        step_desc = "fwd + torch cross_entropy on the logits + bwd (autograd over the HIP ops layer) + torch SGD(momentum)"

        def step():
            opt.zero_grad(set_to_none=True)
            loss = F.cross_entropy(model(x), t)
            loss.backward()
            opt.step()
            return loss
    else:
        from retinal_oct_image_segmentation_via_deep_learning_amd.unet3d import UNet3D
        b = args.batch if args.batch != 32 else 4
        d = args.depth
        h, w = (args.height, args.width) if (args.height, args.width) != (512, 1024) else (512, 512)
        model = UNet3D(1, args.classes if args.classes != 8 else 4, init_features=args.features).to(dev).train()
        ncls = model.conv.out_channels
        x = torch.randn(b, 1, d, h, w, generator=g).to(dev)
        t = torch.randint(0, ncls, (b, d, h, w), generator=g).to(dev)
        opt = FusedSGD(list(model.named_parameters()), lr=0.01, momentum=0.9)
        flops = unet3d_train_flops(args.features, ncls, d, h, w)
        unit, workload = "volumes/s", (f"UNet3D(1,{ncls},init_features={args.features}) train step, {d}x{h}x{w}, batch {b} "
                                      f"(BASELINE configs[4]; no reference counterpart)")
        step_desc = "fwd + CE loss + bwd + fused SGD(momentum), all on liboct_hip.so"

        def step():
            loss = model.forward_backward(x, t)
            opt.step()
            return loss[0]
    if args.config in ("relaynet", "mgunet2") and args.warmup < 8:
        # the autograd-driven block families take ~8 steps to reach their steady state (caching allocator, per-shape weight
        # packing): with 3 warm-up steps mgunet2 read 600 B-scans/s for a steady 850; the line reports the count used
        args.warmup = 8
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    value = b * args.steps / elapsed
    achieved = value * flops / 1e12
    print(json.dumps({
        "metric": f"{unit[:-2]}/sec (train fwd+bwd), {args.config}", "value": round(value, 3), "unit": unit, "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": workload, "global_batch": b, "parallelism": "dp1", "step": step_desc + "; inputs resident in HBM"},
        "loss": float(loss),
        "roofline": ({"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                      "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": None,
                      "kernel": "whole step (algorithmic conv FLOPs of the network / step time)",
                      "flop_per_unit": flops} if flops else None),
        "cpu_baseline": None,
        "peak_memory_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)
    return 0


def pcie_inclusive(args, trainer, x, t):
    """The same step fed from PINNED HOST buffers: batch k+1 crosses PCIe on a copy stream (fp32 image + int64 target,
    two device slots) while batch k trains.  Reported next to `value`, never as `value` (inputs resident is the
    contract); N = 1 only."""
    import torch
    hx, ht = x.cpu().pin_memory(), t.cpu().pin_memory()
    dx, dt = [torch.empty_like(x) for _ in range(2)], [torch.empty_like(t) for _ in range(2)]
    copy = torch.cuda.Stream()
    ready = [torch.cuda.Event() for _ in range(2)]
    free = [torch.cuda.Event() for _ in range(2)]
    for e in free:
        e.record()

    def stage(i):
        with torch.cuda.stream(copy):
            copy.wait_event(free[i])
            dx[i].copy_(hx, non_blocking=True)
            dt[i].copy_(ht, non_blocking=True)
            ready[i].record(copy)

    steps = max(4, min(args.steps, 20))
    stage(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        i = k & 1
        if k + 1 < steps:
            stage(i ^ 1)
        torch.cuda.current_stream().wait_event(ready[i])
        trainer.step(dx[i], dt[i])
        free[i].record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"value": round(args.batch * steps / el, 2), "unit": "B-scans/s", "ms_per_step": round(el / steps * 1e3, 3), "steps": steps,
            "host_bytes_per_step": hx.numel() * hx.element_size() + ht.numel() * ht.element_size(),
            "how": "pinned host batch -> device on a copy stream, double buffered, overlapped with the previous step"}


def parity_mode_rate(args, dev, x, t):
    """The same training step in compute_dtype="f32" (fp32 activations, v_mfma_f32_32x32x2_f32): the mode the parity
    tests assert arg-max identity and Dice / IoU within 1e-5 in.  A few steps; never `value`."""
    import torch
    from retinal_oct_image_segmentation_via_deep_learning_amd import UNet, ddp
    torch.manual_seed(0)
    model = UNet(1, args.classes, init_features=args.features, compute_dtype="f32").to(dev).train()
    trainer = ddp.DataParallelTrainer(model, lr=0.01, momentum=0.9)
    b = min(args.batch, 8)          # fp32 activations of batch 32 would double the resident set for a side figure
    xs, ts = x[:b].contiguous(), t[:b].contiguous()
    for _ in range(2):
        trainer.step(xs, ts)
    torch.cuda.synchronize()
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.step(xs, ts)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = {"value": round(b * steps / el, 2), "unit": "B-scans/s", "ms_per_step": round(el / steps * 1e3, 3), "batch": b,
           "steps": steps, "dtype": "f32",
           "how": "compute_dtype='f32': the same kernels' generic path with fp32 storage and exact fp32 MFMA; the precision of "
                  "the arg-max / Dice parity contract (tests/test_gpu_unet.py, test_gpu_fullsize.py)"}
    del trainer, model
    torch.cuda.empty_cache()
    return out


def worker(args) -> int:
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: refusing to report a mislabelled number",
              file=sys.stderr)
        return 2

    if os.environ.get("OCT_BENCH_FAIL_RANK") == os.environ.get("RANK", "0"):   # test hook: a rank that dies before rendezvous
        print("bench.py: OCT_BENCH_FAIL_RANK: this rank exits before the rendezvous", file=sys.stderr)
        return 3
    import torch
    import torch.distributed as dist
    from retinal_oct_image_segmentation_via_deep_learning_amd import UNet, ddp

    if args.config != "cfg2":
        if args.gpus != 1 or args.dry_run:
            print("bench.py: --config cfg1 / cfg4 / cfg5 / relaynet / mgunet2 are single-GPU lines", file=sys.stderr)
            return 2
        return side_config(args)
    backend = "gloo" if args.dry_run and args.backend == "nccl" and not torch.cuda.is_available() else args.backend
    rank, world, local = ddp.init_from_env(backend)
    if args.dry_run:
        return dry_run(args, rank, world)

    if backend == "gloo" and torch.cuda.device_count() > 0:
        local = local % torch.cuda.device_count()   # rehearsal on fewer GPUs than ranks: gloo lets ranks share a device (RCCL does not)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    torch.manual_seed(0)
    model = UNet(1, args.classes, init_features=args.features, compute_dtype="bf16").to(dev).train()
    trainer = ddp.DataParallelTrainer(model, lr=0.01, momentum=0.9, use_graph=args.graph,
                                      graph_warmup=max(1, min(2, args.warmup - 1)),
                                      bucket_cap_bytes=int(args.bucket_mb * (1 << 20)))
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(args.batch, 1, args.height, args.width, generator=g).to(dev)
    t = torch.randint(0, args.classes, (args.batch, args.height, args.width), generator=g).to(dev)

    def sync():
        if world > 1:
            ddp.barrier(local)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(x, t)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(x, t)
    sync()
    elapsed = time.perf_counter() - t0
    rank_ms = [elapsed / args.steps * 1e3] * 2        # [min, max] over ranks of the per-rank mean step time
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        lo = tt.clone()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        rank_ms = [lo.item() / args.steps * 1e3, tt.item() / args.steps * 1e3]
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
    traffic, traffic_source = traffic_from_profile(args, len(prof))
    roofline = {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
        "kernel": "igemm2_kernel + wgrad2_kernel + first_fprop_mfma_kernel + first_wgrad_mfma_kernel (conv stack: every "
                  "convolution / transposed convolution launch of the step except the 1x1 head, %d launches/step)" % len(prof),
        "avg_launch_ms": round(conv_ms / max(len(prof), 1), 4),
        "flop_per_launch": flops_bscan * args.batch / max(len(prof), 1),
        "conv_ms_per_step": round(conv_ms, 3),
        "by_kernel_ms": {k: [v[0], round(v[1], 3)] for k, v in kinds.items()},
        "whole_step_frac": round(value / world * flops_bscan / 1e12 / PEAK_BF16_TFLOPS, 4),
    }

    pcie = None
    if world == 1 and not args.no_h2d:
        pcie = pcie_inclusive(args, trainer, x, t)

    # the precision the parity contract is stated in (arg-max identical / Dice within 1e-5: fp32 storage, exact fp32 MFMA):
    # the same step, a few iterations, so that its cost is visible next to the bf16 headline
    parity = None
    if world == 1 and not args.no_parity_mode:
        parity = parity_mode_rate(args, dev, x, t)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    if rank == 0:
        out = {
            "metric": "B-scans/sec (train fwd+bwd) at batch 32, 512x1024", "value": round(value, 2),
            "unit": "B-scans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "backend": dist.get_backend() if (world > 1 and dist.is_initialized()) else None,
            "world_size_seen": dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1,
            "rank_ms_per_step": {"min": round(rank_ms[0], 3), "max": round(rank_ms[1], 3)},
            "config": {"workload": f"SOTAS/Layers_Segment UNet(1,{args.classes},init_features={args.features}) "
                                   f"train step, {args.height}x{args.width}, batch {args.batch}/GPU "
                                   f"(BASELINE configs[1]{'/[2] data-parallel' if world > 1 else ''})",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "step": "fwd + CE loss + bwd + bucketed grad all-reduce + SGD(momentum); inputs resident in "
                               "HBM (same batch every step, no H2D in the loop); wall-clock mean over the timed steps "
                               "between barrier+synchronize",
                       "grad_buckets_bytes": [4 * (hi - lo) for _, lo, hi in trainer.reducer.buckets],
                       "hip_graph": graph_used, "hip_graph_error": trainer.graph_error},
            "loss": float(loss[0].item()),
            "roofline": roofline, "cpu_baseline": cpu, "pcie_inclusive": pcie, "parity_mode": parity,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    sys.exit(worker(args))


if __name__ == "__main__":
    main()
