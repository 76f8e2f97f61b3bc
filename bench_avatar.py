#!/usr/bin/env python
"""Secondary benchmark: the avatar TRAINING ITERATION of SURVEY.md 8(d) config 3 / 4 on the MI355X path.

    python bench_avatar.py --gpus N --steps K --warmup W [--views V] [--no-viewdirs] [--infer]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench_avatar.py --gpus N ...)

One step = one camera view of one pose, exactly what one iteration of the reference trainer does with the render path
(main_avatar.py:186-262; the LPIPS loss tail of SURVEY.md 8f-1 is added with --lpips):

    get_pose_map (LBS of the canonical points, no grad)  ->  AvatarNet.render: 3 x DualStyleUNet (586 GFLOP each,
    MFMA fp32 convolutions) + view-direction encoder -> fused gather/activations -> LBS -> rasterizer @1024^2
    ->  L1 to a fixed random target + 0.005 * |offset|  ->  backward through everything  ->  Adam step.

`--views V` (config 3 proper: "training step, 4 views"): V cameras of the SAME pose per step through
`AvatarNet.render_views` -- position / other networks, 77 % of the colour network, the assembly and the LBS are evaluated
(and back-propagated) once per step instead of once per view; measured 34 views/s at V = 4 against 10.9 at V = 1.

Synthetic subject (AvatarNet.synthetic: 268 348 Gaussians on the 1024x2048 front|back canvas, 4-sparse LBS weights,
55 random rigid joint transforms), default-initialised networks (224 M parameters), 8 free-view cameras round-robin.
N > 1: views are sharded over ranks, gradients exchanged by BucketedGradSync (RCCL all-reduce of 128-MB buckets
launched from autograd hooks, overlapped with the rest of the backward).  `bench.py` stays the headline (raster-only,
BASELINE.json configs[1]); this line documents the whole path.  The `roofline` here is the MFMA one: conv FLOPs of the
step (3 nets x 586 GFLOP x 3 for fwd + input-grad + weight-grad) over the step time, against 157.3 TFLOP/s fp32 MFMA.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TF = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
NET_FWD_GFLOP = 585.8         # per DualStyleUNet forward (profiles/conv_layers.py)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-viewdirs", action="store_true")
    ap.add_argument("--infer", action="store_true", help="eval-mode render only (animation / free-view synthesis)")
    ap.add_argument("--lpips", action="store_true", help="add the reference's full loss tail: boundary compositing, L1, mask loss, "
                    "512^2 crop and LPIPS-VGG16 (weight 0.1) -- SURVEY.md 8f-1")
    ap.add_argument("--graphs", action="store_true", help="with --infer: run the networks from captured hipGraphs")
    ap.add_argument("--views", type=int, default=1, help="cameras of the same pose per step (multi-view step: "
                    "pose-dependent work shared through AvatarNet.render_views)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from animatablegaussians_amd import synth
    from animatablegaussians_amd.avatar import AvatarNet
    from animatablegaussians_amd.parallel import BucketedGradSync

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # one rank per GPU over RCCL ("nccl"); AG_DIST_BACKEND=gloo lets the N > 1 control flow be exercised on a box with
        # fewer GPUs than ranks (ranks then share devices) -- a functional check only, never a measurement
        backend = os.environ.get("AG_DIST_BACKEND", "nccl")
        local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    torch.manual_seed(31359)                                    # the reference's seed (main_avatar.py:817)
    net = AvatarNet.synthetic({'with_viewdirs': not args.no_viewdirs}, device=dev)
    n_params = sum(p.numel() for p in net.parameters())
    J = net.lbs.shape[1]
    g = torch.Generator().manual_seed(7)
    ax = torch.nn.functional.normalize(torch.randn(J, 3, generator=g))
    ang = torch.rand(J, generator=g) * (np.pi / 6)
    K = torch.zeros(J, 3, 3)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -ax[:, 2], ax[:, 1], ax[:, 2], -ax[:, 0], -ax[:, 1], ax[:, 0]
    A = torch.eye(4)[None].repeat(J, 1, 1)
    A[:, :3, :3] = torch.eye(3)[None] + torch.sin(ang)[:, None, None] * K + (1 - torch.cos(ang))[:, None, None] * (K @ K)
    A[:, :3, 3] = (torch.rand(J, 3, generator=g) - 0.5) * 0.1
    A = A.to(dev)
    W = H = 1024
    cams = synth.free_view_cameras(8, img=W)
    views = [{'cano2live_jnt_mats': A, 'cano2live_jnt_mats_woRoot': A,
              'extr': torch.from_numpy(np.ascontiguousarray(c["extr"])).float().to(dev),
              'intr': torch.from_numpy(np.ascontiguousarray(c["intr"])).float().to(dev), 'img_w': W, 'img_h': H} for c in cams]
    target = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(11)).to(dev)

    if args.infer:
        net.eval()
        net.enable_graphs(args.graphs)
        sync = opt = None
    else:
        net.train()
        sync = BucketedGradSync(list(net.parameters()))
        opt = torch.optim.Adam(net.parameters(), lr=5e-4, fused=True)      # one pass over the 224 M parameters

    V = args.views
    lp = None
    if args.lpips and not args.infer:
        from animatablegaussians_amd import losses, synth as _synth
        from animatablegaussians_amd.lpips import LPIPS
        lp = LPIPS(net='vgg').to(dev)
        m = torch.from_numpy(_synth.body_mask(H).copy()).to(dev)
        gt_items = {'color_img': target, 'mask_img': m, 'boundary_mask_img': torch.zeros_like(m),
                    'mask_bbox': losses.mask_bbox(_synth.body_mask(H))}      # from the host copy, as a data loader would
        bg_dev = torch.zeros(3, device=dev)
        weights = {'l1': 1.0, 'mask': 0.1, 'lpips': 0.1, 'offset': 0.005}

    def loss_of(out):
        if lp is not None:
            return losses.training_loss(out, gt_items, bg_dev, weights, lpips=lp, patch_size=512)[0]
        return (out['rgb_map'] - target).abs().mean() + 0.005 * torch.linalg.norm(out['offset'], dim=-1).mean()

    def step(i: int):
        mine = [views[((i * world + rank) * V + j) % len(views)] for j in range(V)]      # this rank's cameras of the step
        items = dict(mine[0])
        net.get_pose_map(items)
        if args.infer:
            with torch.no_grad():
                net.render(items, bg_color=(0., 0., 0.)) if V == 1 else net.render_views(items, mine, bg_color=(0., 0., 0.))
            return
        sync.zero()
        if V == 1:
            loss = loss_of(net.render(items, bg_color=(0., 0., 0.)))
        else:
            loss = sum(loss_of(o) for o in net.render_views(items, mine, bg_color=(0., 0., 0.))) / V
        loss.backward()
        sync.finish()
        opt.step()

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        # conv FLOPs of a step: position + other nets once, colour net = shared 77 % once + 23 % per view
        flops = (2 + 0.77 + 0.23 * V) * NET_FWD_GFLOP * (1 if args.infer else 3) * 1e9
        ach = flops / (ms * 1e-3) / 1e12
        print(json.dumps({
            "metric": ("avatar render (3 StyleUNets + assembly + LBS + raster) views/sec @1024^2" if args.infer else
                       "avatar training iterations/sec (3 StyleUNets + assembly + LBS + raster fwd+bwd + Adam) @1024^2"),
            "value": round(args.gpus * args.steps * V / elapsed, 3), "unit": "views/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"SURVEY 8d config 3: {V} view(s) of one pose per step, whole render path"
                                   + (" (eval)" if args.infer else " + loss + backward + Adam"),
                       "gaussians": int(net.lbs.shape[0]), "parameters": int(n_params), "with_viewdirs": bool(net.with_viewdirs), "views_per_step": V, "lpips_loss_tail": lp is not None, "hip_graphs": bool(args.infer and args.graphs),
                       "parallelism": "1 process" if world == 1 else f"view-sharded x{world}, bucketed RCCL all-reduce of {n_params * 4 >> 20} MB grads"},
            "roofline": {"kernel": "gather_conv_kernel + wgrad_kernel (all StyleUNet convolutions of the step)", "bound": "mfma",
                         "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TF, 4),
                         "traffic": None, "note": "conv FLOPs of the step / WHOLE step time (lower bound on the kernels' own rate)"},
        }), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
