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
(and back-propagated) once per step instead of once per view; measured (round 2, profiles/r02e_bench.json) 60 views/s at V = 4 against 18.6 at V = 1.

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

MFMA_BF16_PEAK_TF = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TF = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
NET_FWD_GFLOP = 585.8         # per DualStyleUNet forward (profiles/conv_layers.py)


def joint_transforms(J, dev, seed=7, max_angle=3.14159265 / 6, max_shift=0.05):
    """J random rigid transforms (rotations <= 30 deg, translations <= 5 cm): SURVEY.md 8d config 3."""
    import torch
    g = torch.Generator().manual_seed(seed)
    ax = torch.nn.functional.normalize(torch.randn(J, 3, generator=g))
    ang = torch.rand(J, generator=g) * max_angle
    K = torch.zeros(J, 3, 3)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -ax[:, 2], ax[:, 1], ax[:, 2], -ax[:, 0], -ax[:, 1], ax[:, 0]
    A = torch.eye(4)[None].repeat(J, 1, 1)
    A[:, :3, :3] = torch.eye(3)[None] + torch.sin(ang)[:, None, None] * K + (1 - torch.cos(ang))[:, None, None] * (K @ K)
    A[:, :3, 3] = (torch.rand(J, 3, generator=g) - 0.5) * 2 * max_shift
    return A.to(dev)


class TrainingStep:
    """The training iteration of config 3 as a callable: ``step(i, V)`` renders V cameras of one pose, takes the loss, back-propagates,
    exchanges gradients (N > 1) and applies fused Adam.  Shared by this script and by bench.py's ``full_step`` leg."""

    def __init__(self, dev, viewdirs=True, lpips=False, world=1, rank=0, lr=None, pose_per_rank=False):
        import numpy as np
        import torch
        from animatablegaussians_amd import synth
        from animatablegaussians_amd.avatar import AvatarNet
        from animatablegaussians_amd.parallel import BucketedGradSync
        torch.manual_seed(31359)                                    # the reference's seed (main_avatar.py:817)
        self.dev, self.world, self.rank = dev, world, rank
        self.net = net = AvatarNet.synthetic({'with_viewdirs': viewdirs}, device=dev)
        self.n_params = sum(p.numel() for p in net.parameters())
        # pose_per_rank (round 5, DESIGN section 6 mode (b)): every rank trains on ITS OWN pose (another frame of the sequence, as a data-parallel
        # trainer would deal them) -- nothing of the step is replicated across ranks, so the job scales weakly; the default shards the views of ONE pose
        self.pose_per_rank = bool(pose_per_rank)
        A = joint_transforms(net.lbs.shape[1], dev, seed=7 + (rank if pose_per_rank else 0))
        W = H = 1024
        cams = synth.free_view_cameras(8, img=W)
        self.views = [{'cano2live_jnt_mats': A, 'cano2live_jnt_mats_woRoot': A,
                       'extr': torch.from_numpy(np.ascontiguousarray(c["extr"])).float().to(dev),
                       'intr': torch.from_numpy(np.ascontiguousarray(c["intr"])).float().to(dev), 'img_w': W, 'img_h': H} for c in cams]
        self.target = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(11)).to(dev)
        net.train()
        self.sync = BucketedGradSync(list(net.parameters()))
        # One fused pass over the 224 M parameters.  The benchmark's learning rate is 1e-7, not the trainer's 5e-4: the loss target is random
        # noise, and at 5e-4 a few dozen Adam steps towards it inflate the Gaussians until every view has several times the instances
        # (steps of the SAME process went from 66 to 214 ms within 40 steps, profiles/r04_fullstep_degrade.txt) -- the timed workload must
        # not drift while it is being timed.  Adam's kernel does the same work at any learning rate.  AG_BENCH_LR overrides.
        if lr is None:
            lr = float(os.environ.get("AG_BENCH_LR", "1e-7"))
        self.lr = lr
        # the optimizer step: FusedAdam = torch.optim.Adam's update as this library's streaming kernel (include/ag_optim.h; same state layout);
        # AG_BENCH_ADAM=torch: torch's own fused multi-tensor kernel (the A/B)
        self.adam = os.environ.get("AG_BENCH_ADAM", "ag")
        if self.adam == "torch":
            self.opt = torch.optim.Adam(net.parameters(), lr=lr, fused=True)
        else:
            from animatablegaussians_amd.optim import FusedAdam
            self.opt = FusedAdam(net.parameters(), lr=lr)
        self.lp = None
        if lpips:
            from animatablegaussians_amd import losses
            from animatablegaussians_amd.lpips import LPIPS
            self.lp = LPIPS(net='vgg').to(dev)
            m = torch.from_numpy(synth.body_mask(H).copy()).to(dev)
            self.gt_items = {'color_img': self.target, 'mask_img': m, 'boundary_mask_img': torch.zeros_like(m),
                             'mask_bbox': losses.mask_bbox(synth.body_mask(H))}      # from the host copy, as a data loader would
            self.bg_dev = torch.zeros(3, device=dev)
            self.weights = {'l1': 1.0, 'mask': 0.1, 'lpips': 0.1, 'offset': 0.005}

    def loss_of(self, out):
        import torch
        if self.lp is not None:
            from animatablegaussians_amd import losses
            return losses.training_loss(out, self.gt_items, self.bg_dev, self.weights, lpips=self.lp, patch_size=512)[0]
        return (out['rgb_map'] - self.target).abs().mean() + 0.005 * torch.linalg.norm(out['offset'], dim=-1).mean()

    def cameras(self, i, V):
        if self.pose_per_rank:          # every rank its own pose: the cameras need not differ between ranks
            return [self.views[(i * V + j + self.rank) % len(self.views)] for j in range(V)]
        return [self.views[((i * self.world + self.rank) * V + j) % len(self.views)] for j in range(V)]

    def infer(self, i, V=1):
        import torch
        mine = self.cameras(i, V)
        items = dict(mine[0])
        self.net.get_pose_map(items)
        with torch.no_grad():
            self.net.render(items, bg_color=(0., 0., 0.)) if V == 1 else self.net.render_views(items, mine, bg_color=(0., 0., 0.))

    def __call__(self, i, V=1):
        mine = self.cameras(i, V)
        items = dict(mine[0])
        self.net.get_pose_map(items)
        self.sync.zero()
        if V == 1:
            loss = self.loss_of(self.net.render(items, bg_color=(0., 0., 0.)))
        else:
            loss = sum(self.loss_of(o) for o in self.net.render_views(items, mine, bg_color=(0., 0., 0.))) / V
        loss.backward()
        self.sync.finish()
        self.opt.step()


def timed(fn, steps, warmup, dev):
    """ms per call of fn(i) over `steps` calls after `warmup`, device-synchronised on both sides."""
    import torch
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        fn(warmup + i)
    torch.cuda.synchronize(dev)
    return 1e3 * (time.perf_counter() - t0) / steps


def timed_median(fn, steps, warmup, dev):
    """median ms of `steps` individually synchronised calls (robust against a one-off stall, e.g. the allocator re-growing its pool
    after the arithmetic mode changed)."""
    import numpy as np
    import torch
    for i in range(warmup):
        fn(i)
    ts = []
    for i in range(steps):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        fn(warmup + i)
        torch.cuda.synchronize(dev)
        ts.append(1e3 * (time.perf_counter() - t0))
    return float(np.median(ts))


def conv_roofline(dev, steps=3):
    """MFMA roofline of the convolution kernels, measured live, in the product's arithmetic (split_f16: three fp16 products per fp32
    product; split_bf16: six bf16 products) and, beside it, in the fp32-MFMA mode.  One DualStyleUNet (the colour / position configuration) forward + backward on ONE
    stream with every gather-conv / wgrad launch bracketed by HIP events on its launch stream (ag_prof_*); achieved = the launches' own
    algorithmic FLOPs (2 per multiply-add of the un-padded implicit GEMM, summed by the library) / their summed duration."""
    import torch
    from animatablegaussians_amd import _lib, conv as agc, synth
    from animatablegaussians_amd.avatar import AvatarNet
    torch.manual_seed(31359)
    av = AvatarNet.synthetic({'with_viewdirs': True}, device=dev)       # the product's three networks, run as the product runs them:
    net = av                                                            # one grouped chain (grouped.py), G = 3 encoders / G = 6 decoders
    pose = synth.pose_map(512).to(dev)[0]
    gen = torch.Generator().manual_seed(4242)
    vf = [torch.randn(1, 128, 128, 128, generator=gen).to(dev) for _ in range(2)]
    ups = [torch.randn(1, c, 1024, 1024, generator=gen).to(dev) for c in (6, 16, 6)]

    def fwd(_i):
        with torch.no_grad():
            av.get_maps(pose, vf[0], vf[1])

    def one(_i):
        net.zero_grad(set_to_none=True)
        maps = av.get_maps(pose, vf[0], vf[1])
        torch.autograd.backward(list(maps), ups)

    def measure():
        one(0)
        fwd_ms = timed_median(fwd, 5, 2, dev)               # wall time of the pass as the product runs it (two decoder streams)
        both_ms = timed_median(one, 5, 2, dev)
        prev = os.environ.get("AG_SINGLE_STREAM")
        os.environ["AG_SINGLE_STREAM"] = "1"                # per-kernel durations: no co-running kernels
        try:
            for i in range(4):                              # the clocks are at speed before the bracketed passes
                one(i)
            torch.cuda.synchronize(dev)
            _lib.prof_enable([_lib.AG_K_GATHER_CONV, _lib.AG_K_WGRAD])
            for i in range(steps):
                one(i)
            torch.cuda.synchronize(dev)
            n, ms, work = _lib.prof_collect_work()
            _lib.prof_enable([])
        finally:
            if prev is None:
                os.environ.pop("AG_SINGLE_STREAM", None)
            else:
                os.environ["AG_SINGLE_STREAM"] = prev
        r = {"three_networks_forward_ms": round(fwd_ms, 2), "three_networks_forward_backward_ms": round(both_ms, 2),
             "three_networks_forward_TFLOPs": round(3 * NET_FWD_GFLOP / fwd_ms, 1),
             "three_networks_forward_backward_TFLOPs": round(9 * NET_FWD_GFLOP / both_ms, 1)}
        tot_w = tot_ms = 0.0
        for k in ("gather_conv_kernel", "wgrad_kernel"):
            if n[k]:
                r[k] = {"launches_timed": n[k], "avg_launch_us": round(1e3 * ms[k] / n[k], 2), "TFLOPs": round(work[k] / (ms[k] * 1e-3) / 1e12, 2),
                        "GFLOP_per_pass_of_the_three_networks": round(work[k] / steps / 1e9, 1)}
                tot_w += work[k]
                tot_ms += ms[k]
        r["achieved"] = round(tot_w / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0, 2)
        return r, both_ms

    mode0 = agc.get_math()
    try:
        agc.set_math("fp32")
        f32, f32_both = measure()
        f32.update({"peak": MFMA_F32_PEAK_TF, "frac": round(f32["achieved"] / MFMA_F32_PEAK_TF, 4),
                    "whole_network_frac": round(9 * NET_FWD_GFLOP / f32_both / MFMA_F32_PEAK_TF, 4)})
        agc.set_math(mode0)
        out, both_ms = measure()
    finally:
        agc.set_math(mode0)
    terms = {"split_f16": 3, "split_bf16": 6, "split_bf16x3": 3, "f16": 1}.get(mode0, 0)
    part = "fp16" if mode0 in ("split_f16", "f16") else "bf16"
    ach = out["achieved"]
    out.update({"bound": "mfma", "unit": "TFLOP/s", "traffic": None, "math": mode0,
                "kernel": "gather_conv kernels + wgrad kernels (every convolution of the three DualStyleUNets' forward + backward as the product "
                          "runs them: one grouped launch chain, network/avatar.py:93-124)"})
    if terms:
        out.update({"executed": round(terms * ach, 1), "peak": MFMA_BF16_PEAK_TF, "frac": round(terms * ach / MFMA_BF16_PEAK_TF, 4),
                    "frac_of_fp32_mfma_peak": round(ach / MFMA_F32_PEAK_TF, 4),
                    "whole_network_frac_of_fp32_mfma_peak": round(9 * NET_FWD_GFLOP / both_ms / MFMA_F32_PEAK_TF, 4),
                    "note": f"achieved = algorithmic fp32 FLOPs of the bracketed launches / their summed HIP-event durations; every fp32 product is "
                            f"{terms} {part} MFMA products, so executed = {terms} x achieved is what is priced against the dense {part} MFMA peak "
                            "(the same 2.5 PFLOP/s for both 16-bit types).  With real (random-mantissa) operands the 16-bit pipe is power-limited to "
                            "~0.66 of that peak on this part (profiles/r02_conv_split_engine.md)",
                    "matrix_pipe_power_floor": {"ns_per_mfma_per_simd_real_operands": 20.7, "ns_per_mfma_per_simd_constant_operands": 13.6,
                                                "executed_TFLOPs_at_the_floor": 1618.0,
                                                "frac_of_the_floor": round(terms * ach / 1618.0, 4),
                                                "source": "profiles/r05_mfma_floor_f16.txt (profiles/ub/mfma_floor_f16.hip: a pure v_mfma_f32_32x32x16_f16 stream on "
                                                          "operands with random mantissas, any occupancy): no kernel on real data can execute more"}})
    else:
        out.update({"peak": MFMA_F32_PEAK_TF, "frac": round(ach / MFMA_F32_PEAK_TF, 4)})
    out["fp32_mfma_mode"] = f32
    return out


def avatar_kernel_rooflines(dev, reps=20):
    """HBM rooflines of the per-Gaussian assembly and skinning kernels (network/avatar.py:84-124: get_positions / get_others / get_colors,
    transform_cano2live) -- the third hand-written stage north_star names.  Each kernel's launches are captured ``reps`` times in a
    hipGraph and replayed between two HIP events (the host needs ~30 us per call, the kernels 10-25: an eager loop would time the host).
    Algorithmic bytes per Gaussian (fp32; the LBS rows in their sparse form, K (joint u8, weight f32) pairs as the kernels read them):
      gather forward   104 read (pixel index 4, 14 map channels 56, xyz 12, raw opacity / scale / rotation 32) + 56 written
      gather backward  124 read (index 4, 14 upstream gradients 56, 8 other-map channels 32, raws 32) + 56 written, plus the zero fill of
                       the three gradient maps (28 channels x S^2 x 4 B: pixels outside the mask have zero gradient)
      LBS forward      28 + 5K read, 28 written;   LBS backward  56 + 5K read, 28 written.
    PMC traffic of the same kernels: profiles/traffic_head.json (separate rocprofv3 --pmc passes)."""
    import json as _json
    import torch
    from animatablegaussians_amd import avatar_ops as ops
    from animatablegaussians_amd.avatar import AvatarRenderCore

    class _Sub:                                            # stands in for autograd's ctx: the Functions' static methods are called directly
        def __init__(self, needs=()):
            self.saved_tensors, self.needs_input_grad = (), needs

        def save_for_backward(self, *ts):
            self.saved_tensors = ts

    core = AvatarRenderCore.synthetic(device=dev)
    N, S = int(core.xyz.shape[0]), int(core.map_side)
    K = int(core.lbs_sparse.idx.shape[0]) if core.lbs_sparse is not None else None
    gen = torch.Generator().manual_seed(7)
    maps = [torch.randn(1, c, S, S, generator=gen).to(dev) for c in (6, 16, 6)]
    A = joint_transforms(core.lbs.shape[1], dev)
    g5 = [torch.randn(N, c, generator=gen).to(dev) for c in (3, 1, 3, 4, 3)]
    gpr = [torch.randn(N, c, generator=gen).to(dev) for c in (3, 4)]
    with torch.no_grad():
        pos, opa, sca, rot, col = ops.gather_activate(*maps, core.pix, core.xyz, core.opacity_raw, core.scaling_raw, core.rotation_raw)

    def gather_fwd():
        ops._GatherActivate.forward(_Sub(), *maps, core.pix, core.xyz, core.opacity_raw, core.scaling_raw, core.rotation_raw)

    ctx_g = _Sub()
    ops._GatherActivate.forward(ctx_g, *maps, core.pix, core.xyz, core.opacity_raw, core.scaling_raw, core.rotation_raw)

    def gather_bwd():
        ops._GatherActivate.backward(ctx_g, *g5)

    def lbs_fwd():
        ops._LbsTransform.forward(_Sub(), pos, rot, core.lbs, A, core.lbs_sparse)

    ctx_l = _Sub((True, True, False, False, False))
    ops._LbsTransform.forward(ctx_l, pos, rot, core.lbs, A, core.lbs_sparse)

    def lbs_bwd():
        ops._LbsTransform.backward(ctx_l, *gpr)

    def graph_us(fn):
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        us = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            e0.record()
            g.replay()
            e1.record()
            e1.synchronize()
            us.append(1e3 * e0.elapsed_time(e1) / reps)
        return float(sorted(us)[len(us) // 2])

    kk = K if K is not None else 44                        # dense rows: 220 B = 44 five-byte pairs' worth
    alg = {"gather_forward": 160 * N, "gather_backward": 180 * N + 28 * S * S * 4, "lbs_forward": (56 + 5 * kk) * N, "lbs_backward": (84 + 5 * kk) * N}
    fns = {"gather_forward": gather_fwd, "gather_backward": gather_bwd, "lbs_forward": lbs_fwd, "lbs_backward": lbs_bwd}
    traffic = {}
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_head.json")) as f:
            tj = _json.load(f).get("kernels", {})
        for name, frag in (("gather_forward", "gather_forward_kernel"), ("gather_backward", "gather_backward_kernel"),
                           ("lbs_forward", "lbs_forward_kernel"), ("lbs_backward", "lbs_backward_kernel")):
            hit = [v for k, v in tj.items() if frag in k]
            if hit:
                traffic[name] = int(hit[0]["hbm_bytes"])
    except (OSError, ValueError, KeyError):
        pass
    out = {"gaussians": N, "sparse_lbs_pairs_per_gaussian": K, "launches_per_figure": reps,
           "note": "hipGraph replays of `launches_per_figure` back-to-back launches between two HIP events; gather backward = its three "
                   "zero fills + the scatter kernel; traffic = PMC bytes per launch of the kernel alone from profiles/traffic_head.json "
                   "(separate rocprofv3 --pmc passes, not this run)"}
    for name, fn in fns.items():
        us = graph_us(fn)
        gbs = alg[name] / (us * 1e-6) / 1e9
        out[name] = {"bound": "hbm", "algorithmic_bytes_per_launch": int(alg[name]), "avg_launch_us": round(us, 2), "achieved": round(gbs, 1),
                     "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4), "traffic": traffic.get(name)}
    return out


def full_step_probe(dev, block=4, blocks=5):
    """bench.py's ``full_step`` leg: BASELINE configs[2] -- the whole training iteration (3 StyleUNets + assembly + LBS + raster,
    loss, backward, fused Adam) at 1 view per step (the reference's own batch shape) and at 4 views of one pose per step, in the product's
    convolution arithmetic; the same two numbers in the other two modes of include/ag_conv.h beside them.  Per mode and batch shape:
    ``blocks`` blocks of ``block`` pipelined steps (device-synchronised at the block ends), back to back after an allocator reset and
    three settling steps; reported = the MEDIAN block (20 steps per figure), with the fastest and slowest beside it."""
    from animatablegaussians_amd import conv as agc
    import numpy as np
    step = TrainingStep(dev)
    mode0 = agc.get_math()
    modes = [mode0] + [m for m in ("f16", "fp32", "split_bf16", "split_bf16x3") if m != mode0]
    def measure(first_pass):
        # Per arithmetic mode (the product's first) and batch shape: reset the caching allocator, let it settle for three untimed steps, then
        # time the blocks back to back.  (Interleaving the modes, as rounds 1-2 did to share clock history, makes the allocator regrow its
        # pool inside timed blocks -- every mode / shape has its own workspace and activation sizes: reserved memory went 11 -> 31 GiB over
        # two interleaved passes at 4.5 GiB allocated, with single blocks 2-8x slow.)
        import torch
        a1 = {m: [] for m in modes}
        a4 = {m: [] for m in modes}
        for m in modes:
            agc.set_math(m)
            for V, acc in ((1, a1), (4, a4)):
                torch.cuda.empty_cache()
                timed(lambda i: step(i, V), 1, 2, dev)
                for _rep in range(blocks):
                    acc[m].append(timed(lambda i: step(i, V), block, 0, dev))
        return a1, a4

    # A block of the product mode more than 3x slower than its fastest one means the pass was disturbed (seen once in ~12 fresh-box runs: the
    # first process on a box, every block of the leg 8x slow while the legs before it were normal).  The pass is then repeated once and
    # the disturbed one is reported beside the result instead of inside it.
    discarded = None
    try:
        t1, t4 = measure(True)
        if os.environ.get("AG_BENCH_FORCE_REMEASURE") == "1" or max(t1[mode0]) > 3.0 * min(t1[mode0]) or max(t4[mode0]) > 3.0 * min(t4[mode0]):
            discarded = {"ms_per_step_1view": [round(float(x), 2) for x in t1[mode0]], "ms_per_step_4views": [round(float(x), 2) for x in t4[mode0]]}
            t1, t4 = measure(False)
    finally:
        agc.set_math(mode0)

    def rec(m):
        ms1, ms4 = float(np.median(t1[m])), float(np.median(t4[m]))
        return {"views_per_s_1view_per_step": round(1e3 / ms1, 2), "ms_per_step_1view": round(ms1, 2),
                "views_per_s_4views_per_step": round(4e3 / ms4, 2), "ms_per_step_4views": round(ms4, 2),
                "ms_per_step_1view_min_max": [round(float(np.min(t1[m])), 2), round(float(np.max(t1[m])), 2)],
                "ms_per_step_4views_min_max": [round(float(np.min(t4[m])), 2), round(float(np.max(t4[m])), 2)]}

    out = {"workload": "BASELINE configs[2]: StyleUNet x3 + LBS + raster fwd+bwd + L1/offset loss + fused Adam, 268 k Gaussians @1024^2",
           "conv_math": mode0, "adam_lr": step.lr, "adam": "animatablegaussians_amd.optim.FusedAdam (include/ag_optim.h)" if step.adam != "torch" else "torch.optim.Adam(fused=True)",
           "adam_lr_note": "the trainer's 5e-4 (configs/avatarrex_zzr/avatar.yaml) on this random-noise target inflates the Gaussians while the step is being "
                           "timed (profiles/r04_fullstep_degrade.txt); Adam's kernel does the same work at any learning rate (AG_BENCH_LR overrides)"}
    out.update(rec(mode0))
    # BASELINE configs[3] at N = 1 (the anchor of the 8-GPU curve: 16 views of one pose per step, which 8 ranks render 2 each) and the
    # reference's WHOLE loss in the step (main_avatar.py:197-246: boundary compositing, L1, mask loss, 512^2 crop, 0.1 x LPIPS-VGG16),
    # both in the product's arithmetic, blocks of 2 steps
    try:
        import torch
        torch.cuda.empty_cache()
        timed(lambda i: step(i, 16), 1, 1, dev)
        t16 = [timed(lambda i: step(i, 16), 2, 0, dev) for _ in range(3)]
        ms16 = float(np.median(t16))
        out["views16_one_pose_configs3_n1"] = {"views_per_s": round(16e3 / ms16, 2), "ms_per_step": round(ms16, 2),
                                               "ms_per_step_min_max": [round(float(np.min(t16)), 2), round(float(np.max(t16)), 2)], "steps_timed": 6}
        # SURVEY 3.4 (`--mode=test`: animation / free-view synthesis): the eval-mode render of one view (three networks forward, assembly, LBS,
        # rasterizer forward), eager and with the networks replayed from ONE captured hipGraph (AvatarNet.enable_graphs)
        step.net.eval()
        try:
            e_ms = float(np.median([timed(lambda i: step.infer(i, 1), block, 0 if r else 2, dev) for r in range(3)]))
            step.net.enable_graphs(True)
            g_ms = float(np.median([timed(lambda i: step.infer(i, 1), block, 0 if r else 3, dev) for r in range(3)]))
            out["inference_1view"] = {"views_per_s": round(1e3 / e_ms, 2), "ms_per_view": round(e_ms, 2),
                                      "views_per_s_hip_graphs": round(1e3 / g_ms, 2), "ms_per_view_hip_graphs": round(g_ms, 2),
                                      "what": "AvatarNet.render in eval mode under no_grad, pose map included (main_avatar.py:525-776)"}
        finally:
            step.net.enable_graphs(False)
            step.net.train()
        del step
        torch.cuda.empty_cache()
        step_lp = TrainingStep(dev, lpips=True)
        timed(lambda i: step_lp(i, 1), 1, 2, dev)
        tl = [timed(lambda i: step_lp(i, 1), block, 0, dev) for _ in range(blocks)]
        msl = float(np.median(tl))
        out["with_the_references_full_loss"] = {"views_per_s_1view_per_step": round(1e3 / msl, 2), "ms_per_step_1view": round(msl, 2),
                                                "ms_per_step_min_max": [round(float(np.min(tl)), 2), round(float(np.max(tl)), 2)],
                                                "loss": "boundary-mask compositing + L1 + 0.1 mask + 512^2 crop + 0.1 LPIPS-VGG16 + 0.005 offset "
                                                        "(main_avatar.py:197-246), random-initialised VGG trunk", "steps_timed": block * blocks}
        n_params = step_lp.n_params
        del step_lp
    except Exception as e:                       # a leg that fails must not take the line with it
        out["extra_legs_error"] = repr(e)[:300]
        n_params = None
    out.update({"steps_timed": [block * blocks, block * blocks], "parameters": n_params if n_params is not None else 0,
                "note": f"{blocks} blocks of {block} pipelined steps per arithmetic mode and batch shape, back to back after an allocator reset and 3 settling steps; the median block"})
    if discarded is not None:
        out["discarded_first_pass"] = dict(discarded, reason="a block more than 3x slower than the fastest of its pass: the pass was repeated once")
    keys = {"fp32": "conv_math_fp32", "split_bf16x3": "conv_math_split_bf16x3_opt_in_not_fp32_grade", "split_bf16": "conv_math_split_bf16",
            "split_f16": "conv_math_split_f16", "f16": "conv_math_f16_opt_in_the_references_cudnn_tf32_operand_grade"}
    for m in modes[1:]:
        out[keys[m]] = rec(m)
    return out


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
    ap.add_argument("--conv-roofline", action="store_true", help="only print the in-run MFMA roofline of the convolution kernels")
    ap.add_argument("--pose-per-rank", action="store_true", help="N > 1: every rank trains on its own pose (x --views cameras of it) instead of sharding the "
                    "views of one pose: no pose-shared work is replicated, the job scales weakly (DESIGN.md section 6, mode (b))")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = None
    if world > 1:
        from animatablegaussians_amd.parallel import init_distributed
        backend, local_rank = init_distributed(world, local_rank)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if args.conv_roofline:
        print(json.dumps({"roofline_mfma": conv_roofline(dev)}), flush=True)
        return

    ts = TrainingStep(dev, viewdirs=not args.no_viewdirs, lpips=args.lpips and not args.infer, world=world, rank=rank, pose_per_rank=args.pose_per_rank)
    net, n_params, lp = ts.net, ts.n_params, ts.lp
    V = args.views
    if args.infer:
        net.eval()
        net.enable_graphs(args.graphs)
        step = lambda i: ts.infer(i, V)      # noqa: E731
    else:
        step = lambda i: ts(i, V)            # noqa: E731

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
    # N > 1: every rank must hold the same parameters after the run (identical averaged gradients -> identical Adam steps): compare a
    # checksum and the largest element-wise difference to rank 0's copy
    replicas_identical = None
    if world > 1 and not args.infer:
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        ref = flat.clone()
        dist.broadcast(ref, src=0)
        diff = (flat - ref).abs().max().reshape(1)
        dist.all_reduce(diff, op=dist.ReduceOp.MAX)
        replicas_identical = bool(float(diff.item()) == 0.0)
        del flat, ref
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
                       "parallelism": "1 process" if world == 1 else
                                      (f"one pose per rank x{world} ({V} view(s) each)" if args.pose_per_rank else f"view-sharded x{world} (the views of ONE pose)")
                                      + f", bucketed RCCL all-reduce of {n_params * 4 >> 20} MB grads",
                       "scaling_mode": None if world == 1 else ("(b) one pose per rank: weak scaling, nothing replicated" if args.pose_per_rank else
                                                                "(a) views of one pose sharded: the pose-shared StyleUNet work is replicated on every rank "
                                                                "(ceiling = t(all views, 1 GPU) / t(views per rank, 1 GPU): DESIGN.md section 6)"),
                       "backend": None if world == 1 else (backend if backend == "nccl" else f"{backend}: fewer GPUs than ranks, ranks share "
                                                           "devices -- a functional run, NOT a measurement"),
                       "replicas_identical_after_run": replicas_identical},
            "roofline": {"kernel": "gather_conv_kernel + wgrad_kernel (all StyleUNet convolutions of the step)", "bound": "mfma",
                         "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TF, 4),
                         "traffic": None, "note": "conv FLOPs of the step / WHOLE step time (lower bound on the kernels' own rate)"},
        }), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
