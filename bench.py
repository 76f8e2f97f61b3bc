#!/usr/bin/env python
"""Headline benchmark: rendered views/sec (forward + backward) at 1024x1024 over ~250 k Gaussians.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload = BASELINE.json configs[1] ("avatarrex_zzr: 512^2 front/back maps (~250k Gaussians), 1 view @1024^2,
1xMI355X fwd+bwd"), synthetic: animatablegaussians_amd.synth.avatar_map_gaussians() (268 348 Gaussians on the
reference's 1024x2048 front|back canvas) seen from 8 free-view cameras (f = 1100, 2.5 m).  One step = forward + backward
of one view with resident random upstream gradients: preprocess -> tile counts/scan -> scatter -> per-tile sort -> blend,
then blend backward -> preprocess backward.  All inputs are resident in HBM before the timed region.  The headline loop
is the library-owned step (rasterizer.FusedRasterStep: one native call per view, ag_raster_forward_backward); consecutive
views rotate over its three internal HIP streams (--streams), as a multi-view trainer issues them: views are independent,
and the one host synchronisation per view (the reference's num_rendered read-back) overlaps the other view's kernels.  The
same workload through the reference's operator surface (GaussianRasterizer + public torch.autograd.backward) and on ONE stream
with ordered steps is timed next to it (`operator_path`, `sequential`).

Multi-GPU: views are sharded over ranks (weak scaling: every rank renders K views of the same Gaussians).  The
exchange step of view-sharded rendering is the sum over views of the per-Gaussian attribute gradients (14 floats per
Gaussian, 15 MB) -- one RCCL all-reduce per step, issued on a side stream and overlapped with the next view.

The JSON line also carries
  roofline     : dominant kernel (blend backward), ALGORITHMIC bytes/launch (8T + 44R + 28WH + 40P)
                 / its HIP-event duration measured live in the timed region, against 8 TB/s HBM3E;
  cpu_baseline : the CPU oracle (a restatement of the reference's CUDA kernels -- the reference has no CPU
                 rasterizer) timed on this box's host cores on a bounded sample of the same workload;
  full_step    : BASELINE configs[2], the whole training iteration (three StyleUNets + assembly + LBS + raster, loss, backward,
                 fused Adam) at 1 and at 4 views per step, in the product's convolution arithmetic (split_f16) and, in the same
                 process, in the other three modes of include/ag_conv.h;
  roofline_mfma: the convolution kernels' own rate (HIP events around every launch of the three networks' forward + backward), in the
                 product's arithmetic against the dense 16-bit MFMA peak (executed = 3 x algorithmic FLOPs: three fp16 products per fp32
                 product) and in the fp32-MFMA mode against the fp32 MFMA peak;
  stress_1m_2048: BASELINE configs[4] on one GPU (1.07 M Gaussians, 2048^2): views/s, blend-backward us, algorithmic GB/s;
  cpu_baseline_styleunet / cpu_baseline_lbs: SURVEY.md 8(d)(i)(ii) -- the reference's DualStyleUNet forward and forward + backward
                 (oracle/dual_styleunet_oracle.py: the same torch CPU ops in the same order, pinned against the reference module's
                 golden) and the reference's LBS einsum path (oracle/avatar_oracle.py) on this box's host cores.
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

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--prewarm", type=int, default=300, help="untimed steps BEFORE the --warmup steps (clocks, allocator pools, the "
                    "binning capacity of every camera): with the driver's --warmup 5 --steps 20 the timed region is 3 ms long and read 5050 "
                    "views/s where the five blocks after it read 6860-7220 (profiles/r04_bench_driver_cmd.json); not part of the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle timing (profiling runs)")
    ap.add_argument("--breakdown", action="store_true", help="add a per-kernel HIP-event breakdown pass")
    ap.add_argument("--streams", type=int, default=3, help="render consecutive views round-robin on this many HIP streams: "
                    "the host-side wait for a view's instance count (the reference's num_rendered read-back) overlaps the kernels "
                    "of the previous views, and the latency-bound stages of one view fill the gaps of another.  Library-owned step, "
                    "same box: 2 streams 4860, 3 streams 5250, 4 streams 4560 views/s")
    ap.add_argument("--exchange-every", type=int, default=0, help="N > 1: the ranks all-reduce their per-Gaussian gradient sums once per this "
                    "many steps (views) of a rank; 0 = 16 // N, BASELINE configs[3]'s 16-view iteration sharded over the ranks (2 views per "
                    "rank and exchange at N = 8).  In between a rank adds its views' gradients on the device")
    ap.add_argument("--engine-threads", action="store_true", help="operator path: keep autograd's multithreaded engine (default: "
                    "backward nodes run on the calling thread)")
    ap.add_argument("--operator-path", action="store_true", help="time GaussianRasterizer + torch.autograd.backward (the reference's "
                    "operator surface) in the headline loop instead of the library-owned forward+backward step")
    ap.add_argument("--no-full-step", action="store_true", help="skip the full_step / roofline_mfma legs (BASELINE configs[2]: the whole "
                    "training iteration with the three StyleUNets, ~15 s) -- profiling runs of the rasterizer")
    ap.add_argument("--no-stress", action="store_true", help="skip the stress_1m_2048 leg (BASELINE configs[4] on one GPU, ~3 s)")
    ap.add_argument("--_cpu-worker", dest="cpu_worker", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:                                    # child process of a cpu_baseline_* leg (no GPU work)
        return {"lbs": _cpu_lbs_worker, "styleunet": _cpu_styleunet_worker}[args.cpu_worker]()

    import numpy as np
    import torch
    import torch.distributed as dist

    from animatablegaussians_amd import _lib, camera, synth
    from animatablegaussians_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

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

    # ---- synthetic scene, resident on the GPU -------------------------------------------------------------------
    W = H = 1024
    av = synth.avatar_map_gaussians()
    P = av["means3D"].shape[0]
    cams_np = synth.free_view_cameras(8, img=W)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    means3D = t(av["means3D"]).requires_grad_(True)
    scales = t(av["scales"]).requires_grad_(True)
    rotations = t(av["rotations"]).requires_grad_(True)
    opacities = t(av["opacities"]).requires_grad_(True)
    colors = t(av["colors"]).requires_grad_(True)
    leaves = [means3D, scales, rotations, opacities, colors]
    bg = t(av["bg"])
    settings = []
    for c in cams_np:
        cm = camera.camera_from_intr_extr(c["extr"], c["intr"], W, H)
        settings.append(GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cm["tanfovx"], tanfovy=cm["tanfovy"], bg=bg, scale_modifier=1.0,
            viewmatrix=t(cm["viewmatrix"]), projmatrix=t(cm["projmatrix"]), sh_degree=0, campos=t(cm["campos"]),
            prefiltered=False, debug=False))
    rasterizers = [GaussianRasterizer(s) for s in settings]
    up = synth.upstream_grads(W, H, 12345 + rank)
    g_color, g_depth, g_alpha = t(up["dL_dcolor"]), t(up["dL_ddepth"]), t(up["dL_dalpha"])
    comm_stream = None
    exch_every = (args.exchange_every if args.exchange_every > 0 else max(1, 16 // world)) if world > 1 else 0
    R_seen = []

    # One step = forward + backward of one view.  The default goes through the library-owned step (rasterizer.FusedRasterStep ->
    # ag_raster_forward_backward): the upstream image gradients are resident, so forward and backward of a view are ONE native call;
    # consecutive views alternate between the step's internal HIP streams (views are independent, and the one host wait per view --
    # the instance count -- overlaps the other stream's kernels).  The reference's operator path (GaussianRasterizer forward, public
    # torch.autograd.backward) is timed after the headline region and reported as `operator_path`; --operator-path makes it the
    # headline loop instead.  Both run exactly the same kernels (tests/test_raster_gpu.py compares their results).
    from animatablegaussians_amd.rasterizer import FusedRasterStep
    fused = FusedRasterStep(P, W, H, dev, n_streams=max(1, args.streams))
    det = [t.detach() for t in (means3D, colors, opacities, scales, rotations)]
    streams = [torch.cuda.Stream(dev) for _ in range(args.streams)] if args.streams > 1 else None   # operator path only
    if not args.engine_threads:
        # operator path: the graph of a step is ONE node; run it on the calling thread instead of autograd's per-device worker
        torch.autograd.set_multithreading_enabled(False)

    # Exchange step of view sharding (N > 1): parallel.ViewGradExchange -- the per-Gaussian attribute gradients (P x 14 fp32 = 15 MB) packed on
    # the view's own stream, summed over the rank's views on a communication stream, one all-reduce per exch_every steps under the next
    # views' kernels (rounds 1-4 made every step's kernels wait for the previous step's all-reduce: compute and exchange never overlapped)
    n_slots = max(1, args.streams)
    xchg = None
    if world > 1:
        from animatablegaussians_amd.parallel import ViewGradExchange
        xchg = ViewGradExchange(P, 14, dev, n_slots, exch_every)
        comm_stream = xchg.comm

    def exchange(k, grads, producer_stream):
        xchg.submit(k, grads, producer_stream)

    # the cameras of a capture rig are fixed: one prepared handle (argument structures + output images) per (camera, stream slot), all
    # built here, before the warm-up, as a multi-view trainer builds them once at start-up (allocation only: no kernel runs)
    handles = {}
    if not args.operator_path:
        for v in range(len(settings)):
            for k in range(len(fused.slots)):
                handles[(v, k)] = fused.prepare(settings[v], g_color, g_depth, g_alpha, k)

    def step_fused(i: int, slot=None):
        v = (i * world + rank) % len(settings)                     # this rank's view of the step
        k = (i % len(fused.slots)) if slot is None else slot
        h = handles.get((v, k))
        if h is None:
            h = handles[(v, k)] = fused.prepare(settings[v], g_color, g_depth, g_alpha, k)
        _, _, _, _, g = fused.run(h, det[0], det[1], det[2], det[3], det[4], inputs_outlive_join=True)
        if world > 1:
            exchange(k, [g["dL_dmeans3D"], g["dL_dscales"], g["dL_drotations"], g["dL_dopacity"], g["dL_dcolors"]], fused.slots[k]["stream"])

    def step_operator_on(i: int):
        r = rasterizers[(i * world + rank) % len(rasterizers)]
        means2D = torch.zeros_like(means3D, requires_grad=True)
        color, radii, depth, alpha = r(means3D=means3D, means2D=means2D, opacities=opacities, shs=None,
                                       colors_precomp=colors, scales=scales, rotations=rotations, cov3D_precomp=None)
        torch.autograd.backward([color, depth, alpha], [g_color, g_depth, g_alpha])
        if world > 1:
            exchange(i % n_slots, [means3D.grad, scales.grad, rotations.grad, opacities.grad, colors.grad], torch.cuda.current_stream(dev))
        for leaf in leaves:
            leaf.grad = None

    def step_operator(i: int):
        if streams is None:
            return step_operator_on(i)
        st = streams[i % len(streams)]
        st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st):
            step_operator_on(i)

    step = step_operator if args.operator_path else step_fused

    def step_on(i: int):               # one stream, every step ordered after the previous one
        return step_operator_on(i) if args.operator_path else step_fused(i, slot=0)

    def sync_all():
        fused.join()
        if streams is not None:
            for st in streams:
                torch.cuda.current_stream(dev).wait_stream(st)
        if world > 1:
            torch.cuda.current_stream(dev).wait_stream(comm_stream)
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- warm-up, then the timed region -----------------------------------------------------------------------
    for i in range(max(0, args.prewarm)):
        step(i)
    sync_all()
    for i in range(args.warmup):
        step(i)
    sync_all()
    DOM = 5  # AG_K_BLEND_BACKWARD: bracket the dominant kernel with HIP events on its own launch stream
    _lib.prof_enable([DOM])
    sync_all()
    if xchg is not None:
        xchg.reset()                    # N > 1: the timed region starts on an iteration boundary of the exchange
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    sync_all()
    elapsed = time.perf_counter() - t0
    prof = _lib.prof_collect()
    _lib.prof_enable([])
    # stability evidence for short timed regions (the driver's --steps 20 is a 3-ms region that includes the fill and drain of the
    # stream pipeline): the same number of steps again, five more times, each block timed like the region above
    value_blocks = []
    for b in range(5):
        sync_all()
        tb = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + args.steps * (b + 1) + i)
        sync_all()
        value_blocks.append(time.perf_counter() - tb)
    if world > 1:
        tbk = torch.tensor(value_blocks, device=dev, dtype=torch.float64)
        dist.all_reduce(tbk, op=dist.ReduceOp.MAX)
        value_blocks = [float(x) for x in tbk.tolist()]
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # N > 1, untimed: one more iteration with the all-reduce held back, checked by linearity (a checksum of checksums): the rank's
    # accumulator must hold the sum of what its steps packed, and after the all-reduce every rank must hold the same buffer whose
    # checksum is the sum of the ranks' checksums
    exchange_check = None
    if world > 1:
        sync_all()
        xchg.reset()
        xchg.hold_back, xchg.checksums = True, torch.zeros(2, dtype=torch.float64, device=dev)
        for i in range(exch_every):
            step(i)
        sync_all()
        chk = xchg.checksums
        xchg.hold_back, xchg.checksums = False, None
        xchg.reset()
        grad_acc = xchg.acc
        local = torch.stack([grad_acc.sum(dtype=torch.float64), grad_acc.abs().sum(dtype=torch.float64)])
        acc_ok = bool((local[0] - chk[0]).abs() <= 1e-6 * chk[1] + 1e-30) and bool(chk[1] > 0)
        dist.all_reduce(grad_acc)
        glob = grad_acc.sum(dtype=torch.float64).reshape(1)
        want = torch.stack([local[0], local[1]])
        dist.all_reduce(want)                                  # sum over the ranks of (checksum, sum of magnitudes)
        lo, hi = glob.clone(), glob.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        flags = torch.tensor([float(acc_ok), float(bool((glob - want[0]).abs() <= 1e-6 * want[1])), float(bool(lo == hi))], device=dev)
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        exchange_check = {"ok": bool(flags.min().item() == 1.0), "accumulator_equals_the_packed_views": bool(flags[0].item() == 1.0),
                          "allreduce_checksum_is_the_sum_of_the_ranks": bool(flags[1].item() == 1.0),
                          "ranks_hold_the_same_buffer_checksum": bool(flags[2].item() == 1.0), "views_per_rank_checked": exch_every,
                          "sum_of_magnitudes": float(want[1].item()), "allreduces_in_the_run": xchg.reduced}

    # instance count of the views this rank rendered (data-dependent: read back from one extra untimed pass)
    from animatablegaussians_amd.rasterizer import native_rasterize_gaussians
    empty = torch.Tensor([])
    # ... and its (pixel, entry) PAIR EVALUATIONS: the iterations of the reference's per-pixel loops (renderCUDA forward.cu:310-369 runs a pixel
    # until it is done, the backward backward.cu:494-600 from its last contributor back) = the sum over the pixels of n_contrib, read from the
    # forward's own per-pixel counters (the blend kernels are reported against HBM AND as pair evaluations per second)
    import ctypes
    pairs_seen = []
    lay = _lib.AgRasterScratchLayout()
    with torch.no_grad():
        for v in range(len(settings)):
            s = settings[v]
            res = native_rasterize_gaussians(s.bg, means3D, colors, opacities, scales, rotations, 1.0, empty,
                                             s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, H, W, empty, 0,
                                             s.campos, False, False)
            R_seen.append(res[0])
            _lib.check(_lib.lib().ag_raster_describe_scratch(P, W, H, int(res[0]), ctypes.byref(lay)), "describe")
            img = res[7]
            start = ((img.data_ptr() + 255) & ~255) - img.data_ptr() + lay.img_n_contrib_off
            pairs_seen.append(int(img[start:start + W * H * 4].view(torch.int32).sum(dtype=torch.int64).item()))
    views_of_rank = [(i * world + rank) % len(settings) for i in range(args.warmup, args.warmup + args.steps)]
    R_mean = float(np.mean([R_seen[v] for v in views_of_rank]))
    pairs_mean = float(np.mean([pairs_seen[v] for v in views_of_rank]))

    breakdown = None
    if rank == 0 and (args.breakdown or world == 1):      # round 5: always at N = 1 (64 one-stream steps = ~20 ms), so that the driver's record carries
        _lib.prof_enable(range(_lib.AG_K_COUNT))          # every raster kernel's fraction (`roofline_raster_kernels`), not the dominant one's only
        if xchg is not None:
            xchg.hold_back = True                         # rank 0 alone runs this pass: no collective in it
        for i in range(64 if not args.breakdown else min(args.steps, 64)):
            step_on(i)            # one stream: per-kernel times without the overlap of the pipelined timed region
        sync_all() if world == 1 else torch.cuda.synchronize(dev)
        bd = _lib.prof_collect()
        _lib.prof_enable([])
        breakdown = {k: round(1e3 * ms / max(n, 1), 2) for k, (n, ms) in bd.items()}
        if xchg is not None:
            xchg.hold_back = False

    # N > 1: the exchange north_star names for the full training step -- the bucketed all-reduce of the three networks' gradients
    # (223.6 M fp32 = 895 MB, parallel.BucketedGradSync's 128-MB buckets) -- timed on its own after the headline region, so the
    # scaling runs record what RCCL over xGMI delivers for it even though the headline workload is raster-only
    exchange_leg = None
    if world > 1:
        n_el = 223648936
        buf = torch.zeros(n_el, device=dev)
        cap = (128 << 20) // 4
        cuts = [(o, min(n_el, o + cap)) for o in range(0, n_el, cap)]

        def allreduce_all():
            works = [dist.all_reduce(buf[a:b], async_op=True) for a, b in cuts]
            for w in works:
                w.wait()

        for _ in range(2):
            allreduce_all()
        torch.cuda.synchronize(dev)
        dist.barrier()
        t2 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            allreduce_all()
        torch.cuda.synchronize(dev)
        dt = torch.tensor([(time.perf_counter() - t2) / reps], device=dev, dtype=torch.float64)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        ms = 1e3 * float(dt.item())
        # per-rank times (round 6: the first 8-GPU run must yield the whole curve in one command): every rank's own wall time of the same collective
        mine = torch.tensor([(time.perf_counter() - t2) / reps * 1e3], device=dev, dtype=torch.float64)
        per_rank = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        exchange_leg = {"what": "bucketed all-reduce of the StyleUNet gradients (223.6 M fp32, 128-MB buckets), not part of `value`",
                    "bytes": 4 * n_el, "ms": round(ms, 3), "bus_GBps": round(2 * (world - 1) / world * 4 * n_el / (ms * 1e-3) / 1e9, 1),
                    "ms_per_rank": [round(float(t.item()), 3) for t in per_rank]}
        del buf
        # ... and the headline's own exchange on its own: the all-reduce of the per-Gaussian gradient sums (P x 14 fp32), back to back
        vb = torch.zeros(P * 14, device=dev)
        for _ in range(3):
            dist.all_reduce(vb)
        torch.cuda.synchronize(dev)
        dist.barrier()
        t3 = time.perf_counter()
        vreps = 20
        for _ in range(vreps):
            dist.all_reduce(vb)
        torch.cuda.synchronize(dev)
        vmine = torch.tensor([(time.perf_counter() - t3) / vreps * 1e3], device=dev, dtype=torch.float64)
        vper = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(vper, vmine)
        vms = max(float(t.item()) for t in vper)
        exchange_leg["view_gradients"] = {"what": "all-reduce of the per-Gaussian attribute gradient sums (the headline's exchange), back to back, not part of `value`",
                                          "bytes": 4 * P * 14, "ms": round(vms, 4), "ms_per_rank": [round(float(t.item()), 4) for t in vper],
                                          "bus_GBps": round(2 * (world - 1) / world * 4 * P * 14 / (vms * 1e-3) / 1e9, 1),
                                          "views_per_exchange_per_rank": exch_every}
        del vb

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    T_tiles = ((W + 15) // 16) * ((H + 15) // 16)
    n_dom, ms_dom = prof["blend_backward_kernel"]
    dom_us = 1e3 * ms_dom / max(n_dom, 1)
    alg_dom = 8 * T_tiles + 44 * R_mean + 28 * W * H + 40 * P          # bytes per launch (bwd blend)
    alg_step = 392 * P + 132 * R_mean + 52 * W * H + 24 * T_tiles       # whole raster fwd+bwd
    achieved = alg_dom / (dom_us * 1e-6) / 1e9 if dom_us > 0 else 0.0
    value = args.gpus * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # HBM traffic of the dominant kernel comes from PMC counters, which cannot be read from inside this process: the separate
    # rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over the same workload are summarised in profiles/traffic_head.json
    # (profiles/traffic_probe.py + traffic_summarize.py, FETCH x 2.0 / WRITE x 1.0 from the calibration copy of that run); the line
    # quotes that file for the kernel it times -- bytes per launch, like `achieved` -- and says so; null when the file has no entry.
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_head.json")) as f:
            tj = json.load(f)
        hit = [v for k, v in tj.get("kernels", {}).items() if "blend_backward_wave_kernel" in k]
        if hit:
            traffic = int(hit[0]["hbm_bytes"])
            traffic_src = "profiles/traffic_head.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, same workload; not this run)"
    except (OSError, ValueError, KeyError):
        pass

    # the same workload with dependent steps on ONE stream (what a sequential trainer sees), next to the pipelined headline
    seq = None
    if world == 1 and args.streams > 1:
        n_seq = max(50, min(args.steps, 400))
        for i in range(20):
            step_on(i)
        sync_all()
        _lib.prof_enable([DOM])
        t1 = time.perf_counter()
        for i in range(n_seq):
            step_on(i)
        sync_all()
        dt_seq = time.perf_counter() - t1
        pseq = _lib.prof_collect()
        _lib.prof_enable([])
        n1, ms1 = pseq["blend_backward_kernel"]
        seq = {"views_per_s": round(n_seq / dt_seq, 1), "steps": n_seq,
               "blend_backward_avg_launch_us": round(1e3 * ms1 / max(n1, 1), 2),
               "note": "one HIP stream, every step ordered after the previous one; the kernel duration here is the kernel alone (in the "
                       "headline region several views share the GPU and every kernel's wall duration stretches)"}
    # ... and through the reference's operator surface only: GaussianRasterizer forward + public torch.autograd.backward
    oper = None
    if world == 1 and not args.operator_path:
        n_op = max(50, min(args.steps, 400))
        for i in range(20):
            step_operator(i)
        sync_all()
        t1 = time.perf_counter()
        for i in range(n_op):
            step_operator(i)
        sync_all()
        oper = {"views_per_s": round(n_op / (time.perf_counter() - t1), 1), "steps": n_op,
                "note": "GaussianRasterizer (autograd.Function) forward + torch.autograd.backward with explicit image gradients, same "
                        "kernels; the difference to `value` is host time of the autograd machinery"}

    out = {
        "metric": "rendered views/sec (fwd+bwd) @1024^2, ~250k Gaussians",
        "value": round(value, 2), "unit": "views/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "value_blocks": {"views_per_s": [round(args.gpus * args.steps / t, 1) for t in value_blocks],
                         "note": f"five more blocks of {args.steps} steps right after the timed region, each timed the same way (max over ranks): "
                                 "how far `value` moves from block to block at this region length"},
        "config": {
            "workload": "BASELINE configs[1]: avatar front|back map, 1 view @1024x1024 per step, rasterizer fwd+bwd "
                        + ("through GaussianRasterizer + torch.autograd" if args.operator_path else
                           "through the library-owned step (ag_raster_forward_backward, one native call per view; per-camera argument structures and "
                           "output images prepared once, FusedRasterStep.prepare / run)"),
            "gaussians": P, "instances_per_view": int(R_mean), "tiles": T_tiles, "views": len(settings), "streams": args.streams,
            "prewarm": max(0, args.prewarm),      # untimed steps BEFORE the --warmup steps (round 4: the first region of a process is cold)
            "parallelism": "1 process" if world == 1 else
                           f"view-sharded x{world}: every rank renders its own view per step and sums its views' per-Gaussian gradients (14 f32 each, "
                           f"{P * 14 * 4 / 1e6:.1f} MB) on the device; one RCCL all-reduce of the sums per {exch_every} steps of a rank "
                           f"(= a {exch_every * world}-view iteration), on a communication stream under the next views' kernels",
            "exchange_every_steps": exch_every if world > 1 else None,
            "backend": None if world == 1 else (backend if backend == "nccl" else f"{backend}: fewer GPUs than ranks, ranks share devices -- a "
                                                "functional run of the N > 1 control flow, NOT a measurement"),
        },
        "roofline": {
            "kernel": "blend_backward_wave_kernel",
            "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": int(alg_dom), "avg_launch_us": round(dom_us, 2), "launches_timed": n_dom,
            "whole_step_algorithmic_GBps": round(alg_step / (ms_per_step * 1e-3) / 1e9, 2),
            "note": "VALU/LDS/atomic-bound kernel reported against HBM as SURVEY.md 8(d) prescribes; `valu` = the yardstick that fits it",
            "valu": valu_roofline(pairs_mean, dom_us, "backward"),
        },
    }
    if breakdown is not None:
        # algorithmic bytes per launch of every raster kernel (DESIGN.md section 4 / SURVEY.md 8d) against its one-stream duration
        alg = {"preprocess_kernel": 112 * P, "tile_scan_kernel": 8 * T_tiles, "scatter_kernel": 20 * P + 12 * R_mean,
               "tile_sort_kernel": 24 * R_mean, "blend_forward_kernel": 8 * T_tiles + 44 * R_mean + 24 * W * H,
               "blend_backward_kernel": alg_dom, "preprocess_backward_kernel": 220 * P}
        pmc = {}
        try:
            import json as _json
            with open(os.path.join(ROOT, "profiles", "traffic_head.json")) as f:
                tj = _json.load(f).get("kernels", {})
            frag = {"preprocess_kernel": "::preprocess_kernel", "tile_scan_kernel": "tile_scan_kernel", "scatter_kernel": "scatter_kernel",
                    "tile_sort_kernel": "tile_sort_kernel", "blend_forward_kernel": "blend_forward_kernel",
                    "blend_backward_kernel": "blend_backward_wave_kernel", "preprocess_backward_kernel": "preprocess_backward_kernel"}
            for k, fr in frag.items():
                hit = [v["hbm_bytes"] for kk, v in tj.items() if fr in kk]
                if hit:
                    pmc[k] = float(sum(hit))
        except (OSError, ValueError, KeyError):
            pass
        rk = {"what": "every raster kernel of ONE view on ONE stream (64 dependent steps, HIP events around every launch): average launch, "
                      "SURVEY 8(d) algorithmic bytes, achieved = bytes / time against the 8 TB/s HBM peak; pmc_ratio = HBM bytes per launch from the "
                      "PMC counters (profiles/traffic_head.json, separate rocprofv3 --pmc passes, not this run) / algorithmic bytes",
              "one_stream_sum_us": round(sum(breakdown.get(k, 0.0) for k in alg), 2)}
        for k in alg:
            us = breakdown.get(k, 0.0)
            if us > 0:
                gbs = alg[k] / (us * 1e-6) / 1e9
                rk[k] = {"avg_launch_us": us, "algorithmic_MB": round(alg[k] / 1e6, 2), "achieved_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                         "pmc_ratio": round(pmc[k] / alg[k], 2) if k in pmc else None}
        out["roofline_raster_kernels"] = rk
    if breakdown is not None and args.breakdown:
        out["kernels_us"] = breakdown
        out["kernels_algorithmic_GBps"] = {k: round(alg[k] / (breakdown[k] * 1e-6) / 1e9, 1) for k in alg if breakdown.get(k, 0) > 0}
        out["kernels_algorithmic_MB"] = {k: round(alg[k] / 1e6, 2) for k in alg}
        # what a pure streaming kernel achieves at the same footprints on this box: a device copy moving as many bytes (half read, half
        # written) as preprocess / scatter / preprocess backward do -- the ceiling a launch of that size can reach, fixed costs included
        ceil = {}
        for k in ("preprocess_kernel", "scatter_kernel", "preprocess_backward_kernel"):
            n = int(alg[k] // 8)
            a, b = torch.empty(n, device=dev), torch.empty(n, device=dev)
            for _ in range(3):
                b.copy_(a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            us = []
            for _ in range(20):
                e0.record()
                b.copy_(a)
                e1.record()
                e1.synchronize()
                us.append(1e3 * e0.elapsed_time(e1))
            ceil[k] = {"copy_us": round(float(np.median(us)), 2), "copy_GBps": round(alg[k] / (float(np.median(us)) * 1e-6) / 1e9, 1)}
        out["streaming_copy_of_the_same_bytes"] = ceil
    if seq is not None:
        out["sequential"] = seq
        us1 = seq["blend_backward_avg_launch_us"]
        if us1 > 0:
            out["roofline"]["avg_launch_us_one_stream"] = us1
            out["roofline"]["frac_one_stream"] = round(alg_dom / (us1 * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
            out["roofline"]["valu_one_stream"] = valu_roofline(float(np.mean(pairs_seen)), us1, "backward")
    if breakdown is not None and breakdown.get("blend_forward_kernel", 0) > 0:
        out["roofline_raster_kernels"]["blend_forward_kernel"]["valu"] = valu_roofline(float(np.mean(pairs_seen)), breakdown["blend_forward_kernel"], "forward")
        out["roofline_raster_kernels"]["blend_backward_kernel"]["valu"] = valu_roofline(float(np.mean(pairs_seen)), breakdown["blend_backward_kernel"], "backward")
    if oper is not None:
        out["operator_path"] = oper
    if exchange_leg is not None:
        out["exchange_styleunet"] = exchange_leg
    if exchange_check is not None:
        out["exchange_check"] = exchange_check
    if world == 1 and not args.no_full_step:
        # BASELINE configs[2] and the MFMA roofline north_star asks for, measured in this process (about 15 s): the whole training
        # iteration at the reference's batch shape (1 view per step) and at 4 views of one pose per step, and the convolution kernels'
        # own rate from HIP events around every launch of one network forward + backward
        import bench_avatar
        torch.autograd.set_multithreading_enabled(True)      # the networks' six-stream backward uses autograd's worker threads
        for leaf in leaves:
            leaf.grad = None
        torch.cuda.empty_cache()
        out["roofline_mfma"] = bench_avatar.conv_roofline(dev)
        out["roofline_avatar_kernels"] = bench_avatar.avatar_kernel_rooflines(dev)
        out["full_step"] = bench_avatar.full_step_probe(dev)

    if world == 1 and not args.no_stress:
        for leaf in leaves:
            leaf.grad = None
        del fused
        torch.cuda.empty_cache()
        out["stress_1m_2048"] = stress_1m_2048(dev)
    if not args.no_cpu_baseline and world == 1:            # reported at N = 1 only (bounded samples: ~25 s + ~20 s of host time)
        out["cpu_baseline"] = cpu_baseline(av, cams_np, up, W, H)
        out["cpu_baseline_lbs"] = cpu_baseline_lbs()
        out["cpu_baseline_styleunet"] = cpu_baseline_styleunet()
    out["headline"] = headline_of(out)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


VALU_PEAK_LANE_OPS = 78.6e12   # MI355X_MICROARCH.md: 157.3 TFLOP/s fp32 vector = 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz fused multiply-adds
# lane operations of the REFERENCE's loop bodies, counted from its source: every evaluated pair pays the contributor test, d, power, exp, alpha and
# the two cut-offs (forward.cu:325-343 / backward.cu:513-531: 18); a pair that passes them pays the blend (forward.cu:345-362: 14) or the gradient
# chain (backward.cu:533-598, the two divisions at ~8 each: 85).  Share of pairs that pass: measured by the diagnostic build on bench views 0 / 2 / 5
# (profiles/bwd_step_stats.py, profiles/r05b_bwd_step_stats_tight.txt against the views' n_contrib sums): 0.089 / 0.056 / 0.094 -> 0.09
REF_OPS_TEST, REF_OPS_FWD_ACTIVE, REF_OPS_BWD_ACTIVE, ACTIVE_SHARE = 18.0, 14.0, 85.0, 0.09


def valu_roofline(pairs: float, us: float, which: str):
    """the two blend kernels as (pixel, entry) pair evaluations per second, and against the VALU floor of the reference's own
    algorithm: lane operations of its loop body x the pairs it evaluates / the fp32 vector peak.  `frac` 1.0 would mean: as fast as the
    reference's loop could run if every lane of every SIMD did nothing but its arithmetic."""
    if us <= 0 or pairs <= 0:
        return None
    ops = pairs * (REF_OPS_TEST + ACTIVE_SHARE * (REF_OPS_BWD_ACTIVE if which == "backward" else REF_OPS_FWD_ACTIVE))
    return {"pair_evals_per_launch": int(pairs), "pair_evals_per_s": round(pairs / (us * 1e-6), 0), "G_pair_evals_per_s": round(pairs / (us * 1e-6) / 1e9, 1),
            "reference_lane_ops_per_launch": int(ops), "peak_lane_ops_per_s": VALU_PEAK_LANE_OPS,
            "frac": round(ops / (us * 1e-6) / VALU_PEAK_LANE_OPS, 4), "avg_launch_us": round(us, 2)}


def headline_of(out):
    """The numbers a reader of the LAST 2000 characters of the line needs (the driver keeps that much): short keys, no notes."""
    h = {"value_views_per_s": out["value"], "n_gpus": out["n_gpus"]}
    r = out.get("roofline", {})
    h["bwd_us_overlapped"] = r.get("avg_launch_us")
    h["bwd_frac_hbm"] = r.get("frac")
    h["bwd_us_one_stream"] = r.get("avg_launch_us_one_stream")
    h["bwd_frac_hbm_one_stream"] = r.get("frac_one_stream")
    v = r.get("valu_one_stream") or r.get("valu")
    if v:
        h["bwd_G_pair_evals_per_s"] = v["G_pair_evals_per_s"]
        h["bwd_frac_valu"] = v["frac"]
    rk = out.get("roofline_raster_kernels")
    if rk:
        h["raster_one_stream_sum_us"] = rk.get("one_stream_sum_us")
        h["raster_kernel_us"] = {k.replace("_kernel", ""): d["avg_launch_us"] for k, d in rk.items() if isinstance(d, dict) and "avg_launch_us" in d}
        h["raster_kernel_frac_hbm"] = {k.replace("_kernel", ""): d["frac"] for k, d in rk.items() if isinstance(d, dict) and "frac" in d}
        fv = rk.get("blend_forward_kernel", {}).get("valu")
        if fv:
            h["fwd_G_pair_evals_per_s"] = fv["G_pair_evals_per_s"]
            h["fwd_frac_valu"] = fv["frac"]
    if "sequential" in out:
        h["sequential_views_per_s"] = out["sequential"]["views_per_s"]
    if "operator_path" in out:
        h["operator_path_views_per_s"] = out["operator_path"]["views_per_s"]
    fs = out.get("full_step")
    if fs:
        h["full_step_math"] = fs.get("conv_math")
        v16 = fs.get("views16_one_pose_configs3_n1", {})
        h["full_step_views_per_s_1_4_16"] = [fs.get("views_per_s_1view_per_step"), fs.get("views_per_s_4views_per_step"), v16.get("views_per_s")]
        h["full_step_ms_1_4_16"] = [fs.get("ms_per_step_1view"), fs.get("ms_per_step_4views"), v16.get("ms_per_step")]
        if "inference_1view" in fs:
            h["inference_views_per_s"] = fs["inference_1view"].get("views_per_s")
    rm = out.get("roofline_mfma")
    if rm:
        h["mfma_achieved_TF_fp32_equiv"] = rm.get("achieved")
        h["mfma_frac_of_2.5PF_executed"] = rm.get("frac")
    st = out.get("stress_1m_2048")
    if st:
        h["stress_1m_2048_views_per_s"] = st.get("views_per_s")
    cb = out.get("cpu_baseline")
    if cb:
        h["cpu_baseline_views_per_s"] = cb.get("value")
        h["cpu_cores"] = cb.get("cores")
    x = out.get("exchange_styleunet")
    if x:
        h["exchange_styleunet_ms"] = x.get("ms")
        h["exchange_styleunet_bus_GBps"] = x.get("bus_GBps")
    return h


def stress_1m_2048(dev, steps: int = 60, warmup: int = 12):
    """BASELINE configs[4] on ONE GPU: 1.07 M Gaussians (the 2048 x 4096 front|back canvas), one 2048^2 view per step, f = 2200, the same
    library-owned forward + backward step as the headline on three internal streams; blend-backward duration from HIP events on its
    launch stream in a one-stream pass.  Parity at this size: tests/test_raster_gpu.py::test_full_size_1m_gaussians_2048_vs_oracle."""
    import numpy as np
    import torch
    from animatablegaussians_amd import _lib, camera, synth
    from animatablegaussians_amd.rasterizer import FusedRasterStep, GaussianRasterizationSettings, native_rasterize_gaussians
    S = 2048
    av = synth.avatar_map_gaussians(S)
    P = av["means3D"].shape[0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    g = [t(av[k]) for k in ("means3D", "colors", "opacities", "scales", "rotations")]
    bg = t(av["bg"])
    settings = []
    for c in synth.free_view_cameras(8, img=S, focal=2200.0):
        cm = camera.camera_from_intr_extr(c["extr"], c["intr"], S, S)
        settings.append(GaussianRasterizationSettings(image_height=S, image_width=S, tanfovx=cm["tanfovx"], tanfovy=cm["tanfovy"], bg=bg,
                                                      scale_modifier=1.0, viewmatrix=t(cm["viewmatrix"]), projmatrix=t(cm["projmatrix"]),
                                                      sh_degree=0, campos=t(cm["campos"]), prefiltered=False, debug=False))
    up = synth.upstream_grads(S, S, 999)
    gc, gd, ga = t(up["dL_dcolor"]), t(up["dL_ddepth"]), t(up["dL_dalpha"])
    fused = FusedRasterStep(P, S, S, dev, n_streams=3)

    def step(i, slot=None):
        fused.view(settings[i % 8], g[0], g[1], g[2], g[3], g[4], gc, gd, ga, slot=(i % 3) if slot is None else slot)

    def sync():
        fused.join()
        torch.cuda.synchronize(dev)

    for i in range(warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    sync()
    dt = time.perf_counter() - t0
    _lib.prof_enable([5])
    for i in range(24):
        step(i, slot=0)
    sync()
    n, ms = _lib.prof_collect()["blend_backward_kernel"]
    _lib.prof_enable([])
    empty = torch.Tensor([])
    with torch.no_grad():
        R = [native_rasterize_gaussians(s.bg, g[0], g[1], g[2], g[3], g[4], 1.0, empty, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, S, S,
                                        empty, 0, s.campos, False, False)[0] for s in settings]
    R_mean = float(np.mean(R))
    T_tiles = (S // 16) ** 2
    us = 1e3 * ms / max(n, 1)
    alg_dom = 8 * T_tiles + 44 * R_mean + 28 * S * S + 40 * P
    alg_step = 392 * P + 132 * R_mean + 52 * S * S + 24 * T_tiles
    return {"workload": "BASELINE configs[4] on one GPU: 2048 x 4096 canvas, 1 view @2048^2 per step, raster fwd+bwd (library-owned step, 3 streams)",
            "gaussians": P, "instances_per_view": int(R_mean), "steps": steps, "views_per_s": round(steps / dt, 1),
            "ms_per_step": round(1e3 * dt / steps, 4), "blend_backward_avg_launch_us": round(us, 1),
            "blend_backward_algorithmic_GBps": round(alg_dom / (us * 1e-6) / 1e9, 1), "blend_backward_frac_of_hbm": round(alg_dom / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "whole_step_algorithmic_GBps": round(alg_step / (dt / steps) / 1e9, 1)}


def _best_cpu_threads():
    """os.cpu_count() counts hardware threads the container may not own (256 on the pool's boxes, where a 256-thread oneDNN convolution
    runs 30x SLOWER than an 8-thread one): time one mid-sized convolution at a few thread counts and keep the fastest."""
    import torch
    import torch.nn.functional as F
    x, w = torch.randn(1, 256, 64, 64), torch.randn(256, 256, 3, 3)
    best, best_t = 1, float("inf")
    for n in (8, 16, 32, 64, 128):
        if n > (os.cpu_count() or 1):
            break
        torch.set_num_threads(n)
        F.conv2d(x, w, padding=1)
        t0 = time.perf_counter()
        for _ in range(3):
            F.conv2d(x, w, padding=1)
        dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = n, dt
    torch.set_num_threads(best)
    return best


def _cpu_lbs_worker(reps: int = 5):
    import numpy as np
    import torch
    from animatablegaussians_amd import synth
    from oracle import avatar_oracle as ao
    threads = _best_cpu_threads()
    av = synth.avatar_map_gaussians()
    N, J = av["means3D"].shape[0], 55
    gen = torch.Generator().manual_seed(3)
    lbs = torch.softmax(torch.randn(N, J, generator=gen) * 4, dim=1)
    A = torch.eye(4)[None].repeat(J, 1, 1) + 0.01 * torch.randn(J, 4, 4, generator=gen)
    pos = torch.from_numpy(np.ascontiguousarray(av["means3D"])).requires_grad_(True)
    rot = torch.nn.functional.normalize(torch.randn(N, 4, generator=gen)).requires_grad_(True)
    fwd, both = [], []
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        p2, r2 = ao.transform_cano2live(pos, rot, lbs, A)
        t1 = time.perf_counter()
        (p2.sum() + r2.sum()).backward()
        t2 = time.perf_counter()
        pos.grad = rot.grad = None
        fwd.append(t1 - t0)
        both.append(t2 - t0)
    print(json.dumps({"forward_ms": round(1e3 * float(np.median(fwd[1:])), 2), "forward_backward_ms": round(1e3 * float(np.median(both[1:])), 2),
                      "gaussians": N, "joints": J, "threads": threads}), flush=True)


def _cpu_styleunet_worker():
    import numpy as np
    import torch
    from animatablegaussians_amd import synth
    from animatablegaussians_amd.styleunet import DualStyleUNet
    from oracle.dual_styleunet_oracle import DualStyleUNetOracle
    threads = _best_cpu_threads()
    print(json.dumps({"threads": threads}), flush=True)
    shapes = {k: v.shape for k, v in DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2).reference_state_dict().items()}
    sd = synth.named_fill({k: torch.empty(s) for k, s in shapes.items()})
    for k, v in sd.items():
        if not k.startswith("noises."):
            v.requires_grad_(True)
    pose = synth.pose_map(512).requires_grad_(True)
    style = torch.ones(1, 512) / np.sqrt(512)
    net = DualStyleUNetOracle(sd)
    with torch.no_grad():
        t0 = time.perf_counter()
        net.forward(style, pose)
        print(json.dumps({"forward_s": round(time.perf_counter() - t0, 2)}), flush=True)
    t0 = time.perf_counter()
    images = net.forward(style, pose)
    images.square().mean().backward()
    print(json.dumps({"forward_backward_s": round(time.perf_counter() - t0, 2)}), flush=True)


def _run_cpu_worker(which: str, timeout_s: float):
    """The torch-CPU baselines run in a child process with a hard time limit: a host whose visible cores are not really its own must not
    stall the bench line (first run on the pool: 514 s for one network pass with 256 threads).  Returns the merged JSON records the child
    printed before the limit."""
    import subprocess
    rec = {}
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--_cpu-worker", which], capture_output=True, text=True, timeout=timeout_s,
                           env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
        out = p.stdout
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        rec["timed_out_after_s"] = timeout_s
    for ln in out.splitlines():
        if ln.startswith("{"):
            rec.update(json.loads(ln))
    return rec


def cpu_baseline_lbs():
    """the reference's LBS path (network/avatar.py:84-91: einsum over the [N, 55] weights, pytorch3d quaternion <-> matrix)
    as restated in oracle/avatar_oracle.py, torch CPU; the synthetic subject's 268 348 Gaussians."""
    r = _run_cpu_worker("lbs", 60.0)
    r.update({"cores": os.cpu_count() or 1, "kind": "port",
              "sample": "median of 5 calls after one warm-up; torch CPU ops, dense [N, 55] weights as the reference stores them; thread count = the "
                        "fastest of 8..128 on a calibration convolution (`threads`)"})
    return r


def cpu_baseline_styleunet(timeout_s: float = 75.0):
    """the reference's DualStyleUNet (512 -> 1024, the colour / position configuration) on this box's host cores: one
    forward and one forward + backward of oracle/dual_styleunet_oracle.py -- F.conv2d / F.conv_transpose2d and the reference's pure-torch
    upfirdn2d / fused_leaky_relu branches in the reference's order (pinned against the reference module's own golden by
    tests/test_styleunet_oracle_cpu.py).  One pass each inside a child process with a hard limit; what did not finish is reported as such."""
    r = _run_cpu_worker("styleunet", timeout_s)
    if "forward_backward_s" in r:
        r["views_per_s_equivalent_3_networks"] = round(1.0 / (3 * r["forward_backward_s"]), 4)
    r.update({"networks": 1, "GFLOP_forward": 585.8, "cores": os.cpu_count() or 1, "kind": "port",
              "sample": "one DualStyleUNet forward (no grad) and one forward + backward, fp32, torch CPU (oneDNN), `threads` = the fastest of 8..128 "
                        f"on a calibration convolution; child process limited to {timeout_s:.0f} s; a training step of the reference evaluates three "
                        "of these per view"})
    return r


def cpu_baseline(av, cams_np, up, W, H, max_seconds: float = 25.0):
    """The CPU oracle (oracle/raster_oracle.c: scalar C restatement of forward.cu / backward.cu, OpenMP over tiles
    for the two blend loops) on the same scene and upstream gradients, views 0.. until ~max_seconds are spent."""
    import numpy as np
    from animatablegaussians_amd import camera
    from oracle import raster_oracle as ro
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    ro.lib()
    done, t0 = 0, time.perf_counter()
    for c in cams_np:
        cm = camera.camera_from_intr_extr(c["extr"], c["intr"], W, H)
        st = ro.forward(av["means3D"], av["colors"], av["opacities"], av["scales"], av["rotations"], av["bg"],
                        cm["viewmatrix"], cm["projmatrix"], cm["tanfovx"], cm["tanfovy"], W, H, want_fragile=False)
        ro.backward(st, av["means3D"], av["colors"], av["scales"], av["rotations"], av["bg"], cm["viewmatrix"],
                    cm["projmatrix"], cm["tanfovx"], cm["tanfovy"], up["dL_dcolor"], up["dL_ddepth"], up["dL_dalpha"],
                    f32_accum=True)
        done += 1
        if time.perf_counter() - t0 > max_seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 4), "unit": "views/s", "cores": cores, "kind": "port",
            "sample": f"{done} of the 8 views of the same scene, fwd+bwd, {dt:.1f} s wall; per-Gaussian stages "
                      f"single-threaded, blend loops OpenMP x{cores}"}


if __name__ == "__main__":
    main()
