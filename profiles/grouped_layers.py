"""Per-layer time of the grouped StyleUNet chain (round 4):  python profiles/grouped_layers.py [out.csv]
Records every grouped ConvLayer / StyledConv / ToRGB call of one AvatarNet.get_maps (shape, group size), then times each distinct call
forward and backward on its own (HIP events, median of 5) and prints time, conv FLOPs and TFLOP/s (algorithmic fp32 FLOPs: forward 1x,
backward 2x = input gradient + weight gradient)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import grouped as gr  # noqa: E402
from animatablegaussians_amd.avatar import AvatarNet  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(31359)
net = AvatarNet.synthetic({'with_viewdirs': True}, device=dev)
calls = []
orig_layer, orig_rgb, orig_comb = gr._GroupedLayer.apply, gr._GroupedToRGB.apply, gr._GroupedComb.apply


def rec_layer(G, shared, resample, modulated, scale, k_blur, x, *params):
    w = params[0]
    calls.append(("styled" if modulated else "conv", G, int(w.shape[-3]), int(w.shape[-4]), int(x.shape[2]), int(w.shape[-1]), bool(resample), bool(shared)))
    return orig_layer(G, shared, resample, modulated, scale, k_blur, x, *params)


def rec_rgb(runs, scale, k_up, x, *rest):
    R, M = len(runs), int(x.shape[0])
    for r, (s0, e0) in enumerate(runs):
        w = rest[R + s0]
        calls.append(("torgb", e0 - s0, int(w.shape[-3]), int(w.shape[-4]), int(x.shape[2]), 1, rest[r] is not None, False))
    return orig_rgb(runs, scale, k_up, x, *rest)


def rec_comb(begin, scale, x, lev, *rest):
    # kind, members, C1, Cout, H, networks (in the k slot), C2 (in the resample slot)
    calls.append(("comb", int(x.shape[0]), int(x.shape[1]), int(rest[0].shape[0]), int(x.shape[2]), int(lev.shape[0]), int(lev.shape[1]), tuple(begin)))
    return orig_comb(begin, scale, x, lev, *rest)


gr._GroupedLayer.apply, gr._GroupedToRGB.apply, gr._GroupedComb.apply = rec_layer, rec_rgb, rec_comb
pose = torch.randn(3, 512, 512, device=dev)
vf = torch.randn(1, 128, 128, 128, device=dev)
with torch.no_grad():
    net.get_maps(pose, vf, vf)
gr._GroupedLayer.apply, gr._GroupedToRGB.apply, gr._GroupedComb.apply = orig_layer, orig_rgb, orig_comb
kb = net.position_net._k_blur
kbu = net.position_net._k_blur_up


def timed(fn, reps=5):
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts[1:]))


rows = []
seen = {}
for c in calls:
    seen[c] = seen.get(c, 0) + 1
g = torch.Generator().manual_seed(0)
for c, count in seen.items():
    kind, G, Cin, Cout, H, k, res, shared = c
    x = torch.randn(G if (kind == "comb" or not shared) else 1, Cin, H, H, device=dev).requires_grad_(not (shared is True))
    if kind == "comb":
        N, C2, begin = k, res, shared
        OH = H
        lev = torch.randn(N, C2, H, H, device=dev).requires_grad_(True)
        ws = [torch.randn(Cout, Cin + C2, 3, 3, device=dev).requires_grad_(True) for _ in range(N)]
        bs = [torch.zeros(Cout, device=dev).requires_grad_(True) for _ in range(N)]
        net_of = [r for r in range(N) for _ in range(begin[r + 1] - begin[r])]
        fn = lambda: gr._GroupedComb.apply(begin, 1 / ((Cin + C2) * 9) ** 0.5, x, lev, *ws, *[bs[r] for r in net_of])      # noqa: E731
    elif kind == "conv":
        ws = [torch.randn(Cout, Cin, k, k, device=dev).requires_grad_(True) for _ in range(G)]
        bs = [torch.zeros(Cout, device=dev).requires_grad_(True) for _ in range(G)]
        fn = lambda: gr.grouped_conv_layer(x, ws, bs, kb, 1 / (Cin * k * k) ** 0.5, res, shared)      # noqa: E731
        OH = H // 2 if res else H
    elif kind == "styled":
        OH = 2 * H if res else H
        ws = [torch.randn(1, Cout, Cin, k, k, device=dev).requires_grad_(True) for _ in range(G)]
        st = [torch.ones(1, Cin, device=dev).requires_grad_(True) for _ in range(G)]
        nz = [torch.randn(1, 1, OH, OH, device=dev) for _ in range(G)]
        nw = [torch.zeros(1, device=dev).requires_grad_(True) for _ in range(G)]
        bs = [torch.zeros(Cout, device=dev).requires_grad_(True) for _ in range(G)]
        fn = lambda: gr.grouped_styled_conv(x, ws, st, nz, nw, bs, kbu if res else None, 1 / (Cin * k * k) ** 0.5, res)      # noqa: E731
    else:
        OH = H
        ws = [torch.randn(1, Cout, Cin, 1, 1, device=dev).requires_grad_(True) for _ in range(G)]
        st = [torch.ones(1, Cin, device=dev).requires_grad_(True) for _ in range(G)]
        bs = [torch.zeros(Cout, device=dev).requires_grad_(True) for _ in range(G)]
        sk = torch.randn(G, Cout, H // 2, H // 2, device=dev).requires_grad_(True) if res else None
        fn = lambda: gr.grouped_to_rgb(x, ws, st, bs, sk, kbu, 1 / Cin ** 0.5)      # noqa: E731
    if kind == "comb":
        flop = 2.0 * Cout * 9 * H * H * (G * Cin + k * res)
    elif kind == "styled" and res:
        flop = 2.0 * G * Cout * Cin * k * k * H * H              # transposed convolution: every INPUT pixel meets every tap
    else:
        flop = 2.0 * G * Cout * Cin * k * k * OH * OH
    with torch.no_grad():
        tf = timed(fn)
    out = fn()
    up = torch.randn_like(out)
    tb = timed(lambda: torch.autograd.grad(out, [t for t in [x] + ([lev] if kind == "comb" else []) + ws + bs if t.requires_grad], up, retain_graph=True,
                                           allow_unused=True))
    nbwd = 1.0 if shared is True else 2.0
    rows.append((kind, G, Cin, Cout, H, k, res, shared, count, tf, tb, flop / 1e9, flop / tf / 1e9, nbwd * flop / tb / 1e9))
rows.sort(key=lambda r: -(r[9] + r[10]) * r[8])
hdr = "kind,G,Cin,Cout,H,k(comb:networks),resample(comb:C2),shared(comb:member ranges),calls,fwd_ms,bwd_ms,fwd_GFLOP,fwd_TFLOPs,bwd_TFLOPs"
lines = [hdr] + [",".join(str(round(v, 3)) if isinstance(v, float) else str(v).replace(",", " ") for v in r) for r in rows]
tot_f = sum(r[9] * r[8] for r in rows)
tot_b = sum(r[10] * r[8] for r in rows)
lines.append(f"# total of the layer calls: forward {tot_f:.2f} ms, backward {tot_b:.2f} ms; conv FLOPs forward {sum(r[11] * r[8] for r in rows) / 1e3:.2f} TFLOP")
print("\n".join(lines))
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
