#!/usr/bin/env python
"""Golden fixture of the whole DualStyleUNet, produced by the REFERENCE'S OWN MODULE on CPU (build container only).

Imports /root/reference/network/styleunet/dual_styleunet.py (stub modules stand in for its two compiled extensions; CPU
tensors take the pure-torch branches fused_act.py:118-129 / upfirdn2d.py:186-227 and conv2d_gradfix's F.conv2d), fills
every learnable tensor and noise map with ``animatablegaussians_amd.synth.named_fill`` (a pure function of the
state_dict key), and runs forward + backward at the product size (512 -> 1024, style_dim 512, n_mlp 2,
network/avatar.py:34) TWICE: in float64 (the values stored) and in float32 (the reference as shipped).  For every
stored tensor the fixture also records ``err32`` = max |ref_fp32 - ref_fp64| over the stored sub-sample, normalised
by the tensor's max magnitude: the yardstick for "agrees with the reference as well as fp32 arithmetic allows".
~4 min of CPU.

    python tests/golden/make_golden_dual_styleunet.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.modules.setdefault("fused", types.ModuleType("fused"))
sys.modules.setdefault("upfirdn2d", types.ModuleType("upfirdn2d"))
sys.path.insert(0, "/root/reference")
from network.styleunet.dual_styleunet import DualStyleUNet  # noqa: E402  (reference code)

from animatablegaussians_amd import synth  # noqa: E402

OUT_CH = 3


def sub(t, n=256):
    f = t.detach().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].double().numpy().copy()


NBLK = 16


def full_stats(t):
    """Round 6: statistics over EVERY element of a gradient tensor (the 256-sample probe above covers 0.08 % of the 74 M gradient elements; a fault
    confined to one output-channel block of one layer can sit between its samples): sum, sum of magnitudes, sum of squares, and the sums / sums of
    magnitudes of NBLK contiguous blocks of the flattened tensor (dimension 0 -- output channels -- is the slowest), all accumulated in float64."""
    f = t.detach().double().flatten()
    n = f.numel()
    edges = [(n * b) // NBLK for b in range(NBLK + 1)]
    blk = np.array([f[edges[b]:edges[b + 1]].sum().item() for b in range(NBLK)])
    blkabs = np.array([f[edges[b]:edges[b + 1]].abs().sum().item() for b in range(NBLK)])
    return {"sum": f.sum().item(), "abs": f.abs().sum().item(), "sq": (f * f).sum().item(), "blk": blk, "blkabs": blkabs}


def run(dtype):
    torch.manual_seed(0)
    net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=OUT_CH, out_size=1024, style_dim=512, n_mlp=2)
    fill = synth.named_fill(net.state_dict())
    missing, unexpected = net.load_state_dict(fill, strict=False)
    assert not unexpected and all(m.endswith((".kernel", ".ll", ".lh", ".hl", ".hh")) for m in missing), (missing, unexpected)
    net = net.to(dtype)
    pose = synth.pose_map(512).to(dtype).requires_grad_(True)
    style = (torch.ones(1, 512) / np.sqrt(512)).to(dtype)          # network/avatar.py:38
    images, _ = net([style], pose, randomize_noise=False)
    G = torch.randn(images.shape, generator=torch.Generator().manual_seed(4242)).to(dtype)
    (images * G).sum().backward()
    img = images.detach()
    res = {
        "images_sub16": img[0, :, ::16, ::16].double().numpy().copy(),
        "images_crop_a": img[0, :, 500:532, 500:532].double().numpy().copy(),
        "images_crop_b": img[0, :, 100:132, 700:732].double().numpy().copy(),
        "pose_grad_sub8": pose.grad[0, :, ::8, ::8].double().numpy().copy(),
    }
    scal = {"images_absmean": img.abs().double().mean().item(), "images_max": img.abs().max().item(),
            "pose_grad_max": pose.grad.abs().max().item()}
    stats = {}
    for name, p in net.named_parameters():
        res["grad:" + name] = sub(p.grad)
        scal["gmax:" + name] = p.grad.abs().max().item()
        stats[name] = full_stats(p.grad)
    stats["@pose"] = full_stats(pose.grad)
    shapes = {"shape:" + n: np.asarray(p.shape, dtype=np.int64) for n, p in net.named_parameters()}
    shapes.update({"shape:" + n: np.asarray(b.shape, dtype=np.int64) for n, b in net.named_buffers() if n.startswith("noises.")})
    return res, scal, shapes, stats


r64, s64, shapes, st64 = run(torch.float64)
r32, s32, _, st32 = run(torch.float32)
out = dict(shapes)
# full-tensor statistics of the float64 run, and how far the reference's own float32 run is from them (the yardstick, as err32 above)
for name, a in st64.items():
    b = st32[name]
    out["fsum:" + name] = np.float64(a["sum"]); out["fabs:" + name] = np.float64(a["abs"]); out["fsq:" + name] = np.float64(a["sq"])
    out["fblk:" + name] = a["blk"]; out["fblkabs:" + name] = a["blkabs"]
    out["e32sum:" + name] = np.float64(abs(b["sum"] - a["sum"]) / max(a["abs"], 1e-300))
    out["e32sq:" + name] = np.float64(abs(b["sq"] - a["sq"]) / max(a["sq"], 1e-300))
    out["e32blk:" + name] = np.float64(np.max(np.abs(b["blk"] - a["blk"]) / np.maximum(a["blkabs"], 1e-300)))
for k, v in r64.items():
    out[k] = v.astype(np.float32)
    if k.startswith("images"):
        norm = s64["images_max"]
    elif k.startswith("pose"):
        norm = s64["pose_grad_max"]
    else:
        norm = s64["gmax:" + k[len("grad:"):]]
    out["err32:" + k] = np.float64(np.abs(r32[k] - v).max() / max(norm, 1e-300))
for k, v in s64.items():
    out[k] = np.float64(v)
out["err32:images_absmean"] = np.float64(abs(s32["images_absmean"] - s64["images_absmean"]) / s64["images_max"])
np.savez_compressed(os.path.join(HERE, "dual_styleunet_512_1024.npz"), **out)
errs = sorted(((float(out[k]), k) for k in out if k.startswith("err32:")), reverse=True)
print("wrote", len(out), "arrays; images absmean", s64["images_absmean"], "max", s64["images_max"])
print("largest reference fp32-vs-fp64 deviations:", errs[:8])
print("median:", errs[len(errs) // 2])
