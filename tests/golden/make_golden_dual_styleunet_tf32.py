#!/usr/bin/env python
"""How far is the reference's DualStyleUNet from its own float64 run in the arithmetic it actually uses on its own hardware?  (build container only)

The reference's convolutions go through cuDNN with ``torch.backends.cudnn.allow_tf32`` at its default True (network/styleunet/
conv2d_gradfix.py:185-189 passes the flag on; nothing in the repository clears it), i.e. on any Ampere-or-later GPU every convolution --
forward, input gradient, weight gradient -- multiplies operands ROUNDED to TF32 (10 explicit mantissa bits, cvt.rna: nearest, ties away)
and accumulates in fp32.  This script runs the REFERENCE MODULE on CPU in float32 with exactly that operand rounding emulated in its two
convolution entry points (everything else -- EqualLinear, FIR filters, activations -- stays fp32, as on the GPU, where matmul TF32 is off by
default) and stores, for every tensor of tests/golden/dual_styleunet_512_1024.npz, the deviation from the float64 golden there:
``errtf32:<key>``.  It is the yardstick for the opt-in AG_CONV_MATH_F16 mode (one fp16 part per operand: the same 11 significant bits).

    python tests/golden/make_golden_dual_styleunet_tf32.py          (~1 min of CPU)
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.modules.setdefault("fused", types.ModuleType("fused"))
sys.modules.setdefault("upfirdn2d", types.ModuleType("upfirdn2d"))
sys.path.insert(0, "/root/reference")
from network.styleunet import conv2d_gradfix  # noqa: E402  (reference code)
from network.styleunet.dual_styleunet import DualStyleUNet  # noqa: E402  (reference code)

from animatablegaussians_amd import synth  # noqa: E402


def rna_tf32(t):
    """float32 -> TF32 (10 explicit mantissa bits), round to nearest, ties away from zero (PTX cvt.rna.tf32.f32)"""
    b = t.detach().contiguous().view(torch.int32)
    return ((b + 0x1000) & ~0x1FFF).view(torch.float32)


class Tf32Conv(torch.autograd.Function):
    """conv2d / conv_transpose2d (groups 1) as cuDNN computes them with TF32 allowed: every one of the three convolutions rounds ITS two operands"""

    @staticmethod
    def forward(ctx, x, w, stride, padding, transpose):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, transpose)
        f = F.conv_transpose2d if transpose else F.conv2d
        return f(rna_tf32(x), rna_tf32(w), None, stride=stride, padding=padding)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, padding, transpose = ctx.cfg
        g = rna_tf32(gy)
        gx = gw = None
        if not transpose:
            if ctx.needs_input_grad[0]:
                gx = torch.nn.grad.conv2d_input(x.shape, rna_tf32(w), g, stride=stride, padding=padding)
            if ctx.needs_input_grad[1]:
                gw = torch.nn.grad.conv2d_weight(rna_tf32(x), w.shape, g, stride=stride, padding=padding)
        else:
            if ctx.needs_input_grad[0]:
                gx = F.conv2d(g, rna_tf32(w), None, stride=stride, padding=padding)
            if ctx.needs_input_grad[1]:
                gw = torch.nn.grad.conv2d_weight(g, w.shape, rna_tf32(x), stride=stride, padding=padding)
        return gx, gw, None, None, None


def _conv(transpose):
    def fn(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
        assert groups == 1 and dilation == 1 and output_padding == 0, (groups, dilation, output_padding)
        y = Tf32Conv.apply(input, weight, stride, padding, transpose)
        return y if bias is None else y + bias.view(1, -1, 1, 1)
    return fn


conv2d_gradfix.conv2d = lambda input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1: _conv(False)(input, weight, bias, stride, padding, 0, groups, dilation)
conv2d_gradfix.conv_transpose2d = _conv(True)


def sub(t, n=256):
    f = t.detach().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].double().numpy().copy()


gold = np.load(os.path.join(HERE, "dual_styleunet_512_1024.npz"))
torch.manual_seed(0)
net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2)
missing, unexpected = net.load_state_dict(synth.named_fill(net.state_dict()), strict=False)
assert not unexpected
pose = synth.pose_map(512).requires_grad_(True)
style = torch.ones(1, 512) / np.sqrt(512)
images, _ = net([style], pose, randomize_noise=False)
Gm = torch.randn(images.shape, generator=torch.Generator().manual_seed(4242))
(images * Gm).sum().backward()
img = images.detach()
res = {"images_sub16": img[0, :, ::16, ::16].double().numpy(), "images_crop_a": img[0, :, 500:532, 500:532].double().numpy(),
       "images_crop_b": img[0, :, 100:132, 700:732].double().numpy(), "pose_grad_sub8": pose.grad[0, :, ::8, ::8].double().numpy()}
for name, p in net.named_parameters():
    res["grad:" + name] = sub(p.grad)
out = {}
for k, v in res.items():
    norm = float(gold["images_max"]) if k.startswith("images") else float(gold["pose_grad_max"]) if k.startswith("pose") else float(gold["gmax:" + k[len("grad:"):]])
    out["errtf32:" + k] = np.float64(np.abs(v - gold[k]).max() / max(norm, 1e-300))
pd = np.abs(res["pose_grad_sub8"] - gold["pose_grad_sub8"]) / float(gold["pose_grad_max"])
out["errtf32:pose_percentiles_50_90_99_999"] = np.array([np.percentile(pd, q) for q in (50, 90, 99, 99.9)])
np.savez_compressed(os.path.join(HERE, "dual_styleunet_512_1024_tf32.npz"), **out)
errs = sorted(((float(v), k) for k, v in out.items() if np.ndim(v) == 0), reverse=True)
v = np.array([e for e, k in errs if k.startswith("errtf32:grad:") or k.startswith("errtf32:pose_grad")])
print("reference under emulated cuDNN-TF32 vs its float64 golden: forward", float(out["errtf32:images_sub16"]),
      "gradient rows p50/p90/p99/max", [float(np.percentile(v, q)) for q in (50, 90, 99, 100)])
print("largest:", errs[:6])
