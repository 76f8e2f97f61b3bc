"""CPU oracle (torch fp32) of the StyleUNet element-wise / FIR operators.  TEST INFRASTRUCTURE ONLY.

Restates the semantics of the reference's two native ops from their CUDA sources and CPU fall-backs:
  * fused_bias_act : network/styleunet/fused_bias_act_kernel.cu:18-65 (every (act, grad) case of the switch);
                     CPU branch of fused_leaky_relu, network/styleunet/fused_act.py:118-129
  * upfirdn2d      : network/styleunet/upfirdn2d.py:186-227 (upfirdn2d_native) == upfirdn2d_kernel.cu:49-105
Pinned by tests/golden/styleunet_ops.npz, which tests/golden/make_golden_styleunet.py produced by importing and running
the reference's own Python (CPU branches, autograd for the backward) in the build container.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def fused_bias_act(x, bias, ref, act: int, grad: int, alpha: float, scale: float):
    """Element-wise; x any shape [N, C, ...]; bias [C] or None; ref like x or None."""
    v = x
    if bias is not None and bias.numel():
        shape = [1, bias.numel()] + [1] * (x.dim() - 2)
        v = v + bias.view(*shape)
    r = ref if (ref is not None and ref.numel()) else torch.zeros_like(v)
    mode = act * 10 + grad
    if mode in (12, 32):
        y = torch.zeros_like(v)
    elif mode == 30:
        y = torch.where(v > 0, v, v * alpha)
    elif mode == 31:
        y = torch.where(r > 0, v, v * alpha)
    else:
        y = v
    return y * scale


def upfirdn2d(x, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """x [major, in_h, in_w] -> [major, out_h, out_w]."""
    major, in_h, in_w = x.shape
    kh, kw = kernel.shape
    # zero-insertion: sample (y, x) lands at (y*up_y, x*up_x) of an (in_h*up_y, in_w*up_x) grid
    z = x.new_zeros(major, in_h * up_y, in_w * up_x)
    z[:, ::up_y, ::up_x] = x
    # padding; negative pads crop
    z = F.pad(z, [max(pad_x0, 0), max(pad_x1, 0), max(pad_y0, 0), max(pad_y1, 0)])
    z = z[:, max(-pad_y0, 0): z.shape[1] - max(-pad_y1, 0), max(-pad_x0, 0): z.shape[2] - max(-pad_x1, 0)]
    # correlation with the flipped kernel == true convolution with the kernel
    w = torch.flip(kernel, [0, 1])[None, None]
    y = F.conv2d(z[:, None], w)[:, 0]
    return y[:, ::down_y, ::down_x]
