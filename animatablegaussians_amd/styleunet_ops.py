"""Python entry points of the two StyleUNet extension modules, backed by ``libag_hip.so``.

``fused_bias_act`` and ``upfirdn2d`` have exactly the signatures of the reference's pybind modules ``fused`` and
``upfirdn2d`` (``network/styleunet/fused_bias_act.cpp:17-31``, ``upfirdn2d.cpp:17-31``); the thin top-level modules in
``animatablegaussians_amd/dropin/`` re-export them under those names so the reference's ``fused_act.py`` /
``upfirdn2d.py`` (and therefore ``dual_styleunet.py``) import and run unchanged.  No CPU fallback.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() else None


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _check_input(t: torch.Tensor, name: str) -> None:
    # CHECK_INPUT of the reference: CUDA + contiguous (fused_bias_act.cpp:11-15)
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def fused_bias_act(input: torch.Tensor, bias: torch.Tensor, refer: torch.Tensor, act: int, grad: int, alpha: float,
                   scale: float) -> torch.Tensor:
    _check_input(input, "input")
    _check_input(bias, "bias")
    if input.dtype != torch.float32:
        raise RuntimeError("fused_bias_act: only float32 is implemented on this path")
    x = input.contiguous()
    ref = refer.contiguous() if refer.numel() else None
    b = bias.contiguous() if bias.numel() else None
    out = torch.empty_like(x)
    step_b = 1
    for i in range(2, x.dim()):
        step_b *= x.size(i)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ag_fused_bias_act(_p(out), _p(x), _p(b), _p(ref), int(act), int(grad), float(alpha),
                                                float(scale), x.numel(), step_b, b.numel() if b is not None else 0,
                                                _stream(x.device)), "ag_fused_bias_act")
    return out


def upfirdn2d(input: torch.Tensor, kernel: torch.Tensor, up_x: int, up_y: int, down_x: int, down_y: int, pad_x0: int,
              pad_x1: int, pad_y0: int, pad_y1: int) -> torch.Tensor:
    _check_input(input, "input")
    _check_input(kernel, "kernel")
    if input.dim() != 4:
        raise RuntimeError("upfirdn2d expects [major, in_h, in_w, minor]")
    major, in_h, in_w, minor = (int(v) for v in input.shape)
    kh, kw = (int(v) for v in kernel.shape)
    out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) // down_y
    out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) // down_x
    if input.dtype != torch.float32:
        raise RuntimeError("upfirdn2d: only float32 is implemented on this path")
    x = input
    if minor != 1:
        # the reference layout keeps `minor` innermost; every call site in this product has minor == 1
        x = input.permute(0, 3, 1, 2).reshape(major * minor, in_h, in_w, 1).contiguous()
    out = torch.empty((x.shape[0], out_h, out_w, 1), dtype=torch.float32, device=input.device)
    k = kernel.to(torch.float32).contiguous()
    with torch.cuda.device(input.device):
        _lib.check(_lib.lib().ag_upfirdn2d(_p(out), _p(x), _p(k), x.shape[0], in_h, in_w, kh, kw, up_x, up_y, down_x, down_y,
                                           pad_x0, pad_x1, pad_y0, pad_y1, _stream(input.device)), "ag_upfirdn2d")
    if minor != 1:
        out = out.reshape(major, minor, out_h, out_w).permute(0, 2, 3, 1).contiguous()
    return out
