"""``conv2d`` / ``conv_transpose2d`` with the signatures of the reference's ``network/styleunet/conv2d_gradfix.py:22-75``,
backed by the MFMA kernels of ``libag_hip.so`` (``include/ag_conv.h``): forward, input gradient and weight gradient are
all ours (no MIOpen / cuDNN call on this path).

Supported = what the product uses (batch 1, groups 1): conv2d k in {1,3,4}, stride 1|2, zero padding; conv_transpose2d
3x3 stride 2 padding 0.  Anything else raises -- there is no fallback.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib

AG_CONV, AG_CONV_TRANSPOSE = 0, 1
MATH_MODES = {"fp32": 0, "split_bf16": 1, "split_bf16x3": 2, "split_f16": 3, "f16": 4}      # include/ag_conv.h AgConvMath
SCALED_MODES = ("split_f16", "f16")          # the forms that need every operand tensor's largest magnitude (per-tensor power-of-two scale)


def set_math(mode: str) -> str:
    """Arithmetic of the MFMA convolutions, process-wide: ``"split_f16"`` (default: fp32 operands as two fp16 parts under a per-tensor
    power-of-two scale, three products on the fp16 matrix pipe, fp32 accumulation), ``"split_bf16"`` (three bf16 parts, six products) --
    both with products within 2^-23 of the exact ones --, ``"fp32"`` (v_mfma_f32_32x32x2_f32) or the opt-in ``"split_bf16x3"`` (three
    products, 3 * 2^-16 per product: not fp32-grade) or the opt-in ``"f16"`` (one fp16 part per operand under the same scale: 11 significant bits,
    the operand grade of the cuDNN TF32 path the reference's convolutions take on its own hardware).  include/ag_conv.h has the contracts.
    Returns the previous mode."""
    if mode not in MATH_MODES:
        raise ValueError(f"conv math mode must be one of {sorted(MATH_MODES)}")
    prev = get_math()
    _lib.check(_lib.lib().ag_conv_set_math(MATH_MODES[mode]), "ag_conv_set_math")
    return prev


def get_math() -> str:
    m = _lib.lib().ag_conv_get_math()
    return next(k for k, v in MATH_MODES.items() if v == m)


def check_status(clear: bool = True) -> None:
    """Raises ``AgNativeError`` if a convolution enqueued so far met a non-finite accumulator (include/ag_conv.h ``ag_conv_status``: an operand
    beyond the maximum its fp16 scale was taken from -- a stale handed-over maximum -- or non-finite inputs).  Does not synchronise: call it after
    ``torch.cuda.synchronize()`` for a definitive answer.  The next convolution call raises by itself otherwise."""
    _lib.check(_lib.lib().ag_conv_status(int(bool(clear))), "ag_conv_status")


def needs_maxima() -> bool:
    """True in the arithmetic modes that scale every operand tensor by its largest magnitude (fp16 forms)."""
    return get_math() in SCALED_MODES


if os.environ.get("AG_CONV_MATH"):          # A/B hook for the profiles/ scripts: initial mode of the process
    set_math(os.environ["AG_CONV_MATH"])


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _desc(kind, Cin, Cout, H, W, k, stride, padding, weight_scale=1.0):
    d = _lib.AgConvDesc()
    d.kind, d.Cin, d.Cout, d.H, d.W, d.k, d.stride, d.padding = kind, Cin, Cout, H, W, k, stride, padding
    d.weight_scale = float(weight_scale)
    return d


_WS_BYTES = {}


def _workspace(d, dev):
    key = (d.kind, d.Cin, d.Cout, d.k, d.stride, d.padding, d.H, d.W)
    n = _WS_BYTES.get(key)
    if n is None:                                   # a function of the descriptor only: asked once per layer shape
        n = _WS_BYTES[key] = _lib.lib().ag_conv_workspace_bytes(ctypes.byref(d))
    return torch.empty((n,), dtype=torch.uint8, device=dev), n


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, out_scale, kind, stride, padding, weight_scale=1.0):
        L = _lib.lib()
        if x.dim() != 4 or x.shape[0] != 1:
            raise RuntimeError("MFMA conv path: batch must be 1 (the product renders one pose per step)")
        if not (x.is_cuda and w.is_cuda) or x.dtype != torch.float32 or w.dtype != torch.float32:
            raise RuntimeError("MFMA conv path: float32 GPU tensors only")
        x, w = x.contiguous(), w.contiguous()
        _, Cin, H, W = x.shape
        k = int(w.shape[-1])
        Cout = int(w.shape[0] if kind == AG_CONV else w.shape[1])
        if (kind == AG_CONV and w.shape[1] != Cin) or (kind == AG_CONV_TRANSPOSE and w.shape[0] != Cin) or w.shape[-2] != k:
            raise RuntimeError("weight shape does not match the input channels / square kernel")
        if out_scale is not None and out_scale.requires_grad:
            raise RuntimeError("MFMA conv path: out_scale is a constant of the node (no gradient is produced for it); pass a detached "
                               "tensor, or differentiate the demodulation through `weight` as ModulatedConv2d does")
        d = _desc(kind, Cin, Cout, H, W, k, stride, padding, weight_scale)
        if kind == AG_CONV:                         # ag_conv_output_size's formula (the native call validates the descriptor itself)
            oh, ow = (H + 2 * padding - k) // stride + 1, (W + 2 * padding - k) // stride + 1
        else:
            oh, ow = (H - 1) * 2 + k, (W - 1) * 2 + k
        if oh <= 0 or ow <= 0:
            raise RuntimeError("MFMA conv path: kernel larger than the padded input")
        y = torch.empty((1, Cout, oh, ow), dtype=torch.float32, device=x.device)
        ws, n = _workspace(d, x.device)
        b = bias.contiguous() if bias is not None else None
        sc = out_scale.contiguous() if out_scale is not None else None
        with _lib.on_device(x.device):
            _lib.check(L.ag_conv_forward(ctypes.byref(d), _p(x), _p(w), _p(sc), _p(b), _p(y), _p(ws), n, _stream(x.device)),
                       "ag_conv_forward")
        ctx.save_for_backward(x, w, sc)
        ctx.cfg = (kind, stride, padding, bias is not None, float(weight_scale))
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        x, w, sc = ctx.saved_tensors
        kind, stride, padding, has_bias, weight_scale = ctx.cfg
        _, Cin, H, W = x.shape
        k = int(w.shape[-1])
        Cout = int(w.shape[0] if kind == AG_CONV else w.shape[1])
        d = _desc(kind, Cin, Cout, H, W, k, stride, padding, weight_scale)
        gy = gy.contiguous()
        gbias = gy.sum((0, 2, 3)) if has_bias else None
        gscale = None
        if sc is not None:
            # y = conv * s + b: the conv result is not stored; recover d/ds from y would need it, so out_scale is treated as
            # a constant of this node (ModulatedConv2d differentiates its demodulation coefficients through `weight`)
            gy = gy * sc.view(1, -1, 1, 1)
        ws, n = _workspace(d, x.device)
        gx = gw = None
        with _lib.on_device(x.device):
            if ctx.needs_input_grad[0]:
                gx = torch.empty_like(x)
                _lib.check(L.ag_conv_backward_input(ctypes.byref(d), _p(gy), _p(w), _p(gx), _p(ws), n, _stream(x.device)),
                           "ag_conv_backward_input")
            if ctx.needs_input_grad[1]:
                gw = torch.empty_like(w)
                _lib.check(L.ag_conv_backward_weight(ctypes.byref(d), _p(x), _p(gy), _p(gw), _p(ws), n, _stream(x.device)),
                           "ag_conv_backward_weight")
        return gx, gw, gbias, gscale, None, None, None, None


def _one(v):
    return int(v[0]) if isinstance(v, (tuple, list)) else int(v)


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, out_scale=None, weight_scale=1.0):
    """``weight_scale``: convolve with ``weight * weight_scale`` (EqualConv2d, dual_styleunet.py:100-117) without materialising the
    scaled tensor -- the product is formed while the weights are re-packed; the weight gradient is w.r.t. ``weight``."""
    if _one(dilation) != 1 or groups != 1:
        raise RuntimeError("MFMA conv path: dilation 1 and groups 1 only")
    return _Conv.apply(input, weight, bias, out_scale, AG_CONV, _one(stride), _one(padding), weight_scale)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1, out_scale=None):
    if _one(dilation) != 1 or groups != 1 or _one(output_padding) != 0:
        raise RuntimeError("MFMA conv path: dilation 1, groups 1, output_padding 0 only")
    return _Conv.apply(input, weight, bias, out_scale, AG_CONV_TRANSPOSE, _one(stride), _one(padding), 1.0)
