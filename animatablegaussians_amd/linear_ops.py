"""Host side of include/ag_linear.h: groups of EqualLinear layers on one-row inputs (the style path of the StyleUNets,
reference network/styleunet/dual_styleunet.py:131-165, 594-610) and the bilinear resize of the view-direction feature (:881-883).
Plumbing only; no CPU / eager fallback: non-GPU tensors raise."""
from __future__ import annotations

import ctypes
import math
from typing import Optional, Sequence

import torch

from . import _lib

MAX_JOBS = _lib.AG_LINEAR_MAX_JOBS


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _check_gpu(t, what):
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError(f"{what}: float32 GPU tensors only (no CPU path in this package)")


def _pad4(n):
    return (n + 3) & ~3


def _carve(flat, B, outs):
    """[B, out_j] tensors inside one allocation, each starting 16-byte aligned (an output may be the next call's input)."""
    ys, off = [], 0
    for o in outs:
        ys.append(flat[off:off + B * o].view(B, o))
        off += _pad4(B * o)
    return ys


class _EqualLinearGroup(torch.autograd.Function):
    """(y_j [B, out_j]) of n EqualLinear layers ("jobs"): job j reads xs[j] (jobs sharing an input are consecutive and pass the same tensor),
    ``act``: the reference's fused leaky ReLU on every job; ``normalize``: PixelNorm the inputs first (they must not need a gradient)."""

    @staticmethod
    def forward(ctx, n, act, normalize, alphas, bias_muls, *tensors):
        xs, ws, bs = tensors[:n], tensors[n:2 * n], tensors[2 * n:3 * n]
        dev = xs[0].device
        xs_c, prev = [], None
        for x in xs:                      # one contiguous copy per DISTINCT input: jobs that share a tensor must keep sharing the pointer
            _check_gpu(x, "equal_linear")
            if prev is not None and x is prev[0]:
                xs_c.append(prev[1])
                continue
            xc = x.contiguous()
            prev = (x, xc)
            xs_c.append(xc)
        ws_c = [w.contiguous() for w in ws]
        bs_c = [b.contiguous() if b is not None else None for b in bs]
        B, fin = int(xs_c[0].shape[0]), int(xs_c[0].shape[1])
        for x, w, b in zip(xs_c, ws_c, bs_c):
            if x.dim() != 2 or tuple(x.shape) != (B, fin) or w.dim() != 2 or int(w.shape[1]) != fin or (b is not None and b.numel() != w.shape[0]):
                raise RuntimeError("equal_linear: inputs [B, in], weights [out, in], biases [out]")
        # one allocation, one tensor per job (the outputs are separate autograd tensors: a single [B, sum out_j] output that the caller splits
        # costs a concatenation kernel per call in the backward)
        outs = [int(w.shape[0]) for w in ws_c]
        flat = torch.empty(sum(_pad4(B * o) for o in outs), dtype=torch.float32, device=dev)
        ys = _carve(flat, B, outs)
        a = _EqualLinearGroup._args(n, B, fin, act, normalize, alphas, bias_muls, xs_c, ws_c, bs_c, ys)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().ag_equal_linear_forward(ctypes.byref(a), _stream(dev)), "ag_equal_linear_forward")
        ctx.save_for_backward(flat, *xs_c, *ws_c, *[b for b in bs_c if b is not None])
        ctx.cfg = (n, bool(act), bool(normalize), tuple(alphas), tuple(bias_muls), tuple(b is not None for b in bs_c),
                   tuple(tuple(b.shape) if b is not None else None for b in bs))
        return tuple(ys)

    @staticmethod
    def _args(n, B, fin, act, normalize, alphas, bias_muls, xs, ws, bs, ys):
        if not 1 <= n <= MAX_JOBS:
            raise RuntimeError(f"equal_linear: 1 .. {MAX_JOBS} layers per call")
        a = _lib.AgEqualLinearArgs()
        a.n_jobs, a.B, a.in_features, a.act, a.normalize_input = n, B, fin, int(bool(act)), int(bool(normalize))
        for j in range(n):
            a.x[j], a.weight[j] = xs[j].data_ptr(), ws[j].data_ptr()
            a.bias[j] = bs[j].data_ptr() if bs[j] is not None else None
            a.out_features[j] = int(ws[j].shape[0])
            a.alpha[j], a.bias_mul[j] = float(alphas[j]), float(bias_muls[j])
            a.y[j] = ys[j].data_ptr()
        return a

    @staticmethod
    def backward(ctx, *gs):
        n, act, normalize, alphas, bias_muls, has_b, b_shapes = ctx.cfg
        saved = ctx.saved_tensors
        flat, xs, ws = saved[0], saved[1:1 + n], saved[1 + n:1 + 2 * n]
        rest = list(saved[1 + 2 * n:])
        bs = [rest.pop(0) if hb else None for hb in has_b]
        dev = flat.device
        B, fin = int(xs[0].shape[0]), int(xs[0].shape[1])
        ys = _carve(flat, B, [int(w.shape[0]) for w in ws])
        gs = [g.contiguous() if g is not None else torch.zeros_like(y) for g, y in zip(gs, ys)]
        nig = ctx.needs_input_grad[5:]
        need_x, need_w, need_b = nig[:n], nig[n:2 * n], nig[2 * n:3 * n]
        a = _EqualLinearGroup._args(n, B, fin, act, normalize, alphas, bias_muls, xs, ws, bs, ys)
        for j in range(n):
            a.g_y[j] = gs[j].data_ptr()
        gxs, gws, gbs = [None] * n, [None] * n, [None] * n
        group_gx = None
        for j in range(n):
            first = j == 0 or xs[j].data_ptr() != xs[j - 1].data_ptr()
            if first:
                # the jobs of one input: a gradient if ANY of their positions asks for one (they are the same tensor: all or none do)
                k = j
                while k + 1 < n and xs[k + 1].data_ptr() == xs[j].data_ptr():
                    k += 1
                group_gx = torch.empty_like(xs[j]) if any(need_x[j:k + 1]) else None
                gxs[j] = group_gx                          # autograd sums the positions of one tensor: the gradient goes to the first, None to the rest
            a.g_x[j] = group_gx.data_ptr() if group_gx is not None else None
            if need_w[j]:
                gws[j] = torch.empty_like(ws[j])
                a.g_weight[j] = gws[j].data_ptr()
            if need_b[j] and bs[j] is not None:
                gbs[j] = torch.empty(bs[j].numel(), dtype=torch.float32, device=dev)
                a.g_bias[j] = gbs[j].data_ptr()
        scratch = None
        if any(t is not None for t in gxs):
            if normalize:
                raise RuntimeError("equal_linear: no input gradient through the PixelNorm option")
            scratch = torch.empty(int(_lib.lib().ag_equal_linear_scratch_floats(ctypes.byref(a))), dtype=torch.float32, device=dev)
            a.scratch = scratch.data_ptr()
        with _lib.on_device(dev):
            _lib.check(_lib.lib().ag_equal_linear_backward(ctypes.byref(a), _stream(dev)), "ag_equal_linear_backward")
        gbs = [t.view(s) if t is not None else None for t, s in zip(gbs, b_shapes)]
        return (None, None, None, None, None, *gxs, *gws, *gbs)


def equal_linear_group(xs: Sequence[torch.Tensor], weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]], lr_mul: float = 1.0,
                       activation: bool = False, normalize_input: bool = False):
    """``[EqualLinear_j(xs[j]) for j]`` (dual_styleunet.py:131-165: ``F.linear(x, W * scale, b * lr_mul)``, ``scale = lr_mul /
    sqrt(in)``; with ``activation`` the fused leaky ReLU on top) as one native launch; a tuple of [B, out_j] tensors.  ``xs``: one tensor per layer, or a single tensor for all;
    layers that read the same tensor must be consecutive.  ``normalize_input``: PixelNorm (:13-18) the inputs first."""
    n = len(weights)
    if isinstance(xs, torch.Tensor):
        xs = [xs] * n
    if len(xs) != n or len(biases) != n:
        raise RuntimeError("equal_linear_group: one input and one bias entry per layer")
    alphas = [lr_mul / math.sqrt(w.shape[1]) for w in weights]
    return _EqualLinearGroup.apply(n, bool(activation), bool(normalize_input), tuple(alphas), tuple([float(lr_mul)] * n), *xs, *weights, *biases)


class _BilinearResize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, oh, ow):
        _check_gpu(x, "bilinear_resize")
        x = x.contiguous()
        N = int(x.shape[0]) * int(x.shape[1])
        H, W = int(x.shape[2]), int(x.shape[3])
        out = torch.empty((x.shape[0], x.shape[1], oh, ow), dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ag_bilinear_resize_forward(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(x.data_ptr()), N, H, W, int(oh), int(ow),
                                                            _stream(x.device)), "ag_bilinear_resize_forward")
        ctx.cfg = (tuple(x.shape), int(oh), int(ow))
        return out

    @staticmethod
    def backward(ctx, g):
        shape, oh, ow = ctx.cfg
        g = g.contiguous()
        gx = torch.empty(shape, dtype=torch.float32, device=g.device)
        with _lib.on_device(g.device):
            _lib.check(_lib.lib().ag_bilinear_resize_backward(ctypes.c_void_p(gx.data_ptr()), ctypes.c_void_p(g.data_ptr()), shape[0] * shape[1], shape[2],
                                                             shape[3], oh, ow, _stream(g.device)), "ag_bilinear_resize_backward")
        return gx, None, None


def bilinear_resize(x: torch.Tensor, size) -> torch.Tensor:
    """``F.interpolate(x, size, mode="bilinear")`` (align_corners False) of an NCHW tensor as one streaming kernel each way; the backward is a
    gather in a fixed order (torch's adds with atomics)."""
    if x.dim() != 4:
        raise RuntimeError("bilinear_resize: NCHW input")
    return _BilinearResize.apply(x, int(size[0]), int(size[1]))


def plane_sums(x: torch.Tensor) -> torch.Tensor:
    """``x.sum((-2, -1))`` of a contiguous [..., H, W] tensor (the bias gradient of a ToRGB head), deterministic, one streaming pass
    (include/ag_linear.h ag_plane_sums).  No autograd: used inside backward passes."""
    _check_gpu(x, "plane_sums")
    x = x.contiguous()
    planes, n = int(x.numel() // (x.shape[-2] * x.shape[-1])), int(x.shape[-2] * x.shape[-1])
    out = torch.empty(tuple(x.shape[:-2]), dtype=torch.float32, device=x.device)
    scratch = torch.empty(int(_lib.lib().ag_plane_sums_scratch_floats(planes, n)), dtype=torch.float32, device=x.device)
    with _lib.on_device(x.device):
        _lib.check(_lib.lib().ag_plane_sums(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(x.data_ptr()), planes, n, ctypes.c_void_p(scratch.data_ptr()),
                                           _stream(x.device)), "ag_plane_sums")
    return out


def select_add_rows(out: torch.Tensor, src, vf: Optional[torch.Tensor], rows) -> torch.Tensor:
    """``x[m] = out[src[m]]`` (+ ``vf[m - rows[0]]``, bilinearly resized to ``out``'s resolution when its own differs, for ``rows[0] <= m < rows[1]``)
    in one pass (include/ag_linear.h ag_select_add_rows).  No autograd: the forward of grouped._SelectAddRows."""
    _check_gpu(out, "select_add_rows")
    out = out.contiguous()
    M = len(src)
    x = torch.empty((M,) + tuple(out.shape[1:]), dtype=torch.float32, device=out.device)
    tab = (ctypes.c_int32 * M)(*[int(v) for v in src])
    if vf is not None:
        _check_gpu(vf, "select_add_rows")
        vf = vf.contiguous()
        if vf.shape[0] != rows[1] - rows[0] or vf.shape[1] != out.shape[1]:
            raise RuntimeError("select_add_rows: one view feature [C, h, w] per member of the row range")
    with _lib.on_device(out.device):
        _lib.check(_lib.lib().ag_select_add_rows(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.cast(tab, ctypes.c_void_p), M,
                                                int(out.shape[1]), int(out.shape[2]), int(out.shape[3]),
                                                ctypes.c_void_p(vf.data_ptr()) if vf is not None else None, int(rows[0]) if vf is not None else 0,
                                                int(rows[1]) if vf is not None else 0, int(vf.shape[2]) if vf is not None else 0,
                                                int(vf.shape[3]) if vf is not None else 0, _stream(out.device)), "ag_select_add_rows")
    return x


def bilinear_resize_backward(g: torch.Tensor, size) -> torch.Tensor:
    """The adjoint of ``bilinear_resize`` to ``size`` applied to ``g`` [N, C, OH, OW] (a contiguous tensor or a leading-dimension slice of one)."""
    _check_gpu(g, "bilinear_resize_backward")
    g = g.contiguous()
    gx = torch.empty((g.shape[0], g.shape[1], int(size[0]), int(size[1])), dtype=torch.float32, device=g.device)
    with _lib.on_device(g.device):
        _lib.check(_lib.lib().ag_bilinear_resize_backward(ctypes.c_void_p(gx.data_ptr()), ctypes.c_void_p(g.data_ptr()), int(g.shape[0] * g.shape[1]),
                                                         int(size[0]), int(size[1]), int(g.shape[2]), int(g.shape[3]), _stream(g.device)),
                   "ag_bilinear_resize_backward")
    return gx
