"""Layer-level autograd nodes of the StyleUNet: ConvLayer, StyledConv and ToRGB (dual_styleunet.py:326-371, 570-604, 607-633) each as
ONE ``torch.autograd.Function`` that runs the same kernels, in the same order, as the chain of per-kernel Functions in
``styleunet_ops`` / ``conv`` -- so the results are bit-identical -- but costs autograd one node instead of three to six.

Why: after the convolutions moved to the bf16 matrix pipe the training step was bound by the host (profiles/host_op_times.py: 1300
custom-op calls per step, 13-32 us each inside their bodies plus autograd's own per-node bookkeeping).  The per-kernel Functions stay
the single implementation of every kernel call: a fused node calls their ``forward`` / ``backward`` static methods as plain functions
with a stand-in for autograd's ``ctx`` (``_Sub``), it does not duplicate them.
"""
from __future__ import annotations

import torch

from . import conv as agc
from .styleunet_ops import (_HAAR_ANALYSIS, _HAAR_SYNTHESIS, _Block2x2, _ModulateWeight, _NoiseBiasAct, _transpose4, _UpFirDn2d)

_SQRT2 = 2 ** 0.5


class _Sub:
    """Stands in for the ``ctx`` of a custom Function whose forward / backward is called directly inside a fused node."""
    __slots__ = ("saved_tensors", "cfg", "needs_input_grad")

    def __init__(self, needs_input_grad=()):
        self.saved_tensors = ()
        self.cfg = None
        self.needs_input_grad = needs_input_grad

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


def _stash(ctx, subs):
    """Hand every tensor the sub-steps saved to autograd's own save_for_backward (an OUTPUT of the node kept as a plain attribute would
    form a reference cycle node -> output -> node and its memory would wait for the garbage collector)."""
    flat, layout = [], []
    for s in subs:
        n = len(s.saved_tensors) if s is not None else 0
        layout.append(n)
        if n:
            flat.extend(s.saved_tensors)
            s.saved_tensors = ()
    ctx.save_for_backward(*flat)
    ctx.subs, ctx.layout = subs, layout


def _unstash(ctx):
    saved, i = ctx.saved_tensors, 0
    for s, n in zip(ctx.subs, ctx.layout):
        if n:
            s.saved_tensors = tuple(saved[i:i + n])
            i += n
    return ctx.subs


def _release(ctx):
    """Drop the sub-steps' references again when the backward is done: they include the node's own output (the activation kernel's
    backward reads it), and a reference from the node to its output that autograd does not know about is a cycle only the garbage
    collector breaks -- gigabytes of activations per step would pile up until it runs."""
    for s in ctx.subs:
        if s is not None:
            s.saved_tensors = ()


def _releasing(backward):
    def wrapped(ctx, *grads):
        try:
            return backward(ctx, *grads)
        finally:
            _release(ctx)
    return wrapped


class _ConvLayer(torch.autograd.Function):
    """[Blur] + EqualConv2d + FusedLeakyReLU(bias)."""

    @staticmethod
    def forward(ctx, x, w, bias, k_blur, scale, downsample):
        s_blur = s_conv = None
        k = int(w.shape[-1])
        if downsample:
            s_blur = _Sub()
            x = _UpFirDn2d.forward(s_blur, x, k_blur, (1, 1), (1, 1), (2, 2, 2, 2))
        s_conv = _Sub()
        y = agc._Conv.forward(s_conv, x, w, None, None, agc.AG_CONV, 2 if downsample else 1, 0 if downsample else k // 2, scale)
        s_act = _Sub()
        out = _NoiseBiasAct.forward(s_act, y, None, None, bias, 0.2, _SQRT2)
        _stash(ctx, (s_blur, s_conv, s_act))
        return out

    @staticmethod
    @_releasing
    def backward(ctx, g):
        s_blur, s_conv, s_act = _unstash(ctx)
        nx, nw, nb = ctx.needs_input_grad[:3]
        s_act.needs_input_grad = (True, False, False, nb, False, False)
        g, _, _, gb, _, _ = _NoiseBiasAct.backward(s_act, g)
        s_conv.needs_input_grad = (nx, nw)
        gx, gw = agc._Conv.backward(s_conv, g)[:2]
        if s_blur is not None and gx is not None:
            gx = _UpFirDn2d.backward(s_blur, gx)[0]
        return gx, gw, gb, None, None, None


class _StyledConv(torch.autograd.Function):
    """ModulatedConv2d (modulate + demodulate the weight, [transposed] convolution, [blur]) + NoiseInjection + FusedLeakyReLU."""

    @staticmethod
    def forward(ctx, x, w, style, noise, noise_weight, act_bias, k_blur, mod_scale, upsample):
        s_mod = _Sub()
        wm = _ModulateWeight.forward(s_mod, w, style, mod_scale, True, upsample)
        s_conv = _Sub()
        s_blur = None
        if upsample:
            y = agc._Conv.forward(s_conv, x, wm, None, None, agc.AG_CONV_TRANSPOSE, 2, 0, 1.0)
            s_blur = _Sub()
            y = _UpFirDn2d.forward(s_blur, y, k_blur, (1, 1), (1, 1), (1, 1, 1, 1))
        else:
            y = agc._Conv.forward(s_conv, x, wm, None, None, agc.AG_CONV, 1, 1, 1.0)
        s_act = _Sub()
        out = _NoiseBiasAct.forward(s_act, y, noise, noise_weight if noise is not None else None, act_bias, 0.2, _SQRT2)
        _stash(ctx, (s_mod, s_conv, s_blur, s_act))
        return out

    @staticmethod
    @_releasing
    def backward(ctx, g):
        s_mod, s_conv, s_blur, s_act = _unstash(ctx)
        nx, nw, ns, _, nnw, nb = ctx.needs_input_grad[:6]
        s_act.needs_input_grad = (True, False, nnw, nb, False, False)
        g, _, gnw, gb, _, _ = _NoiseBiasAct.backward(s_act, g)
        if s_blur is not None:
            g = _UpFirDn2d.backward(s_blur, g)[0]
        s_conv.needs_input_grad = (nx, nw or ns)
        gx, gwm = agc._Conv.backward(s_conv, g)[:2]
        gw = gs = None
        if gwm is not None:
            gw, gs = _ModulateWeight.backward(s_mod, gwm)[:2]
        return gx, gw, gs, None, gnw, gb, None, None, None


class _ToRGB(torch.autograd.Function):
    """ToRGB: modulated 1 x 1 convolution (no demodulation) + bias [+ wavelet-domain upsampled skip: iwt -> Upsample -> dwt]."""

    @staticmethod
    def forward(ctx, x, w, style, bias, skip, k_blur_up, mod_scale):
        s_mod = _Sub()
        wm = _ModulateWeight.forward(s_mod, w, style, mod_scale, False, False)
        s_conv = _Sub()
        out = agc._Conv.forward(s_conv, x, wm, bias, None, agc.AG_CONV, 1, 0, 1.0)
        s_up = None
        if skip is not None:
            s_m, s_u, s_s = _Sub(), _Sub(), _Sub()
            t = _Block2x2.forward(s_m, skip, _HAAR_SYNTHESIS, True)
            t = _UpFirDn2d.forward(s_u, t, k_blur_up, (2, 2), (1, 1), (2, 1, 2, 1))
            out.add_(_Block2x2.forward(s_s, t, _HAAR_ANALYSIS, False))
            s_up = s_u
        _stash(ctx, (s_mod, s_conv, s_up))
        return out

    @staticmethod
    @_releasing
    def backward(ctx, g):
        s_mod, s_conv, s_up = _unstash(ctx)
        nx, nw, ns, nb, nskip = ctx.needs_input_grad[:5]
        g = g.contiguous()
        gskip = None
        if s_up is not None and nskip:
            # adjoints of the two fixed 2 x 2 block transforms: the other transform with the transposed matrix
            t = _Block2x2.forward(_Sub(), g, _transpose4(_HAAR_ANALYSIS), True)
            t = _UpFirDn2d.backward(s_up, t)[0]
            gskip = _Block2x2.forward(_Sub(), t, _transpose4(_HAAR_SYNTHESIS), False)
        s_conv.needs_input_grad = (nx, nw or ns)
        s_conv.cfg = s_conv.cfg[:3] + (bool(nb),) + s_conv.cfg[4:]          # bias gradient only when the bias wants one
        gx, gwm, gb = agc._Conv.backward(s_conv, g)[:3]
        gw = gs = None
        if gwm is not None:
            gw, gs = _ModulateWeight.backward(s_mod, gwm)[:2]
        return gx, gw, gs, gb, gskip, None, None


def conv_layer(x, w, bias, k_blur, scale, downsample):
    return _ConvLayer.apply(x, w, bias, k_blur, float(scale), bool(downsample))


def styled_conv(x, w, style, noise, noise_weight, act_bias, k_blur, mod_scale, upsample):
    return _StyledConv.apply(x, w, style, noise, noise_weight, act_bias, k_blur, float(mod_scale), bool(upsample))


def to_rgb(x, w, style, bias, skip, k_blur_up, mod_scale):
    return _ToRGB.apply(x, w, style, bias, skip, k_blur_up, float(mod_scale))
