"""Layer-level autograd nodes of the StyleUNet: ConvLayer, StyledConv and ToRGB (dual_styleunet.py:326-371, 570-604, 607-633) each as
ONE ``torch.autograd.Function`` that runs the same kernels, in the same order, as the chain of per-kernel Functions in
``styleunet_ops`` / ``conv`` -- so the results are bit-identical -- but costs autograd one node instead of three to six.

Why: after the convolutions moved to the 16-bit matrix pipe (round 2) the training step was bound by the host (profiles/host_op_times.py: 1300
custom-op calls per step, 13-32 us each inside their bodies plus autograd's own per-node bookkeeping).  The per-kernel Functions stay
the single implementation of every kernel call: a fused node calls their ``forward`` / ``backward`` static methods as plain functions
with a stand-in for autograd's ``ctx`` (``_Sub``), it does not duplicate them.
"""
from __future__ import annotations

import torch

from . import conv as agc
import os

from .styleunet_ops import (_HAAR_ANALYSIS, _HAAR_SYNTHESIS, _Block2x2, _ModulateWeight, _NoiseBiasAct, _transpose4, _UpFirDn2d,
                            skip_chain_backward, skip_chain_forward_)

# ToRGB's iwt -> Upsample -> dwt skip path as one kernel (AG_SKIP_CHAIN=0: the three kernels + the addition, for the A/B and the equality test)
_SKIP_CHAIN = os.environ.get("AG_SKIP_CHAIN") != "0"


def set_skip_chain(on: bool) -> bool:
    global _SKIP_CHAIN
    prev, _SKIP_CHAIN = _SKIP_CHAIN, bool(on)
    return prev

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
        ctx.skip_k = None
        if skip is not None:
            if _SKIP_CHAIN:                                   # one kernel: the composed linear map, accumulated into the layer's output
                skip_chain_forward_(out, skip, k_blur_up, accumulate=True)
                ctx.skip_k = k_blur_up                        # a module buffer, not an output of this node: a plain attribute is safe
            else:
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
        if ctx.skip_k is not None and nskip:                 # the fused skip path: one adjoint kernel
            gskip = skip_chain_backward(g, ctx.skip_k)
        elif s_up is not None and nskip:
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


# ---------------------------------------------------------------------------------------------------------------------------------
# One NATIVE call per layer (include/ag_layers.h): the same kernel sequence issued from C.  The nodes above cost the host one
# Python -> ctypes transition, one output allocation and one argument check per kernel (3-5 per layer and direction, 13-32 us each);
# these cost one per layer.  AG_LAYER_CALLS=0 selects the nodes above (A/B, and the bit-equality test of the two paths).
# ---------------------------------------------------------------------------------------------------------------------------------
import ctypes  # noqa: E402
import os  # noqa: E402

from . import _lib  # noqa: E402
from .styleunet_ops import _flipped  # noqa: E402

_LAYER_CALLS = os.environ.get("AG_LAYER_CALLS") != "0"
_SIZES = {}          # layer shape -> (OH, OW, forward scratch floats, backward scratch floats, conv workspace bytes)


def set_layer_calls(on: bool) -> bool:
    global _LAYER_CALLS
    prev, _LAYER_CALLS = _LAYER_CALLS, bool(on)
    return prev


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _layer_sizes(a):
    key = (a.Cin, a.Cout, a.H, a.W, a.k, a.resample, a.modulated)
    v = _SIZES.get(key)
    if v is None:
        L = _lib.lib()
        oh, ow = ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.check(L.ag_layer_output_size(ctypes.byref(a), ctypes.byref(oh), ctypes.byref(ow)), "ag_layer_output_size")
        d = _lib.AgConvDesc()
        d.Cin, d.Cout, d.k = a.Cin, a.Cout, a.k
        if not a.modulated:
            d.kind, d.H, d.W = agc.AG_CONV, a.H + (1 if a.resample else 0), a.W + (1 if a.resample else 0)
            d.stride, d.padding = (2, 0) if a.resample else (1, a.k // 2)
        else:
            d.kind, d.H, d.W = (agc.AG_CONV_TRANSPOSE if a.resample else agc.AG_CONV), a.H, a.W
            d.stride, d.padding = (2, 0) if a.resample else (1, a.k // 2)
        d.weight_scale = 1.0
        v = _SIZES[key] = (oh.value, ow.value, int(L.ag_layer_scratch_floats(ctypes.byref(a), 0)), int(L.ag_layer_scratch_floats(ctypes.byref(a), 1)),
                           int(L.ag_conv_workspace_bytes(ctypes.byref(d))))
    return v


def _describe(x, w, k_blur, scale, resample, modulated):
    if x.dim() != 4 or x.shape[0] != 1 or not x.is_cuda or x.dtype != torch.float32 or w.dtype != torch.float32:
        raise RuntimeError("layer call: float32 GPU tensors, batch 1")
    a = _lib.AgLayerArgs()
    a.Cout, a.Cin, a.k = int(w.shape[-4]), int(w.shape[-3]), int(w.shape[-1])
    a.H, a.W = int(x.shape[2]), int(x.shape[3])
    if int(x.shape[1]) != a.Cin:
        raise RuntimeError("weight shape does not match the input channels")
    a.resample, a.modulated = int(bool(resample)), int(bool(modulated))
    a.scale, a.slope, a.act_scale = float(scale), 0.2, _SQRT2
    if resample and (k_blur is None or tuple(k_blur.shape) != (4, 4)):
        raise RuntimeError("resampling layers use the 4 x 4 FIR kernel")
    return a


def _scratch(a, backward, dev):
    """One buffer for the call's intermediates and the convolution workspace; returns (tensor, scratch pointer, workspace pointer, workspace bytes)."""
    _, _, f_fwd, f_bwd, ws = _layer_sizes(a)
    nfl = f_bwd if backward else f_fwd
    buf = torch.empty(nfl * 4 + ws + 512, dtype=torch.uint8, device=dev)
    base = (buf.data_ptr() + 255) & ~255
    return buf, base, base + ((nfl * 4 + 255) & ~255), ws


class _LayerCall(torch.autograd.Function):
    """ConvLayer (modulated = False: style / noise / noise_weight are None) or StyledConv as ONE native call each way."""

    @staticmethod
    def forward(ctx, x, w, style, noise, noise_weight, bias, k_blur, scale, resample, modulated):
        x, w = x.contiguous(), w.contiguous()
        a = _describe(x, w, k_blur, scale, resample, modulated)
        oh, ow = _layer_sizes(a)[:2]
        dev = x.device
        out = torch.empty((1, a.Cout, oh, ow), dtype=torch.float32, device=dev)
        keep = None                                       # what the backward needs besides the inputs and the output
        if modulated:
            style = style.contiguous()
            if style.numel() != a.Cin:
                raise RuntimeError("style must have one entry per input channel")
            keep = torch.empty(w.numel() + a.Cout, dtype=torch.float32, device=dev)      # modulated weight, then the demodulation coefficients
            a.w_mod, a.demod, a.style = keep.data_ptr(), keep.data_ptr() + 4 * w.numel(), style.data_ptr()
            if noise is not None and noise_weight is not None:
                noise = noise.contiguous()
                if noise.numel() != oh * ow:
                    raise RuntimeError("noise must be [1, 1, OH, OW]")
                a.noise, a.noise_weight = noise.data_ptr(), noise_weight.data_ptr()
            else:
                noise = noise_weight = None
        elif resample:
            keep = torch.empty((1, a.Cin, a.H + 1, a.W + 1), dtype=torch.float32, device=dev)    # the blurred input (the weight gradient's operand)
            a.x_blur = keep.data_ptr()
        a.x, a.weight, a.out, a.act_bias = x.data_ptr(), w.data_ptr(), out.data_ptr(), _ptr(bias)
        a.k_blur = _ptr(k_blur) if resample else None
        buf, a.scratch, a.workspace, a.workspace_bytes = _scratch(a, False, dev)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().ag_layer_forward(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "ag_layer_forward")
        ctx.save_for_backward(x, w, style, noise, noise_weight, bias, out, keep, _flipped(k_blur) if resample else None)
        ctx.cfg = (float(scale), bool(resample), bool(modulated))
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, style, noise, noise_weight, bias, out, keep, k_flip = ctx.saved_tensors
        scale, resample, modulated = ctx.cfg
        nx, nw, ns, _, nnw, nb = ctx.needs_input_grad[:6]
        a = _describe(x, w, k_flip, scale, resample, modulated)
        dev = x.device
        g = g.contiguous()
        a.x, a.weight, a.out, a.g_out, a.act_bias = x.data_ptr(), w.data_ptr(), out.data_ptr(), g.data_ptr(), _ptr(bias)
        a.k_blur = _ptr(k_flip)
        gx = torch.empty_like(x) if nx else None
        want_w = nw or (modulated and ns)
        gw = torch.empty_like(w) if want_w else None
        gs = gnw = gb = gbn = None
        if modulated:
            a.style, a.w_mod, a.demod = style.data_ptr(), keep.data_ptr(), keep.data_ptr() + 4 * w.numel()
            if noise is not None:
                a.noise, a.noise_weight = noise.data_ptr(), noise_weight.data_ptr()
            if want_w:
                gs = torch.empty_like(style)
        elif resample:
            a.x_blur = keep.data_ptr()
        a.want_bias = int(bool(nb and bias is not None))
        a.want_noise_weight = int(bool(modulated and nnw and noise is not None))
        if a.want_bias or a.want_noise_weight:
            gbn = torch.empty(a.Cout + 1, dtype=torch.float32, device=dev)
            gb = gbn[:a.Cout] if a.want_bias else None
            gnw = gbn[a.Cout:] if a.want_noise_weight else None
        a.g_x, a.g_weight, a.g_style, a.g_bias_noise = _ptr(gx), _ptr(gw), _ptr(gs), _ptr(gbn)
        buf, a.scratch, a.workspace, a.workspace_bytes = _scratch(a, True, dev)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().ag_layer_backward(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "ag_layer_backward")
        if gnw is not None and noise_weight is not None:
            gnw = gnw.view(noise_weight.shape)
        return gx, gw, gs, None, gnw, gb, None, None, None, None


def conv_layer(x, w, bias, k_blur, scale, downsample):
    if _LAYER_CALLS:
        return _LayerCall.apply(x, w, None, None, None, bias, k_blur if downsample else None, float(scale), bool(downsample), False)
    return _ConvLayer.apply(x, w, bias, k_blur, float(scale), bool(downsample))


def styled_conv(x, w, style, noise, noise_weight, act_bias, k_blur, mod_scale, upsample):
    if _LAYER_CALLS:
        return _LayerCall.apply(x, w, style, noise, noise_weight, act_bias, k_blur if upsample else None, float(mod_scale), bool(upsample), True)
    return _StyledConv.apply(x, w, style, noise, noise_weight, act_bias, k_blur, float(mod_scale), bool(upsample))


def to_rgb(x, w, style, bias, skip, k_blur_up, mod_scale):
    return _ToRGB.apply(x, w, style, bias, skip, k_blur_up, float(mod_scale))
