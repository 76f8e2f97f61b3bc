"""Grouped execution of the avatar's StyleUNets (round 4; include/ag_layers.h ``AgGroupedLayerArgs``).

``network/avatar.py:34-36`` builds three ``DualStyleUNet`` with identical layer shapes -- only the ToRGB heads differ (12 / 12 / 32
rows) -- and ``:93-124`` evaluates all three on the same pose map; each runs two decoders of identical shapes
(``dual_styleunet.py:869-905``).  Run one network at a time that is ~1500 small launches per network pass, most of them layers at
<= 64^2 with a handful of workgroups each.  Here the three encoders run as ONE launch chain with a group dimension G = 3 and the six
decoders as one chain with G = 6: activations are stacked ``[G, C, H, W]``, every kernel gets G times the workgroups, the launch count
drops by the same factor.  The parameters stay the reference's separate tensors (checkpoints, optimiser state and ``parameters()``
order are untouched): a grouped layer takes a table of G device pointers per parameter kind, and writes the G gradients stacked so
that each parameter's gradient is a dense view of its own shape.

Numerics: the same kernels in the same order as the one-network path; the only difference is the split-K slice count of a convolution
(a function of the workgroup count), i.e. fp32 summation order.  ``tests/test_grouped_gpu.py`` pins grouped against one-by-one.
"""
from __future__ import annotations

import ctypes
import math
import os

import torch

from . import _lib
from . import conv as agc
from .linear_ops import bilinear_resize, bilinear_resize_backward, plane_sums, select_add_rows
from .styleunet import latents_of
from .styleunet_ops import _HAAR_SYNTHESIS, _flipped, _skip_taps_host, upfirdn2d_nchw

_SQRT2 = 2 ** 0.5
_SIZES = {}
# the decoders' comb convolutions without the concatenation, the encoder-level half once per network (AG_COMB_SPLIT=0: cat + one convolution
# per member, for the A/B and the equality test)
_COMB_SPLIT = os.environ.get("AG_COMB_SPLIT") != "0"


def set_comb_split(on: bool) -> bool:
    global _COMB_SPLIT
    prev, _COMB_SPLIT = _COMB_SPLIT, bool(on)
    return prev


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


_INDEX = {}


def _index(values, dev):
    """Cached int64 index tensor (built once per device: creating it is a host-to-device copy, which a hipGraph capture does not allow)."""
    key = (tuple(values), dev)
    t = _INDEX.get(key)
    if t is None:
        t = _INDEX[key] = torch.tensor(list(values), dtype=torch.int64, device=dev)
    return t


def _fill(table, tensors):
    for i, t in enumerate(tensors):
        table[i] = t.data_ptr() if t is not None else None


def _layer_args(G, x, w0, resample, modulated, scale, shared):
    a = _lib.AgGroupedLayerArgs()
    a.G = G
    a.Cout, a.Cin, a.k = int(w0.shape[-4]), int(w0.shape[-3]), int(w0.shape[-1])
    a.H, a.W = int(x.shape[2]), int(x.shape[3])
    if int(x.shape[1]) != a.Cin or x.shape[0] != (1 if shared else G):
        raise RuntimeError(f"grouped layer: input {tuple(x.shape)} does not match G = {G}, Cin = {a.Cin} (shared = {shared})")
    a.resample, a.modulated = int(bool(resample)), int(bool(modulated))
    a.scale, a.slope, a.act_scale = float(scale), 0.2, _SQRT2
    a.x_group_stride = 0 if shared else a.Cin * a.H * a.W
    return a


def _layer_sizes(a):
    key = ("L", a.G, a.Cin, a.Cout, a.H, a.W, a.k, a.resample, a.modulated)
    v = _SIZES.get(key)
    if v is None:
        L = _lib.lib()
        oh, ow = ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.check(L.ag_grouped_layer_output_size(ctypes.byref(a), ctypes.byref(oh), ctypes.byref(ow)), "ag_grouped_layer_output_size")
        v = _SIZES[key] = (oh.value, ow.value, int(L.ag_grouped_layer_scratch_floats(ctypes.byref(a), 0)),
                           int(L.ag_grouped_layer_scratch_floats(ctypes.byref(a), 1)), int(L.ag_grouped_layer_workspace_bytes(ctypes.byref(a))))
    return v


_MAXIMA = []


def _maxima_floats():
    if not _MAXIMA:
        _MAXIMA.extend([int(_lib.lib().ag_grouped_layer_maxima_floats()), int(_lib.lib().ag_grouped_comb_maxima_floats())])
    return _MAXIMA[0]


def _comb_maxima_floats():
    _maxima_floats()
    return _MAXIMA[1]


_OUT_MAXIMA_FLOATS = _lib.AG_MAX_GROUPS * 256


# ---------------------------------------------------------------------------------------------------------------------------------
# Frozen weights (round 5): inference runs the same parameters and styles frame after frame (the reference's --mode test, main_avatar.py:525-776).
# Their modulation, maxima and tile-blocked fp16 images are a seventh of a frame's device time; under no_grad a GroupedStyleUNets keeps them
# (include/ag_layers.h AgGroupedLayerArgs.packed_weights / weights_cached) for as long as no parameter's version counter, no style input and
# not the arithmetic mode changes.  ``_FROZEN``: the cache of the GroupedStyleUNets whose forward is running, or None (training, captures).
# ---------------------------------------------------------------------------------------------------------------------------------
_FROZEN = None


def frozen_weights_enabled() -> bool:
    return os.environ.get("AG_FROZEN_WEIGHTS", "1") != "0"


def _frozen_entry(site):
    """(entry dict, hit) of the layer call identified by ``site`` in the active cache, or (None, False)."""
    if _FROZEN is None:
        return None, False
    e = _FROZEN.get(site)
    if e is not None and e.get("_filled"):
        return e, True
    # (an entry counts as a hit only once the native call that fills it has succeeded: _frozen_commit.  An exception in between must not leave an
    # empty 'hit' behind for the next frame)
    e = _FROZEN[site] = {}
    return e, False


def _live_tensors(nets):
    """The parameter and buffer tensors the networks hold NOW (a replaced Parameter object is seen).  DualStyleUNet keeps a flat (holder module,
    leaf name) table of all its tensors: two dict lookups per tensor, 0.2 ms for the three networks' 1095 tensors against 1.45 ms for a
    parameters() + buffers() walk of the module tree -- per inference frame, where a millisecond is a tenth of the frame."""
    out = []
    for n in nets:
        slots = getattr(n, "_slots", None)
        if slots is None:
            out += list(n.parameters()) + list(n.buffers())
            continue
        for m, leaf in slots.values():
            t = m._parameters.get(leaf)
            out.append(t if t is not None else m._buffers[leaf])
        out += [b for b in n._buffers.values() if b is not None]          # the network's own FIR kernels
    return out


def _frozen_commit(e):
    if e is not None:
        e["_filled"] = True


def _handed_maxima(x):
    """The largest magnitudes of ``x`` as the grouped call that produced it left them (``out_maxima`` of include/ag_layers.h), or None.
    They travel as an attribute of the very tensor the producer returned: any operation in between yields another tensor without it."""
    m = getattr(x, "_ag_maxima", None)
    if m is None or not agc.needs_maxima() or not x.is_contiguous():
        return None
    if x._version != getattr(x, "_ag_maxima_version", -1):      # written in place since: the maxima are stale (too small a maximum would overflow fp16)
        return None
    return m


def _new_out_maxima(out):
    if not agc.needs_maxima():
        return None
    m = torch.empty(_OUT_MAXIMA_FLOATS, dtype=torch.float32, device=out.device)
    out._ag_maxima = m
    out._ag_maxima_version = out._version
    return m


def _merge_shared(stacked, owners, needs, view_shape=None):
    """Gradients of stacked members -> one gradient per DISTINCT parameter.  ``stacked`` [G, ...]: member g's gradient; ``owners[g]``: the parameter
    tensor it belongs to (several views of a pose run the colour network's tail once per view, all on the same parameters); ``needs[g]``.
    Returns G entries: the SUM over a parameter's members at its first member, None at the others (autograd then has nothing to accumulate: it
    used to add the members' gradients pairwise, 41 small additions per extra view of a training step, profiles/r05_per_view_breakdown_*.txt)."""
    G = len(owners)
    if stacked is None:
        return [None] * G
    slot, first = [], {}
    for g, t in enumerate(owners):
        slot.append(first.setdefault(t if isinstance(t, int) else t.data_ptr(), len(first)))
    def shaped(t):
        return t.view(view_shape) if view_shape is not None else t
    if len(first) == G:
        return [shaped(stacked[g]) if needs[g] else None for g in range(G)]
    R = len(first)
    if G % R == 0 and all(slot[g] == g % R for g in range(G)):            # view-major order (view, branch): one strided reduction
        summed = stacked.reshape((G // R, R) + tuple(stacked.shape[1:])).sum(0)
    elif G % R == 0 and all(slot[g] == g // (G // R) for g in range(G)):   # runs of equal length
        summed = stacked.reshape((R, G // R) + tuple(stacked.shape[1:])).sum(1)
    else:
        # mixed pattern (networks with one member beside a network with V views): per distinct parameter a strided-slice sum -- the members of a
        # parameter sit at an arithmetic progression (view-major order) -- or a gather + sum; a parameter with one member keeps its view
        # (one index_add_ over the stack ran 70 us per call on its generic kernel)
        members = {}
        for g in range(G):
            members.setdefault(slot[g], []).append(g)
        summed = {}
        for r, ms in members.items():
            if len(ms) == 1:
                summed[r] = stacked[ms[0]]
            elif all(ms[i + 1] - ms[i] == ms[1] - ms[0] for i in range(len(ms) - 1)):
                summed[r] = stacked[ms[0]:ms[-1] + 1:ms[1] - ms[0]].sum(0)
            else:
                summed[r] = stacked.index_select(0, _index(ms, stacked.device)).sum(0)
    seen, out = set(), []
    for g in range(G):
        if slot[g] in seen or not needs[g]:
            out.append(None)
        else:
            out.append(shaped(summed[slot[g]]))
        seen.add(slot[g])
    return out


def _scratch(nfloats, ws_bytes, dev):
    buf = torch.empty(nfloats * 4 + ws_bytes + 512, dtype=torch.uint8, device=dev)
    base = (buf.data_ptr() + 255) & ~255
    return buf, base, base + ((nfloats * 4 + 255) & ~255)


class _GroupedLayer(torch.autograd.Function):
    """G instances of a ConvLayer (``modulated`` False) or StyledConv as ONE native call each way.
    ``params`` = weights[G] (+ styles[G], noises[G], noise_weights[G] when modulated) + biases[G]."""

    @staticmethod
    def forward(ctx, G, shared, resample, modulated, scale, k_blur, x, *params):
        xm = _handed_maxima(x)
        x = x.contiguous()
        ws = [p.contiguous() for p in params[:G]]
        if modulated:
            styles = [p.contiguous() for p in params[G:2 * G]]
            noises, nws, biases = params[2 * G:3 * G], params[3 * G:4 * G], params[4 * G:5 * G]
        else:
            styles, noises, nws, biases = [None] * G, [None] * G, [None] * G, params[G:2 * G]
        dev = x.device
        a = _layer_args(G, x, ws[0], resample, modulated, scale, shared)
        oh, ow, f_fwd, _, wsb = _layer_sizes(a)
        out = torch.empty((G, a.Cout, oh, ow), dtype=torch.float32, device=dev)
        keep = None
        wnum = ws[0].numel()
        # frozen weights (inference): the modulated weights, the weight maxima and the packed image persist between frames
        fz, fz_hit = _frozen_entry(("layer", tuple(w.data_ptr() for w in ws), tuple(st.data_ptr() for st in styles) if modulated else (), G, resample))
        if modulated:
            keep = fz["keep"] if fz_hit else torch.empty(G * (wnum + a.Cout), dtype=torch.float32, device=dev)   # modulated weights, then the demodulation coefficients
            a.w_mod, a.demod = keep.data_ptr(), keep.data_ptr() + 4 * G * wnum
            for st in styles:
                if st.numel() != a.Cin:
                    raise RuntimeError("style must have one entry per input channel")
            for nz in noises:
                if nz is not None and nz.numel() != oh * ow:
                    raise RuntimeError("noise must be [1, 1, OH, OW]")
            _fill(a.style, styles)
            _fill(a.noise, noises)
            _fill(a.noise_weight, nws)
        elif resample:
            keep = torch.empty(((1 if shared else G), a.Cin, a.H + 1, a.W + 1), dtype=torch.float32, device=dev)     # the blurred input
            a.x_blur = keep.data_ptr()
        _fill(a.weight, ws)
        _fill(a.act_bias, biases)
        a.x, a.out = x.data_ptr(), out.data_ptr()
        a.k_blur = k_blur.data_ptr() if resample else None
        buf, a.scratch, a.workspace = _scratch(f_fwd, wsb, dev)
        a.workspace_bytes = wsb
        # operand maxima of the fp16 split form (include/ag_layers.h): written by this call, read by the backward
        mx = (fz["mx"] if fz_hit else torch.empty(_maxima_floats(), dtype=torch.float32, device=dev)) if a.k >= 3 else None
        a.operand_maxima = mx.data_ptr() if mx is not None else None
        a.x_maxima = xm.data_ptr() if xm is not None else None
        om = _new_out_maxima(out)
        a.out_maxima = om.data_ptr() if om is not None else None
        if fz is not None:
            packed = fz["packed"] if fz_hit else torch.empty(int(_lib.lib().ag_grouped_layer_packed_bytes(ctypes.byref(a))), dtype=torch.uint8, device=dev)
            a.packed_weights, a.weights_cached = packed.data_ptr(), int(fz_hit)
            fz.update(keep=keep if modulated else None, mx=mx, packed=packed)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().ag_grouped_layer_forward(ctypes.byref(a), _stream(dev)), "ag_grouped_layer_forward")
        _frozen_commit(fz)
        ctx.save_for_backward(x, out, keep, _flipped(k_blur) if resample else None, mx, xm, *ws, *styles, *noises, *nws, *biases)
        ctx.math = agc.get_math()
        ctx.cfg = (G, bool(shared), bool(resample), bool(modulated), float(scale))
        return out

    @staticmethod
    def backward(ctx, g):
        G, shared, resample, modulated, scale = ctx.cfg
        x, out, keep, k_flip, mx, xm = ctx.saved_tensors[:6]
        rest = ctx.saved_tensors[6:]
        ws, styles, noises, nws, biases = (rest[i * G:(i + 1) * G] for i in range(5))
        dev = x.device
        g = g.contiguous()
        a = _layer_args(G, x, ws[0], resample, modulated, scale, shared)
        _, _, _, f_bwd, wsb = _layer_sizes(a)
        nig = ctx.needs_input_grad
        nx = nig[6]
        pn = nig[7:]
        need_w = any(pn[:G])
        if modulated:
            need_s, need_nw, need_b = any(pn[G:2 * G]), any(pn[3 * G:4 * G]), any(pn[4 * G:5 * G])
        else:
            need_s, need_nw, need_b = False, False, any(pn[G:2 * G])
        wnum = ws[0].numel()
        if nx and shared and G > 1:
            raise RuntimeError("grouped layer: an input shared by several instances cannot receive a gradient")
        gx = torch.empty_like(x) if nx else None
        want_w = need_w or (modulated and need_s)
        gw = torch.empty((G,) + tuple(ws[0].shape), dtype=torch.float32, device=dev) if want_w else None
        gs = gbn = None
        _fill(a.weight, ws)
        _fill(a.act_bias, biases)
        if modulated:
            a.w_mod, a.demod = keep.data_ptr(), keep.data_ptr() + 4 * G * wnum
            _fill(a.style, styles)
            _fill(a.noise, noises)
            _fill(a.noise_weight, nws)
            if want_w:
                gs = torch.empty((G,) + tuple(styles[0].shape), dtype=torch.float32, device=dev)
        elif resample:
            a.x_blur = keep.data_ptr()
        has_bias = all(b is not None for b in biases)
        has_noise = modulated and all(n is not None and w is not None for n, w in zip(noises, nws))
        a.want_bias = int(bool(need_b and has_bias))
        a.want_noise_weight = int(bool(need_nw and has_noise))
        if a.want_bias or a.want_noise_weight:
            gbn = torch.empty((G, a.Cout + 1), dtype=torch.float32, device=dev)
        a.x, a.out, a.g_out = x.data_ptr(), out.data_ptr(), g.data_ptr()
        a.k_blur = k_flip.data_ptr() if resample else None
        a.g_x = gx.data_ptr() if gx is not None else None
        a.g_weight = gw.data_ptr() if gw is not None else None
        a.g_style = gs.data_ptr() if gs is not None else None
        a.g_bias_noise = gbn.data_ptr() if gbn is not None else None
        buf, a.scratch, a.workspace = _scratch(f_bwd, wsb, dev)
        a.workspace_bytes = wsb
        same_math = ctx.math == agc.get_math()                                                         # (a mode switch in between: retaken)
        a.operand_maxima = mx.data_ptr() if (mx is not None and same_math) else None
        a.x_maxima = xm.data_ptr() if (xm is not None and same_math) else None
        with _lib.on_device(dev):
            _lib.check(_lib.lib().ag_grouped_layer_backward(ctypes.byref(a), _stream(dev)), "ag_grouped_layer_backward")
        none = [None] * G
        # one gradient per distinct parameter (members that share a parameter -- the views of a pose -- are summed here, not by autograd)
        g_ws = _merge_shared(gw, ws, pn[:G])
        if not modulated:
            g_bs = _merge_shared(gbn[:, :a.Cout] if a.want_bias else None, biases, pn[G:2 * G])
            return (None, None, None, None, None, None, gx, *g_ws, *g_bs)
        g_bs = _merge_shared(gbn[:, :a.Cout] if a.want_bias else None, biases, pn[4 * G:5 * G])
        g_ss = _merge_shared(gs, styles, pn[G:2 * G])
        g_nws = _merge_shared(gbn[:, a.Cout:] if a.want_noise_weight else None, nws, pn[3 * G:4 * G], view_shape=tuple(nws[0].shape)) if has_noise else none
        return (None, None, None, None, None, None, gx, *g_ws, *g_ss, *none, *g_nws, *g_bs)


def grouped_conv_layer(x, weights, biases, k_blur, scale, downsample, shared=False):
    """G ConvLayers ([Blur +] EqualConv2d + FusedLeakyReLU, dual_styleunet.py:326-371): x [G, Cin, H, W] ([1, ...] when ``shared``)."""
    G = len(weights)
    return _GroupedLayer.apply(G, bool(shared), bool(downsample), False, float(scale), k_blur, x, *weights, *biases)


def grouped_styled_conv(x, weights, styles, noises, noise_weights, biases, k_blur, scale, upsample):
    """G StyledConvs (ModulatedConv2d + NoiseInjection + FusedLeakyReLU, dual_styleunet.py:225-313,570-604): x [G, Cin, H, W]."""
    G = len(weights)
    return _GroupedLayer.apply(G, False, bool(upsample), True, float(scale), k_blur, x, *weights, *styles, *noises, *noise_weights, *biases)


# ---------------------------------------------------------------------------------------------------------------------------------
# ToRGB
# ---------------------------------------------------------------------------------------------------------------------------------
def _rgb_args(G, x, w0, scale):
    a = _lib.AgGroupedToRgbArgs()
    a.G, a.Cout, a.Cin = G, int(w0.shape[-4]), int(w0.shape[-3])
    a.H, a.W = int(x.shape[2]), int(x.shape[3])
    if tuple(x.shape[:2]) != (G, a.Cin):
        raise RuntimeError("grouped ToRGB: input does not match the weights")
    a.scale = float(scale)
    return a


def _rgb_sizes(a):
    key = ("R", a.G, a.Cin, a.Cout, a.H, a.W)
    v = _SIZES.get(key)
    if v is None:
        L = _lib.lib()
        v = _SIZES[key] = (int(L.ag_grouped_to_rgb_scratch_floats(ctypes.byref(a), 0)), int(L.ag_grouped_to_rgb_scratch_floats(ctypes.byref(a), 1)),
                           int(L.ag_grouped_to_rgb_workspace_bytes(ctypes.byref(a))))
    return v


class _GroupedToRGB(torch.autograd.Function):
    """The ToRGB heads (dual_styleunet.py:607-633: modulated 1 x 1 convolution + bias + the wavelet skip) of ALL stacked members as one
    autograd node.  ``runs`` = [(start, end)] of members with the same head width (12 / 12 / 32 rows: network/avatar.py:34-36): one native
    call per run on its slice of the input, one output per run.  One node (and not one per run on a slice of ``x``) because the input
    gradient is then written by the runs straight into ONE tensor: slicing ``x`` outside would make autograd pad every run's gradient
    to the full size (a zero fill + a copy each) and add the padded tensors up (measured: ~1 ms of fills / copies / additions per step).
    ``rest`` = skips[len(runs)] (None where a run has none) + weights[M] + styles[M] + biases[M]."""

    @staticmethod
    def forward(ctx, runs, scale, k_up, x, *rest):
        x = x.contiguous()
        R, M = len(runs), int(x.shape[0])
        skips = [t.contiguous() if t is not None else None for t in rest[:R]]
        ws = [p.contiguous() for p in rest[R:R + M]]
        styles = [p.contiguous() for p in rest[R + M:R + 2 * M]]
        biases = [p.contiguous() for p in rest[R + 2 * M:R + 3 * M]]
        dev = x.device
        Cin, H, W = int(x.shape[1]), int(x.shape[2]), int(x.shape[3])
        outs, wms = [], []
        for (s, e), skip in zip(runs, skips):
            G = e - s
            a = _rgb_args(G, x[s:e], ws[s], scale)
            f_fwd, _, wsb = _rgb_sizes(a)
            out = torch.empty((G, a.Cout, H, W), dtype=torch.float32, device=dev)
            fz, fz_hit = _frozen_entry(("rgb", tuple(w.data_ptr() for w in ws[s:e]), tuple(st.data_ptr() for st in styles[s:e])))
            wm = fz["wm"] if fz_hit else torch.empty(G * a.Cout * Cin, dtype=torch.float32, device=dev)
            if fz is not None:
                fz["wm"] = wm
                a.weights_cached = int(fz_hit)
            _fill(a.weight, ws[s:e])
            _fill(a.style, styles[s:e])
            _fill(a.bias, biases[s:e])
            a.x, a.out, a.w_mod = x.data_ptr() + 4 * s * Cin * H * W, out.data_ptr(), wm.data_ptr()
            if skip is not None:
                if tuple(skip.shape) != (G, a.Cout, H // 2, W // 2):
                    raise RuntimeError("grouped ToRGB: skip must be [G, Cout, H / 2, W / 2]")
                a.skip, a.skip_taps = skip.data_ptr(), _skip_taps_host(k_up)
            buf, a.scratch, a.workspace = _scratch(f_fwd, wsb, dev)
            a.workspace_bytes = wsb
            with _lib.on_device(dev):
                _lib.check(_lib.lib().ag_grouped_to_rgb_forward(ctypes.byref(a), _stream(dev)), "ag_grouped_to_rgb_forward")
            _frozen_commit(fz)
            outs.append(out)
            wms.append(wm)
        ctx.save_for_backward(x, *wms, *ws, *styles)
        ctx.bias_owners = [b.data_ptr() for b in biases]      # the parameters' identity, for the backward's sums over shared heads
        ctx.cfg = (tuple(runs), float(scale), tuple(t is not None for t in skips), tuple(tuple(b.shape) for b in biases))
        ctx.k_up = k_up                       # a module buffer, not an output of this node
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        runs, scale, has_skip, bshapes = ctx.cfg
        R = len(runs)
        x = ctx.saved_tensors[0]
        M = int(x.shape[0])
        wms = ctx.saved_tensors[1:1 + R]
        ws, styles = ctx.saved_tensors[1 + R:1 + R + M], ctx.saved_tensors[1 + R + M:1 + R + 2 * M]
        dev = x.device
        Cin, H, W = int(x.shape[1]), int(x.shape[2]), int(x.shape[3])
        nig = ctx.needs_input_grad
        nx, nskips, pn = nig[3], nig[4:4 + R], nig[4 + R:]
        gx = torch.empty_like(x) if nx else None                     # every run writes its slice
        g_skips, g_ws, g_ss, g_bs = [None] * R, [None] * M, [None] * M, [None] * M
        for r, ((s, e), g) in enumerate(zip(runs, gs)):
            G = e - s
            g = g.contiguous()
            a = _rgb_args(G, x[s:e], ws[s], scale)
            _, f_bwd, wsb = _rgb_sizes(a)
            want_w = any(pn[s:e]) or any(pn[M + s:M + e])
            gw = torch.empty((G,) + tuple(ws[s].shape), dtype=torch.float32, device=dev) if want_w else None
            gst = torch.empty((G,) + tuple(styles[s].shape), dtype=torch.float32, device=dev) if want_w else None
            gskip = torch.empty((G, a.Cout, H // 2, W // 2), dtype=torch.float32, device=dev) if (has_skip[r] and nskips[r]) else None
            gb = plane_sums(g) if any(pn[2 * M + s:2 * M + e]) else None          # [G, Cout]: fixed slices, fixed order (ag_plane_sums)
            _fill(a.weight, ws[s:e])
            _fill(a.style, styles[s:e])
            a.x, a.w_mod, a.g_out = x.data_ptr() + 4 * s * Cin * H * W, wms[r].data_ptr(), g.data_ptr()
            a.g_x = gx.data_ptr() + 4 * s * Cin * H * W if gx is not None else None
            a.g_weight = gw.data_ptr() if gw is not None else None
            a.g_style = gst.data_ptr() if gst is not None else None
            if gskip is not None:
                a.g_skip, a.skip_taps = gskip.data_ptr(), _skip_taps_host(ctx.k_up)
            buf, a.scratch, a.workspace = _scratch(f_bwd, wsb, dev)
            a.workspace_bytes = wsb
            with _lib.on_device(dev):
                _lib.check(_lib.lib().ag_grouped_to_rgb_backward(ctypes.byref(a), _stream(dev)), "ag_grouped_to_rgb_backward")
            g_skips[r] = gskip
            # one gradient per distinct parameter of the run (the views of a pose share the heads' parameters)
            g_ws[s:e] = _merge_shared(gw, ws[s:e], pn[s:e])
            g_ss[s:e] = _merge_shared(gst, styles[s:e], pn[M + s:M + e])
            g_bs[s:e] = _merge_shared(gb, ctx.bias_owners[s:e], pn[2 * M + s:2 * M + e], view_shape=bshapes[s])
        return (None, None, None, gx, *g_skips, *g_ws, *g_ss, *g_bs)


def grouped_to_rgb(x, weights, styles, biases, skip, k_up, scale):
    """One run: the G = len(weights) heads of one width.  -> [G, Cout, H, W]."""
    G = len(weights)
    return _GroupedToRGB.apply(((0, G),), float(scale), k_up, x, skip, *weights, *styles, *biases)[0]


def grouped_to_rgb_runs(x, runs, weights, styles, biases, skips, k_up, scale):
    """All members' heads, ``runs`` = [(start, end)] of equal head width; skips[r] = the run's skip tensor or None.  -> one output per run."""
    return _GroupedToRGB.apply(tuple(runs), float(scale), k_up, x, *skips, *weights, *styles, *biases)


def _comb_args(begin, x, lev, w0, scale):
    a = _lib.AgGroupedCombArgs()
    a.M, a.N = int(x.shape[0]), int(lev.shape[0])
    a.C1, a.C2, a.Cout = int(x.shape[1]), int(lev.shape[1]), int(w0.shape[0])
    a.H, a.W = int(x.shape[2]), int(x.shape[3])
    if len(begin) != a.N + 1 or begin[0] != 0 or begin[-1] != a.M or tuple(lev.shape[2:]) != tuple(x.shape[2:]) or tuple(w0.shape[1:]) != (a.C1 + a.C2, 3, 3):
        raise RuntimeError("grouped comb convolution: shapes / member ranges do not match")
    for i, b in enumerate(begin):
        a.member_begin[i] = int(b)
    a.scale, a.slope, a.act_scale = float(scale), 0.2, _SQRT2
    return a


def _comb_sizes(a):
    key = ("C", a.M, a.N, a.C1, a.C2, a.Cout, a.H, a.W)
    v = _SIZES.get(key)
    if v is None:
        L = _lib.lib()
        v = _SIZES[key] = (int(L.ag_grouped_comb_scratch_floats(ctypes.byref(a), 0)), int(L.ag_grouped_comb_scratch_floats(ctypes.byref(a), 1)),
                           int(L.ag_grouped_comb_workspace_bytes(ctypes.byref(a))))
        if not v[0] or not v[2]:
            raise RuntimeError("grouped comb convolution: the library refused the shape")
    return v


class _GroupedComb(torch.autograd.Function):
    """The comb convolution of a decoder stage for the stacked members WITHOUT the concatenation (include/ag_layers.h AgGroupedCombArgs):
    conv(cat(x_m, lev_r), W_r) = conv(x_m, W_r[:, :C1]) + conv(lev_r, W_r[:, C1:]), the second half once per network.
    ``begin``: member ranges of the networks; ``rest`` = weights[N] (per network) + biases[M] (per member)."""

    @staticmethod
    def forward(ctx, begin, scale, x, lev, *rest):
        xm = _handed_maxima(x)
        x, lev = x.contiguous(), lev.contiguous()
        N, M = int(lev.shape[0]), int(x.shape[0])
        ws = [p.contiguous() for p in rest[:N]]
        bs = [p.contiguous() for p in rest[N:N + M]]
        dev = x.device
        a = _comb_args(begin, x, lev, ws[0], scale)
        f_fwd, _, wsb = _comb_sizes(a)
        out = torch.empty((M, a.Cout, a.H, a.W), dtype=torch.float32, device=dev)
        _fill(a.weight, ws)
        _fill(a.act_bias, bs)
        a.x, a.lev, a.out = x.data_ptr(), lev.data_ptr(), out.data_ptr()
        buf, a.scratch, a.workspace = _scratch(f_fwd, wsb, dev)
        a.workspace_bytes = wsb
        fz, fz_hit = _frozen_entry(("comb", tuple(w.data_ptr() for w in ws), tuple(begin)))
        mx = fz["mx"] if fz_hit else torch.empty(_comb_maxima_floats(), dtype=torch.float32, device=dev)      # operand maxima of the fp16 split form, kept for the backward
        a.operand_maxima = mx.data_ptr()
        a.x_maxima = xm.data_ptr() if xm is not None else None
        om = _new_out_maxima(out)
        a.out_maxima = om.data_ptr() if om is not None else None
        if fz is not None:
            if not fz_hit:
                fz.update(mx=mx, px=torch.empty(int(_lib.lib().ag_grouped_comb_packed_bytes(ctypes.byref(a), 0)), dtype=torch.uint8, device=dev),
                          pl=torch.empty(int(_lib.lib().ag_grouped_comb_packed_bytes(ctypes.byref(a), 1)), dtype=torch.uint8, device=dev))
            a.packed_x, a.packed_lev, a.weights_cached = fz["px"].data_ptr(), fz["pl"].data_ptr(), int(fz_hit)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().ag_grouped_comb_forward(ctypes.byref(a), _stream(dev)), "ag_grouped_comb_forward")
        _frozen_commit(fz)
        ctx.save_for_backward(x, lev, out, mx, xm, *ws, *bs)
        ctx.cfg = (tuple(begin), float(scale))
        ctx.math = agc.get_math()
        return out

    @staticmethod
    def backward(ctx, g):
        begin, scale = ctx.cfg
        x, lev, out, mx, xm = ctx.saved_tensors[:5]
        N, M = int(lev.shape[0]), int(x.shape[0])
        ws, bs = ctx.saved_tensors[5:5 + N], ctx.saved_tensors[5 + N:5 + N + M]
        dev = x.device
        g = g.contiguous()
        a = _comb_args(begin, x, lev, ws[0], scale)
        _, f_bwd, wsb = _comb_sizes(a)
        nig = ctx.needs_input_grad
        nx, nlev, pn = nig[2], nig[3], nig[4:]
        need_w, need_b = any(pn[:N]), any(pn[N:N + M])
        gx = torch.empty_like(x) if nx else None
        glev = torch.empty_like(lev) if nlev else None
        # the weight gradients come back in the parameters' own layout: the members of a network accumulate into its tensor's first C1 channels,
        # the level half fills the rest (the native call zeroes it) -- no sum over the members, no concatenation
        gW = torch.empty((N, a.Cout, a.C1 + a.C2, 3, 3), dtype=torch.float32, device=dev) if need_w else None
        gb = torch.empty((M, a.Cout), dtype=torch.float32, device=dev) if need_b else None
        _fill(a.weight, ws)
        _fill(a.act_bias, bs)
        a.x, a.lev, a.out, a.g_out = x.data_ptr(), lev.data_ptr(), out.data_ptr(), g.data_ptr()
        a.g_x = gx.data_ptr() if gx is not None else None
        a.g_lev = glev.data_ptr() if glev is not None else None
        a.g_weight = gW.data_ptr() if gW is not None else None
        same_math = ctx.math == agc.get_math()
        a.operand_maxima = mx.data_ptr() if same_math else None
        a.x_maxima = xm.data_ptr() if (xm is not None and same_math) else None
        a.g_bias = gb.data_ptr() if gb is not None else None
        buf, a.scratch, a.workspace = _scratch(f_bwd, wsb, dev)
        a.workspace_bytes = wsb
        with _lib.on_device(dev):
            _lib.check(_lib.lib().ag_grouped_comb_backward(ctypes.byref(a), _stream(dev)), "ag_grouped_comb_backward")
        g_ws = [gW[r] if (need_w and pn[r]) else None for r in range(N)]
        g_bs = _merge_shared(gb, bs, pn[N:N + M])                   # the views of a pose share the members' biases
        return (None, None, gx, glev, *g_ws, *g_bs)


class _SelectAddRows(torch.autograd.Function):
    """x[m] = out[src[m]] (+ vf on the rows [r0, r1)): the input of a view-dependent stage, where member m continues the shared state
    ``out[src[m]]`` and the colour members add their view-direction feature (dual_styleunet.py:881-883)."""

    @staticmethod
    def forward(ctx, out, vf, src, rows):
        ident = list(src) == list(range(out.shape[0]))
        # one pass (ag_select_add_rows): the row selection, the addition and -- ``vf`` may come at its own resolution -- the bilinear resize of
        # the members' view features (F.interpolate, dual_styleunet.py:881-883); an index_select, V resizes, a concatenation and an add_ before
        x = select_add_rows(out, src, vf, rows)
        ctx.cfg = (tuple(src), rows, ident, int(out.shape[0]), vf is not None, tuple(vf.shape[-2:]) if vf is not None else None)
        return x

    @staticmethod
    def backward(ctx, g):
        src, rows, ident, n_out, has_vf, vf_size = ctx.cfg
        g_out = g_vf = None
        if ctx.needs_input_grad[0]:
            if ident:
                g_out = g
            else:
                # row r of the shared state collects the gradients of the members that continued it: a copy (one member), a strided-slice sum
                # (the V views of a colour member sit at an arithmetic progression) -- one pass over the bytes instead of a zero fill plus
                # index_add_'s generic kernel (150 us per extra view on the [*, 128, 256, 256] state)
                g_out = torch.empty((n_out,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
                for r in range(n_out):
                    ms = [m for m, sr in enumerate(src) if sr == r]
                    if not ms:
                        g_out[r].zero_()
                    elif len(ms) == 1:
                        g_out[r].copy_(g[ms[0]])
                    elif all(ms[i + 1] - ms[i] == ms[1] - ms[0] for i in range(len(ms) - 1)):
                        torch.sum(g[ms[0]:ms[-1] + 1:ms[1] - ms[0]], 0, out=g_out[r])
                    else:
                        torch.sum(g.index_select(0, _index(ms, g.device)), 0, out=g_out[r])
        if has_vf and ctx.needs_input_grad[1]:
            g_vf = g[rows[0]:rows[1]]
            if vf_size != tuple(g.shape[-2:]):
                g_vf = bilinear_resize_backward(g_vf, vf_size)
        return g_out, g_vf, None, None


class _GroupedHaarMerge(torch.autograd.Function):
    """InverseHaarTransform (dual_styleunet.py:406-425) of G stacked tensors: [G, 4C, h, w] -> [G, C, 2h, 2w], one kernel."""

    @staticmethod
    def _run(x, matrix, merge):
        x = x.contiguous()
        G = int(x.shape[0])
        if merge:
            C, h, w = int(x.shape[1]) // 4, int(x.shape[2]), int(x.shape[3])
            out = torch.empty((G, C, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
        else:
            C, h, w = int(x.shape[1]), int(x.shape[2]) // 2, int(x.shape[3]) // 2
            out = torch.empty((G, 4 * C, h, w), dtype=torch.float32, device=x.device)
        m = (ctypes.c_float * 16)(*matrix)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ag_grouped_block2x2(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.cast(m, ctypes.c_void_p),
                                                      int(merge), G, C, h, w, _stream(x.device)), "ag_grouped_block2x2")
        return out

    @staticmethod
    def forward(ctx, x):
        return _GroupedHaarMerge._run(x, _HAAR_SYNTHESIS, True)

    @staticmethod
    def backward(ctx, g):
        # adjoint of a merge with B = a split with B^T (the Haar basis is orthonormal: B^T = the analysis matrix)
        bt = tuple(_HAAR_SYNTHESIS[4 * c + r] for r in range(4) for c in range(4))
        return _GroupedHaarMerge._run(g, bt, False)


class _SplitPairs(torch.autograd.Function):
    """[2 V, C, H, W] -> V tensors [1, 2 C, H, W] (the two branches of a (network, view) are neighbouring rows of one stacked image: their channel
    concatenation is a view).  As plain slices every one of the V outputs had its own SliceBackward -- a zero fill of the WHOLE stack, a copy and an
    addition per view, quadratic in V (0.1 ms per extra view at V = 3, 6 ms of a 16-view step); here the backward stacks the V gradients once."""

    @staticmethod
    def forward(ctx, img):
        V = img.shape[0] // 2
        ctx.shape = tuple(img.shape)
        pairs = img.view(V, 2 * img.shape[1], *img.shape[2:])
        return tuple(pairs[v:v + 1] for v in range(V))

    @staticmethod
    def backward(ctx, *gs):
        shape = ctx.shape
        V = shape[0] // 2
        ref = next(g for g in gs if g is not None)
        out = torch.empty((V, 2 * shape[1]) + shape[2:], dtype=ref.dtype, device=ref.device)
        for v, g in enumerate(gs):
            if g is None:
                out[v].zero_()
            else:
                out[v].copy_(g[0])
        return out.view(shape)


class _CatLevels(torch.autograd.Function):
    """Input of a decoder stage's comb convolution for the stacked members (dual_styleunet.py:877-879): ``cat([out, level], channel)`` with
    member m reading ``out[src[m]]`` (+ its view-direction feature, :881-883) and the encoder level of its network ``lev[net[m]]``.
    One buffer, filled by slice copies; the gradients are slice reads (sums where a source feeds several members)."""

    @staticmethod
    def forward(ctx, out, lev, vf, src, net, vf_rows):
        M = len(src)
        C1, C2 = int(out.shape[1]), int(lev.shape[1])
        buf = torch.empty((M, C1 + C2) + tuple(out.shape[2:]), dtype=out.dtype, device=out.device)
        ident = list(src) == list(range(out.shape[0]))
        pairs = M == 2 * lev.shape[0] and list(net) == [i // 2 for i in range(M)]
        if ident:
            buf[:, :C1].copy_(out)
        else:
            buf[:, :C1].copy_(out.index_select(0, _index(src, out.device)))
        one = list(net) == list(range(lev.shape[0]))
        if pairs:
            buf.view(M // 2, 2, C1 + C2, *out.shape[2:])[:, :, C1:].copy_(lev[:, None])
        elif one:
            buf[:, C1:].copy_(lev)
        else:
            buf[:, C1:].copy_(lev.index_select(0, _index(net, out.device)))
        if vf is not None:
            r0, r1 = vf_rows
            buf[r0:r1, :C1].add_(vf)
        ctx.cfg = (tuple(src), tuple(net), vf_rows, ident, pairs, C1, int(out.shape[0]), int(lev.shape[0]), vf is not None)
        return buf

    @staticmethod
    def backward(ctx, g):
        src, net, vf_rows, ident, pairs, C1, n_out, n_lev, has_vf = ctx.cfg
        M = g.shape[0]
        g_out = g_lev = g_vf = None
        if ctx.needs_input_grad[0]:
            if ident:
                g_out = g[:, :C1].contiguous()
            else:
                g_out = torch.zeros((n_out, C1) + tuple(g.shape[2:]), dtype=g.dtype, device=g.device)
                g_out.index_add_(0, _index(src, g.device), g[:, :C1])
        if ctx.needs_input_grad[1]:
            if pairs:
                g_lev = g.view(M // 2, 2, *g.shape[1:])[:, :, C1:].sum(1)
            elif list(net) == list(range(n_lev)):
                g_lev = g[:, C1:].contiguous()
            else:
                g_lev = torch.zeros((n_lev, g.shape[1] - C1) + tuple(g.shape[2:]), dtype=g.dtype, device=g.device)
                g_lev.index_add_(0, _index(net, g.device), g[:, C1:])
        if has_vf and ctx.needs_input_grad[2]:
            g_vf = g[vf_rows[0]:vf_rows[1], :C1].contiguous()
        return g_out, g_lev, g_vf, None, None, None


# ---------------------------------------------------------------------------------------------------------------------------------
# The three networks as one chain
# ---------------------------------------------------------------------------------------------------------------------------------
def _runs(values):
    """[(start, end)] of the maximal runs of equal consecutive values."""
    out, s = [], 0
    for i in range(1, len(values) + 1):
        if i == len(values) or values[i] != values[s]:
            out.append((s, i))
            s = i
    return out


class GroupedStyleUNets:
    """Evaluates several ``DualStyleUNet`` modules of the same trunk shape (their ``out_ch`` may differ) on ONE conditioning image as grouped
    launch chains.  The modules keep their parameters; this object holds no state of its own besides shape tables."""

    def __init__(self, nets):
        self.nets = list(nets)
        n0 = self.nets[0]
        for n in self.nets[1:]:
            if (n.inp_size, n.inp_ch, n.out_size, n.style_dim, n.enc, n.dec, n.n_comb, n.num_layers) != \
               (n0.inp_size, n0.inp_ch, n0.out_size, n0.style_dim, n0.enc, n0.dec, n0.n_comb, n0.num_layers):
                raise ValueError("grouped execution needs networks of one trunk shape")
        if len(self.nets) * 2 > _lib.AG_MAX_GROUPS:
            raise ValueError("too many networks for one group")
        if not (n0.VIEW_STAGE + 1 < n0.n_comb and n0.VIEW_STAGE + 1 < len(n0.dec)):
            raise ValueError("grouped execution expects the view-dependent stage to start with a comb convolution")

    @staticmethod
    def supported(nets) -> bool:
        try:
            GroupedStyleUNets(nets)
            return True
        except ValueError:
            return False

    # ---- layers over a list of (network, parameter prefix) ----------------------------------------------------------------------
    @staticmethod
    def _conv_layer(x, nets, prefix, downsample=False, shared=False):
        base = 1 if downsample else 0
        ws = [n._p(f"{prefix}.{base}.weight") for n in nets]
        bs = [n._p(f"{prefix}.{base + 1}.bias") for n in nets]
        k = ws[0].shape[-1]
        return grouped_conv_layer(x, ws, bs, nets[0]._k_blur, 1 / math.sqrt(ws[0].shape[1] * k * k), downsample, shared)

    def _encode(self, x):
        """Pose map [1, inp_ch, S, S] -> levels, finest first, each [n_nets, C, h, w]  (dual_styleunet.py:854-864 for every network)."""
        nets, n0 = self.nets, self.nets[0]
        out = self._conv_layer(x, nets, "conv_in", downsample=True, shared=True)
        levels = [out]
        img = x
        for n, _, _ in n0.enc:
            img = upfirdn2d_nchw(img, n0._k_blur, down=2, pad=(1, 1))            # Downsample: the same image pyramid for all networks
            out = self._conv_layer(img, nets, f"from_rgbs.{n}.conv", shared=True) + out
            out = self._conv_layer(out, nets, f"cond_convs.{n}.conv1")
            out = self._conv_layer(out, nets, f"cond_convs.{n}.conv2", downsample=True)
            levels.append(out)
        return levels

    def _stage(self, n, members, styles, noises, out, skips, levels, src=None, vf=None, vf_rows=None):
        """One decoder stage for the stacked members.  members: [(network index, branch)]; styles[m]: {prefix: style} of member m;
        out: [M', C, h, w] of the previous stage (None at stage 0), member m continues ``out[src[m]]``; skips: {(start, end): tensor}
        per run of equal ToRGB width."""
        nets, n0 = self.nets, self.nets[0]
        M = len(members)
        mnets = [nets[i] for i, _ in members]
        net_idx = [i for i, _ in members]
        if n == 0:
            o3 = self._conv_layer(levels[-1], nets, f"comb_convs.{n0.n_comb - 1}")     # branch-independent: once per network (:873 runs it per branch)
            out = o3.index_select(0, _index(net_idx, o3.device))
        elif n < n0.n_comb:
            src = list(range(M)) if src is None else list(src)
            if _COMB_SPLIT and net_idx == sorted(net_idx):
                # the comb convolution without the concatenation: the encoder-level half once per network (ag_grouped_comb_*)
                if vf is not None or src != list(range(out.shape[0])):
                    out = _SelectAddRows.apply(out, vf, tuple(src), vf_rows)
                used = sorted(set(net_idx))
                begin = [net_idx.index(r) for r in used] + [M]
                lev = levels[-1 - n]
                if used != list(range(lev.shape[0])):
                    lev = lev.index_select(0, _index(used, lev.device))
                prefix = f"comb_convs.{n0.n_comb - 1 - n}"
                wts = [nets[r]._p(f"{prefix}.0.weight") for r in used]
                out = _GroupedComb.apply(tuple(begin), 1 / math.sqrt(wts[0].shape[1] * 9), out, lev, *wts, *[net._p(f"{prefix}.1.bias") for net in mnets])
            else:
                if vf is not None and vf.shape[-2:] != out.shape[-2:]:
                    vf = bilinear_resize(vf, out.shape[-2:])
                cat = _CatLevels.apply(out, levels[-1 - n], vf, tuple(src), tuple(net_idx), vf_rows)
                out = self._conv_layer(cat, mnets, f"comb_convs.{n0.n_comb - 1 - n}")
        pre = [f"convs{b}.{2 * n}" for _, b in members]
        w = [net._p(f"{p}.conv.weight") for net, p in zip(mnets, pre)]
        k = w[0].shape[-1]
        out = grouped_styled_conv(out, w, [styles[m][f"{p}.conv"] for m, p in enumerate(pre)], [noises[i][2 * n] for i in net_idx],
                                  [net._p(f"{p}.noise.weight") for net, p in zip(mnets, pre)],
                                  [net._p(f"{p}.activate.bias") for net, p in zip(mnets, pre)], n0._k_blur_up, 1 / math.sqrt(w[0].shape[2] * k * k), True)
        pre = [f"convs{b}.{2 * n + 1}" for _, b in members]
        w = [net._p(f"{p}.conv.weight") for net, p in zip(mnets, pre)]
        out = grouped_styled_conv(out, w, [styles[m][f"{p}.conv"] for m, p in enumerate(pre)], [noises[i][2 * n + 1] for i in net_idx],
                                  [net._p(f"{p}.noise.weight") for net, p in zip(mnets, pre)],
                                  [net._p(f"{p}.activate.bias") for net, p in zip(mnets, pre)], None, 1 / math.sqrt(w[0].shape[2] * k * k), False)
        # ToRGB: one native call per run of members with the same head width, all runs in ONE autograd node
        runs = _runs([net.out_ch for net in mnets])
        pr = [f"to_rgbs{b}.{n}" for _, b in members]
        wr = [net._p(f"{p}.conv.weight") for net, p in zip(mnets, pr)]
        outs = grouped_to_rgb_runs(out, runs, wr, [styles[m][f"{p}.conv"] for m, p in enumerate(pr)],
                                   [net._p(f"{p}.bias").reshape(-1) for net, p in zip(mnets, pr)],
                                   [skips.get(r) if skips else None for r in runs], n0._k_blur_up, 1 / math.sqrt(wr[0].shape[2]))
        new_skips = dict(zip(runs, outs))
        return out, new_skips

    def forward(self, styles, x, view_features=None, frozen=False):
        """styles[i]: the style vector [1, style_dim] of network i; x: [1, inp_ch, S, S]; view_features: {network index: (f1, f2)} or
        {network index: [(f1, f2), ...]} for several views of the pose (the view-dependent stages then run once per view).
        Returns per network the image [1, 2 out_ch, S', S'] (= ``DualStyleUNet.forward(...)[0]``), or a list of them per view.
        ``frozen``: the caller's promise that ``styles`` are PERSISTENT tensors (module buffers, not per-call temporaries whose address and version
        a later temporary may repeat): under no_grad the weights' modulation, maxima and packed images are then kept between calls (``_FROZEN``)."""
        global _FROZEN
        frozen = self._frozen_cache_for(styles) if frozen else None
        prev, _FROZEN = _FROZEN, frozen
        try:
            return self._forward(styles, x, view_features)
        finally:
            _FROZEN = prev

    def _frozen_cache_for(self, styles):
        """The frozen-weight cache of this object (a dict the layer calls fill and read, see ``_FROZEN``) when the call is an inference call --
        no autograd, no hipGraph capture, one stream -- and None otherwise.  It is emptied whenever a parameter or buffer of the networks, a style
        input or the arithmetic mode has changed since it was filled (version counters; ``optim.FusedAdam`` bumps them like torch's optimizers)."""
        if torch.is_grad_enabled() or not frozen_weights_enabled() or torch.cuda.is_current_stream_capturing() or \
           int(os.environ.get("AG_GROUPED_STREAMS", "1")) != 1:
            return None
        # the tensor list is rebuilt on every call (a REPLACED Parameter object must be seen) and the token is a tuple, not a sum (sums of addresses
        # and versions can collide); a non-contiguous parameter or style would be cached under the address of a per-call temporary: no cache then
        tensors = _live_tensors(self.nets)
        if not all(t.is_contiguous() for t in tensors) or not all(s.is_contiguous() for s in styles):
            return None
        token = (agc.get_math(), tuple((s.data_ptr(), s._version) for s in styles), tuple((t.data_ptr(), t._version) for t in tensors))
        if getattr(self, "_frozen_token", None) != token:
            self._frozen_token, self._frozen_cache = token, {}
        return self._frozen_cache

    def invalidate_frozen(self):
        """Drop the frozen-weight cache (after writing parameters in a way their version counters do not see, e.g. through ``.data``)."""
        self._frozen_token, self._frozen_cache = None, {}

    def _styles(self, i, b, stages, lat):
        """``nets[i]._stage_styles(b, stages, lat[i])``; frozen weights: the SAME tensors frame after frame (the layer calls' cache is keyed on them)."""
        if _FROZEN is None:
            return self.nets[i]._stage_styles(b, stages, lat[i])
        key = ("styles", i, b, tuple(stages))
        if key not in _FROZEN:
            _FROZEN[key] = self.nets[i]._stage_styles(b, stages, lat[i])
        return _FROZEN[key]

    def _forward(self, styles, x, view_features=None):
        nets = self.nets
        view_features = dict(view_features or {})
        # the mapping networks of all networks: one native launch per layer (styleunet.latents_of); fixed noise buffers (randomize_noise False)
        if _FROZEN is not None and ("lat",) in _FROZEN:
            lat = _FROZEN[("lat",)]
        else:
            lat = latents_of(nets, list(styles))
            lat = [w[:, 0] if w.dim() == 3 else w for w in lat]
            if _FROZEN is not None:
                _FROZEN[("lat",)] = lat
        noises = [n._latent_and_noise([w], True, None, False)[1] for n, w in zip(nets, lat)]
        levels = self._encode(x)
        for i, v in view_features.items():
            if isinstance(v, (list, tuple)) and len(v) == 0:
                raise ValueError(f"grouped networks: network {i} was given an empty list of views (pass None / omit it for no view features)")
        multi = {i: isinstance(v, (list, tuple)) and len(v) > 0 and isinstance(v[0], (list, tuple)) for i, v in view_features.items()}
        views = {i: (list(view_features[i]) if multi[i] else [view_features[i]]) for i in view_features}
        # AG_GROUPED_STREAMS=2: the decoders as two chains (branch 1 / branch 2 of every network, G = 3 each) on two HIP streams
        # instead of one chain with G = 6 -- twice the decoder launches, but the latency-bound layers at <= 32^2 overlap
        nstreams = int(os.environ.get("AG_GROUPED_STREAMS", "1"))
        if nstreams == 2 and not torch.cuda.is_current_stream_capturing() and os.environ.get("AG_SINGLE_STREAM") != "1":
            cur = torch.cuda.current_stream()
            if getattr(self, "_side", None) is None or self._side.device != cur.device:
                self._side = torch.cuda.Stream(cur.device)
                quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
                if quiet is not None:
                    quiet(False)
            side = self._side
            side.wait_stream(cur)
            shared = list(levels) + list(lat) + [f for vs in views.values() for fb in vs if fb is not None for f in fb if f is not None]
            for t in shared:
                t.record_stream(side)
            with torch.cuda.stream(side):
                r2 = self._decode([(i, 2) for i in range(len(nets))], lat, noises, levels, views)
            results = self._decode([(i, 1) for i in range(len(nets))], lat, noises, levels, views)
            cur.wait_stream(side)
            for t in r2.values():
                t.record_stream(cur)
            results.update(r2)
        else:
            results = self._decode([(i, b) for i in range(len(nets)) for b in (1, 2)], lat, noises, levels, views)
        outs = []
        for i in range(len(nets)):
            nv = len(views.get(i, [None]))
            per_view = []
            for v in range(nv):
                if (i, v, "pair") in results:
                    per_view.append(results[(i, v, "pair")])
                    continue
                a, bb = results[(i, v, 1)], results[(i, v, 2)]
                # the two branches of a (network, view) are neighbours in one stacked tensor: their channel concatenation is a view
                if a._base is not None and a._base is bb._base and a.data_ptr() + a.numel() * 4 == bb.data_ptr():
                    base = a._base
                    off = (a.data_ptr() - base.data_ptr()) // (4 * a.numel())
                    per_view.append(base[off:off + 2].reshape(1, 2 * a.shape[1], *a.shape[2:]))
                else:
                    per_view.append(torch.cat([a, bb], 1))
            outs.append(per_view if multi.get(i, False) else per_view[0])
        return outs

    def _decode(self, members, lat, noises, levels, views):
        """The decoders of ``members`` [(network index, branch)] as one chain -> {(network, view, branch): image [1, out_ch, S', S']}."""
        nets, n0 = self.nets, self.nets[0]
        shared_stages = list(range(0, min(n0.VIEW_STAGE + 1, len(n0.dec))))
        tail_stages = list(range(n0.VIEW_STAGE + 1, len(n0.dec)))
        st = [self._styles(i, b, shared_stages, lat) for i, b in members]
        out, skips = None, None
        for n in shared_stages:
            out, skips = self._stage(n, members, st, noises, out, skips, levels)
        # view-dependent tail: every member once, the members of a network with V views V times
        tail = []                                                      # (network, branch, source member, view number)
        for m, (i, b) in enumerate(members):
            if m and members[m - 1][0] == i:
                continue                                               # the members of network i are handled together below
            ms = [(mm, bb) for mm, (ii, bb) in enumerate(members) if ii == i]
            for v in range(len(views.get(i, [None]))):
                for mm, bb in ms:
                    tail.append((i, bb, mm, v))
        results = {}
        step = max(2, min(_lib.AG_MAX_GROUPS, int(os.environ.get("AG_GROUPED_TAIL_CHUNK", _lib.AG_MAX_GROUPS))))
        # the tail stages' styles depend on (network, branch) only: one modulation GEMM per pair, shared by all views and chunks
        tail_styles = {ib: self._styles(ib[0], ib[1], tail_stages, lat) for ib in dict.fromkeys((i, b) for i, b, _, _ in tail)}
        for c0 in range(0, len(tail), step):
            chunk = tail[c0:c0 + step]
            tm = [(i, b) for i, b, _, _ in chunk]
            src = [m for _, _, m, _ in chunk]
            tst = [tail_styles[ib] for ib in tm]
            # view features: the members that have one must be one contiguous run (they are: members are ordered by network)
            rows = [r for r, (i, b, _, v) in enumerate(chunk) if views.get(i) and views[i][v] is not None and views[i][v][b - 1] is not None]
            vf = None
            if rows:
                if rows != list(range(rows[0], rows[-1] + 1)):
                    raise RuntimeError("grouped networks: view features must belong to consecutive members")
                # the members' view features, stacked at THEIR resolution (8 MB each at 128 planes 128^2): the stage's input kernel resizes them
                # (F.interpolate(..., mode="bilinear"), dual_styleunet.py:881-883) while it adds them
                feats = [views[chunk[r][0]][chunk[r][3]][chunk[r][1] - 1] for r in rows]
                if any(f.shape != feats[0].shape for f in feats):
                    feats = [f if f.shape[-2:] == out.shape[-2:] else bilinear_resize(f, out.shape[-2:]) for f in feats]
                vf = torch.cat(feats, 0) if len(feats) > 1 else feats[0]
            o, sk = out, None
            for j, n in enumerate(tail_stages):
                if j == 0:
                    # skips of the chunk's members, per run of equal head width
                    if tm == list(members):
                        sk = skips                                          # one view: the stacked skips as they are
                    else:
                        sk = {}
                        for (s, e) in _runs([nets[i].out_ch for i, _ in tm]):
                            keys = [next(k for k in skips if k[0] <= src[r] < k[1]) for r in range(s, e)]
                            if all(k == keys[0] for k in keys):
                                # all members of the run continue rows of ONE stacked skip tensor: a row selection whose backward sums the views'
                                # gradients per row in one pass (slices + cat left a zero fill, a copy and an addition per member to autograd)
                                rows_sel = tuple(src[r] - keys[0][0] for r in range(s, e))
                                src_t = skips[keys[0]]
                                sk[(s, e)] = src_t if rows_sel == tuple(range(src_t.shape[0])) else _SelectAddRows.apply(src_t, None, rows_sel, None)
                                continue
                            parts = [skips[k][src[r] - k[0]:src[r] - k[0] + 1] for r, k in zip(range(s, e), keys)]
                            sk[(s, e)] = torch.cat(parts, 0) if len(parts) > 1 else parts[0]
                    o, sk = self._stage(n, tm, tst, noises, o, sk, levels, src=src, vf=vf, vf_rows=(rows[0], rows[-1] + 1) if rows else None)
                else:
                    o, sk = self._stage(n, tm, tst, noises, o, sk, levels)
            for (s, e), t in sk.items():
                img = _GroupedHaarMerge.apply(t)                             # [e - s, out_ch, S', S']
                rows = [chunk[r] for r in range(s, e)]
                paired = (e - s) % 2 == 0 and all(rows[2 * k][0] == rows[2 * k + 1][0] and rows[2 * k][3] == rows[2 * k + 1][3] and
                                                  (rows[2 * k][1], rows[2 * k + 1][1]) == (1, 2) for k in range((e - s) // 2))
                if paired and img.is_contiguous():
                    # branch 1 | branch 2 of every (network, view) of the run: the forward's channel concatenation, one autograd node for all of them
                    for k, pair in enumerate(_SplitPairs.apply(img)):
                        results[(rows[2 * k][0], rows[2 * k][3], "pair")] = pair
                    continue
                for r in range(s, e):
                    i, b, _, v = chunk[r]
                    results[(i, v, b)] = img[r - s:r - s + 1]
        return results
