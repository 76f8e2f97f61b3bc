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
    with _lib.on_device(x.device):
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
    with _lib.on_device(input.device):
        _lib.check(_lib.lib().ag_upfirdn2d(_p(out), _p(x), _p(k), x.shape[0], in_h, in_w, kh, kw, up_x, up_y, down_x, down_y,
                                           pad_x0, pad_x1, pad_y0, pad_y1, _stream(input.device)), "ag_upfirdn2d")
    if minor != 1:
        out = out.reshape(major, minor, out_h, out_w).permute(0, 2, 3, 1).contiguous()
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Differentiable NCHW front ends (what network/styleunet/fused_act.py:79-138 and upfirdn2d.py:35-184 wrap around the
# two raw ops).  First-order gradients only: the avatar trainer has no gradient-penalty term, so the reference's
# double-backward Functions (fused_act.py:35-76, upfirdn2d.py:35-103) are never differentiated on the product path.
# ---------------------------------------------------------------------------------------------------------------------
_EMPTY = {}
_FLIPPED = {}


def _flipped(kernel):
    """FIR kernel flipped in both axes (the backward of an upfirdn2d), cached per kernel TENSOR and version: the network's three FIR
    kernels are constants, and a flip is one more small launch per upfirdn2d backward otherwise.  The entry holds the source tensor
    itself and is only used while it is that very object: a key made of the address alone would be handed to another kernel once the
    first is freed and the allocator re-uses its block (a second model's blur kernel with another gain, a temporary passed to
    ``upfirdn2d``) -- stale flipped taps, silently wrong gradients."""
    key = (kernel.data_ptr(), kernel._version, tuple(kernel.shape))
    e = _FLIPPED.get(key)
    if e is None or e[0] is not kernel:
        if len(_FLIPPED) > 64:
            _FLIPPED.clear()
        e = _FLIPPED[key] = (kernel, torch.flip(kernel, [0, 1]).contiguous())
    return e[1]


def _empty(dev):
    e = _EMPTY.get(dev)
    if e is None:
        e = _EMPTY[dev] = torch.empty(0, dtype=torch.float32, device=dev)
    return e


class _FusedLeakyReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, negative_slope, scale):
        e = _empty(x.device)
        out = fused_bias_act(x.contiguous(), bias if bias is not None else e, e, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.cfg = (bias is not None, negative_slope, scale)
        return out

    @staticmethod
    def backward(ctx, gy):
        (out,) = ctx.saved_tensors
        has_bias, slope, scale = ctx.cfg
        e = _empty(gy.device)
        # the sign of the saved OUTPUT selects the slope (act=3, grad=1: fused_bias_act_kernel.cu:41)
        gx = fused_bias_act(gy.contiguous(), e, out, 3, 1, slope, scale)
        gb = None
        if has_bias:
            dims = [0] + list(range(2, gx.dim()))
            gb = gx.sum(dims)
        return gx, gb, None, None


def fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    """leaky_relu(input + bias[None, :, None, None]) * scale  (fused_act.py:117-138)."""
    return _FusedLeakyReLU.apply(input, bias, float(negative_slope), float(scale))


def _out_size(n, up, down, p0, p1, k):
    return (n * up + p0 + p1 - k + down) // down


class _UpFirDn2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, up, down, pad):
        n, c, h, w = x.shape
        kh, kw = kernel.shape
        px0, px1, py0, py1 = pad
        oh, ow = _out_size(h, up[1], down[1], py0, py1, kh), _out_size(w, up[0], down[0], px0, px1, kw)
        y = upfirdn2d(x.reshape(n * c, h, w, 1).contiguous(), kernel, up[0], up[1], down[0], down[1], px0, px1, py0, py1)
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, pad, (n, c, h, w), (oh, ow))
        return y.view(n, c, oh, ow)

    @staticmethod
    def backward(ctx, gy):
        (kernel,) = ctx.saved_tensors
        up, down, (px0, px1, py0, py1), (n, c, h, w), (oh, ow) = ctx.cfg
        kh, kw = kernel.shape
        # adjoint of (zero-stuff by up, pad, FIR, decimate by down) = the same pipeline with up/down exchanged, the
        # kernel flipped, and pads chosen so that the output has the input's size (upfirdn2d.py:131-136)
        gx0, gy0 = kw - px0 - 1, kh - py0 - 1
        gx1 = w * up[0] - ow * down[0] + px0 - up[0] + 1
        gy1 = h * up[1] - oh * down[1] + py0 - up[1] + 1
        g = upfirdn2d(gy.reshape(n * c, oh, ow, 1).contiguous(), _flipped(kernel), down[0], down[1],
                      up[0], up[1], gx0, gx1, gy0, gy1)
        return g.view(n, c, h, w), None, None, None, None


def upfirdn2d_nchw(input, kernel, up=1, down=1, pad=(0, 0)):
    """``upfirdn2d(input [N,C,H,W], kernel, up, down, pad)`` of upfirdn2d.py:168-184 (pad of 2 = same for x and y)."""
    up = (up, up) if isinstance(up, int) else tuple(up)
    down = (down, down) if isinstance(down, int) else tuple(down)
    pad = tuple(pad)
    if len(pad) == 2:
        pad = (pad[0], pad[1], pad[0], pad[1])
    return _UpFirDn2d.apply(input, kernel, up, down, pad)


_PARTIALS = {}


def _partial_floats(kind, a, b):
    """Floats of scratch the deterministic parameter-gradient reductions need (a function of the shape: asked once)."""
    key = (kind, a, b)
    n = _PARTIALS.get(key)
    if n is None:
        L = _lib.lib()
        n = _PARTIALS[key] = max(1, int(L.ag_noise_bias_act_partial_floats(a, b) if kind == "nba" else L.ag_modulate_weight_partial_floats(a, b)))
    return n


class _NoiseBiasAct(torch.autograd.Function):
    """StyledConv tail ``lrelu(x + w * noise + bias) * sqrt(2)`` in one kernel each way (include/ag_styleunet.h)."""

    @staticmethod
    def forward(ctx, x, noise, noise_weight, bias, slope, scale):
        if x.dim() != 4 or x.shape[0] != 1 or not x.is_cuda or x.dtype != torch.float32:
            raise RuntimeError("noise_bias_act: float32 GPU tensor [1, C, H, W]")
        x = x.contiguous()
        C, HW = int(x.shape[1]), int(x.shape[2] * x.shape[3])
        if noise is not None:
            noise = noise.contiguous()
            if noise.numel() != HW:
                raise RuntimeError("noise must be [1, 1, H, W]")
        y = torch.empty_like(x)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ag_noise_bias_act_forward(_p(y), _p(x), _p(noise), _p(noise_weight), _p(bias), C, HW,
                                                            float(slope), float(scale), _stream(x.device)),
                       "ag_noise_bias_act_forward")
        ctx.save_for_backward(y, noise)
        ctx.cfg = (C, HW, float(slope), float(scale), noise_weight is not None and noise is not None, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, noise = ctx.saved_tensors
        C, HW, slope, scale, has_nw, has_bias = ctx.cfg
        gy = gy.contiguous()
        gx = torch.empty_like(y)
        # frozen parameters (the VGG trunk of LPIPS): no reduction, no buffer
        want_b, want_w = has_bias and ctx.needs_input_grad[3], has_nw and ctx.needs_input_grad[2]
        gb = torch.empty(C, dtype=torch.float32, device=y.device) if want_b else None
        gw = torch.empty(1, dtype=torch.float32, device=y.device) if want_w else None
        part = None
        if want_b or want_w:                        # per-workgroup partial sums of the deterministic two-stage reductions
            part = torch.empty(_partial_floats("nba", C, HW), dtype=torch.float32, device=y.device)
        with _lib.on_device(y.device):
            _lib.check(_lib.lib().ag_noise_bias_act_backward(_p(gx), _p(gy), _p(y), _p(noise) if gw is not None else None, _p(gb), _p(gw),
                                                             _p(part), C, HW, slope, scale, _stream(y.device)),
                       "ag_noise_bias_act_backward")
        return gx, None, gw, gb, None, None


def noise_bias_act(x, noise, noise_weight, bias, negative_slope=0.2, scale=2 ** 0.5):
    """``fused_leaky_relu(x + noise_weight * noise, bias)`` (NoiseInjection + FusedLeakyReLU) without the intermediate
    tensors; ``noise`` / ``bias`` may be None.  No gradient flows to ``noise`` (a fixed buffer or a fresh random draw)."""
    return _NoiseBiasAct.apply(x, noise, noise_weight if noise is not None else None, bias, float(negative_slope), float(scale))


class _ModulateWeight(torch.autograd.Function):
    """``(scale * W) * style`` with optional demodulation (ModulatedConv2d, fused branch) in one kernel each way."""

    @staticmethod
    def forward(ctx, weight, style, scale, demodulate, transposed):
        w = weight.contiguous()
        st = style.contiguous()
        Co, Ci, K2 = int(w.shape[-4]), int(w.shape[-3]), int(w.shape[-2] * w.shape[-1])
        k = int(w.shape[-1])
        if st.numel() != Ci or not w.is_cuda or w.dtype != torch.float32:
            raise RuntimeError("modulate_weight: weight [.., Co, Ci, k, k] float32 on the GPU, style with Ci entries")
        out = torch.empty((Ci, Co, k, k) if transposed else (Co, Ci, k, k), dtype=torch.float32, device=w.device)
        dcoef = torch.empty(Co, dtype=torch.float32, device=w.device) if demodulate else None
        with _lib.on_device(w.device):
            _lib.check(_lib.lib().ag_modulate_weight_forward(_p(out), _p(dcoef), _p(w), _p(st), float(scale), int(demodulate), Co, Ci,
                                                             K2, int(transposed), _stream(w.device)), "ag_modulate_weight_forward")
        ctx.save_for_backward(w, st, dcoef)
        ctx.cfg = (float(scale), bool(demodulate), bool(transposed), Co, Ci, K2, weight.shape, style.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        w, st, dcoef = ctx.saved_tensors
        scale, demodulate, transposed, Co, Ci, K2, wshape, sshape = ctx.cfg
        g = g.contiguous()
        dW = torch.empty_like(w)
        ds = torch.empty(Ci, dtype=torch.float32, device=w.device)
        part = torch.empty(_partial_floats("mod", Co, Ci), dtype=torch.float32, device=w.device)
        with _lib.on_device(w.device):
            _lib.check(_lib.lib().ag_modulate_weight_backward(_p(dW), _p(ds), _p(part), _p(g), _p(w), _p(st), _p(dcoef), scale, int(demodulate),
                                                              Co, Ci, K2, int(transposed), _stream(w.device)),
                       "ag_modulate_weight_backward")
        return dW.view(wshape), ds.view(sshape), None, None, None


def modulate_weight(weight, style, scale, demodulate=True, transposed=False):
    """weight [1, Co, Ci, k, k] or [Co, Ci, k, k], style [1, Ci] -> modulated (and demodulated) conv weight [Co, Ci, k, k]
    (``transposed``: [Ci, Co, k, k] for conv_transpose2d)."""
    return _ModulateWeight.apply(weight, style, float(scale), bool(demodulate), bool(transposed))


# Haar analysis matrix A[band][p], p = 2*dy + dx over a 2x2 input block, bands (ll, lh, hl, hh) in the order HaarTransform
# concatenates them.  From upfirdn2d(x, k_b, down=2) = correlation with the FLIPPED kernel: out[i][j] = sum k_b[1-dy][1-dx] *
# x[2i+dy][2j+dx], with the kernels of get_haar_wavelet (dual_styleunet.py:374-384): ll = +.5 everywhere,
# lh = [[-.5,-.5],[.5,.5]], hl = [[-.5,.5],[-.5,.5]], hh = [[.5,-.5],[-.5,.5]].
_HAAR_ANALYSIS = (0.5, 0.5, 0.5, 0.5,
                  0.5, 0.5, -0.5, -0.5,
                  0.5, -0.5, 0.5, -0.5,
                  0.5, -0.5, -0.5, 0.5)
# Synthesis B[p][band]: upfirdn2d(y_b, k'_b, up=2, pad=(1,0,1,0)) puts k'_b[dy][dx] * y_b[i][j] at (2i+dy, 2j+dx), with
# k' = (ll, -lh, -hl, hh) (InverseHaarTransform, :406-425).  The Haar basis is orthonormal: B = A^T.
_HAAR_SYNTHESIS = tuple(_HAAR_ANALYSIS[4 * b + p] for p in range(4) for b in range(4))


def _transpose4(m):
    return tuple(m[4 * c + r] for r in range(4) for c in range(4))


class _Block2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, matrix, merge):
        if x.dim() != 4 or x.shape[0] != 1 or not x.is_cuda or x.dtype != torch.float32:
            raise RuntimeError("block2x2: float32 GPU tensor [1, C, H, W]")
        x = x.contiguous()
        if merge:
            C, h, w = int(x.shape[1]) // 4, int(x.shape[2]), int(x.shape[3])
            if x.shape[1] % 4:
                raise RuntimeError("merge expects 4*C channels")
            out = torch.empty((1, C, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
        else:
            C, h, w = int(x.shape[1]), int(x.shape[2]) // 2, int(x.shape[3]) // 2
            if x.shape[2] % 2 or x.shape[3] % 2:
                raise RuntimeError("split expects even height and width")
            out = torch.empty((1, 4 * C, h, w), dtype=torch.float32, device=x.device)
        m = (ctypes.c_float * 16)(*matrix)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ag_block2x2_transform(_p(out), _p(x), ctypes.cast(m, ctypes.c_void_p), int(merge), C, h, w,
                                                        _stream(x.device)), "ag_block2x2_transform")
        ctx.cfg = (matrix, merge)
        return out

    @staticmethod
    def backward(ctx, g):
        matrix, merge = ctx.cfg
        return _Block2x2.apply(g, _transpose4(matrix), not merge), None, None


def haar_split(x):
    """HaarTransform (dual_styleunet.py:387-403): [1, C, 2h, 2w] -> [1, 4C, h, w] = cat(ll, lh, hl, hh)."""
    return _Block2x2.apply(x, _HAAR_ANALYSIS, False)


def haar_merge(y):
    """InverseHaarTransform (dual_styleunet.py:406-425): [1, 4C, h, w] -> [1, C, 2h, 2w]."""
    return _Block2x2.apply(y, _HAAR_SYNTHESIS, True)


# ---------------------------------------------------------------------------------------------------------------------------------
# ToRGB's skip path as one kernel (round 3): InverseHaarTransform -> Upsample -> HaarTransform (dual_styleunet.py:607-633) is a linear,
# local map; its coefficients follow from the two 4 x 4 Haar matrices above and the FIR kernel by pushing unit impulses through the
# three stages on a 5 x 5 grid of sites (numpy on the host, once per FIR kernel).
# ---------------------------------------------------------------------------------------------------------------------------------
def skip_chain_taps(k_up) -> "list[float]":
    """Coefficients [s'][py][px][s][a][b] (4 x 2 x 2 x 4 x 3 x 3) with out[s'][2i+py][2j+px] = sum taps * skip[s][i+a-1][j+b-1].
    ``k_up``: the 4 x 4 Upsample kernel incl. its gain (``upfirdn2d(x, k_up, up=2, pad=(2, 1))``, dual_styleunet.py:32-50).  Stage
    semantics: merge / split = the 2 x 2 block transforms above; upfirdn2d = zero stuffing, zero padding (2 before, 1 after),
    correlation with the flipped kernel (upfirdn2d.py:167-183)."""
    import numpy as np
    k = np.asarray(k_up, np.float64).reshape(4, 4)
    A = np.asarray(_HAAR_ANALYSIS, np.float64).reshape(4, 4)
    B = np.asarray(_HAAR_SYNTHESIS, np.float64).reshape(4, 4)
    n = 5
    taps = np.zeros((4, 2, 2, 4, 3, 3))
    kf = k[::-1, ::-1]
    for s in range(4):
        skip = np.zeros((4, n, n))
        skip[s, 2, 2] = 1.0
        M = np.zeros((2 * n, 2 * n))
        for i in range(n):
            for j in range(n):
                o = B @ skip[:, i, j]
                M[2 * i, 2 * j], M[2 * i, 2 * j + 1], M[2 * i + 1, 2 * j], M[2 * i + 1, 2 * j + 1] = o
        Z = np.zeros((4 * n, 4 * n))
        Z[::2, ::2] = M
        Zp = np.pad(Z, ((2, 1), (2, 1)))
        U = np.zeros((4 * n, 4 * n))
        for ky in range(4):
            for kx in range(4):
                U += kf[ky, kx] * Zp[ky:ky + 4 * n, kx:kx + 4 * n]
        O = np.zeros((4, 2 * n, 2 * n))
        for i in range(2 * n):
            for j in range(2 * n):
                O[:, i, j] = A @ np.array([U[2 * i, 2 * j], U[2 * i, 2 * j + 1], U[2 * i + 1, 2 * j], U[2 * i + 1, 2 * j + 1]])
        inside = np.zeros_like(O, dtype=bool)
        for a in range(3):
            for b in range(3):
                I, J = 2 - (a - 1), 2 - (b - 1)          # the site that reads the impulse at offset (a - 1, b - 1)
                for py in range(2):
                    for px in range(2):
                        taps[:, py, px, s, a, b] = O[:, 2 * I + py, 2 * J + px]
                        inside[:, 2 * I + py, 2 * J + px] = True
        if np.abs(O[~inside]).max() != 0.0:
            raise RuntimeError("skip_chain_taps: the composed map reaches beyond the 3 x 3 neighbourhood (unexpected FIR kernel)")
    return [float(v) for v in taps.astype(np.float32).reshape(-1)]


def skip_chain_taps_1d(k_up) -> "list[float]":
    """The composed map as two 1-D factors: ``wy[u'][p][u][a]`` (24 floats) then ``wx[u'][p][u][b]`` (24), with
    taps[(u'y, u'x)][py][px][(uy, ux)][a][b] = wy[u'y][py][uy][a] * wx[u'x][px][ux][b] and sub-band index uy + 2 ux (the reference's order
    ll, lh, hl, hh: 'lh' is high-pass along y).  Every stage is an outer product of 1-D maps when the FIR kernel is (the reference's is:
    outer([1, 3, 3, 1]), dual_styleunet.py:21-29); a kernel for which the factorisation does not reproduce the 2-D taps is rejected."""
    import numpy as np
    W = np.asarray(skip_chain_taps(k_up), np.float64).reshape(4, 2, 2, 4, 3, 3)
    a0, b0 = np.unravel_index(np.argmax(np.abs(W[0, 0, 0, 0])), (3, 3))
    ref = W[0, 0, 0, 0, a0, b0]
    wy = np.zeros((2, 2, 2, 3))
    wx = np.zeros((2, 2, 2, 3))
    for v in range(2):
        for p in range(2):
            for u in range(2):
                wy[v, p, u, :] = W[v, p, 0, u, :, b0]                 # x indices fixed at (low, parity 0, low, b0)
                wx[v, p, u, :] = W[2 * v, 0, p, 2 * u, a0, :] / ref
    R = np.einsum("vpua,wqxb->vwpquxab", wy, wx)                       # [u'y][u'x][py][px][uy][ux][a][b]
    R = R.transpose(1, 0, 2, 3, 5, 4, 6, 7).reshape(4, 2, 2, 4, 3, 3)      # sub-band index = uy + 2 ux: ux is the slow index
    if np.abs(R - W).max() > 1e-6 * np.abs(W).max():
        raise RuntimeError("skip_chain_taps_1d: the Upsample kernel is not an outer product of 1-D kernels")
    return [float(x) for x in np.concatenate([wy.reshape(-1), wx.reshape(-1)]).astype(np.float32)]


_SKIP_TAPS = {}


def _skip_taps_host(k_up: torch.Tensor):
    """The 48 coefficients as a ctypes float array (host memory: the library passes them to its kernels by value), cached per kernel
    tensor -- one read-back of 16 floats the first time."""
    key = (k_up.device, k_up.data_ptr(), k_up._version)
    e = _SKIP_TAPS.get(key)
    if e is None or e[0] is not k_up:       # the entry holds its source tensor: a freed block handed to ANOTHER kernel tensor is not a hit
        if len(_SKIP_TAPS) > 64:
            _SKIP_TAPS.clear()
        e = _SKIP_TAPS[key] = (k_up, (ctypes.c_float * 48)(*skip_chain_taps_1d(k_up.detach().cpu().numpy())))
    return ctypes.cast(e[1], ctypes.c_void_p)


def skip_chain_forward_(out: torch.Tensor, skip: torch.Tensor, k_up: torch.Tensor, accumulate: bool = True) -> torch.Tensor:
    """out [1, 4C, 2h, 2w] (+)= HaarTransform(Upsample(InverseHaarTransform(skip [1, 4C, h, w]))) in place; returns ``out``."""
    if skip.dim() != 4 or skip.shape[0] != 1 or skip.shape[1] % 4 or not skip.is_cuda or skip.dtype != torch.float32:
        raise RuntimeError("skip_chain: float32 GPU tensor [1, 4C, h, w]")
    C, h, w = int(skip.shape[1]) // 4, int(skip.shape[2]), int(skip.shape[3])
    if tuple(out.shape) != (1, 4 * C, 2 * h, 2 * w) or not out.is_contiguous() or out.dtype != torch.float32:
        raise RuntimeError("skip_chain: out must be a contiguous float32 [1, 4C, 2h, 2w]")
    skip = skip.contiguous()
    taps = _skip_taps_host(k_up)
    with _lib.on_device(skip.device):
        _lib.check(_lib.lib().ag_skip_chain_forward(_p(out), _p(skip), taps, C, h, w, int(bool(accumulate)), _stream(skip.device)),
                   "ag_skip_chain_forward")
    return out


def skip_chain_backward(g: torch.Tensor, k_up: torch.Tensor) -> torch.Tensor:
    """Adjoint of the skip path: g [1, 4C, 2h, 2w] -> gskip [1, 4C, h, w]."""
    g = g.contiguous()
    C, h, w = int(g.shape[1]) // 4, int(g.shape[2]) // 2, int(g.shape[3]) // 2
    gskip = torch.empty((1, 4 * C, h, w), dtype=torch.float32, device=g.device)
    taps = _skip_taps_host(k_up)
    with _lib.on_device(g.device):
        _lib.check(_lib.lib().ag_skip_chain_backward(_p(gskip), _p(g), taps, C, h, w, _stream(g.device)), "ag_skip_chain_backward")
    return gskip


class _SkipChain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, skip, k_up):
        C, h, w = int(skip.shape[1]) // 4, int(skip.shape[2]), int(skip.shape[3])
        out = torch.empty((1, 4 * C, 2 * h, 2 * w), dtype=torch.float32, device=skip.device)
        ctx.k_up = k_up
        return skip_chain_forward_(out, skip, k_up, accumulate=False)

    @staticmethod
    def backward(ctx, g):
        return skip_chain_backward(g, ctx.k_up), None


def skip_chain(skip: torch.Tensor, k_up: torch.Tensor) -> torch.Tensor:
    """HaarTransform(Upsample(InverseHaarTransform(skip))) as one kernel, differentiable w.r.t. ``skip``."""
    return _SkipChain.apply(skip, k_up)
