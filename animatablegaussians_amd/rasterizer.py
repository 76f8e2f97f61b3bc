"""Python operator surface of the depth+alpha Gaussian rasterizer, backed by ``libag_hip.so``.

Same names, argument order, return tuples and error behaviour as the reference package
``diff_gaussian_rasterization_depth_alpha`` (``.../diff_gaussian_rasterization_depth_alpha/__init__.py:21-223``) and its
pybind module ``_C`` (``ext.cpp:15-19``, ``rasterize_points.cu:35-229``), so ``gaussians/gaussian_renderer.py`` can
import it unchanged:

* ``GaussianRasterizationSettings`` (NamedTuple, ``__init__.py:160-172``)
* ``GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs, colors_precomp, scales, rotations,
  cov3D_precomp) -> (color[3,H,W], radii[P] int32, depth[1,H,W], alpha[1,H,W])`` and ``.markVisible``
* ``_C.rasterize_gaussians`` / ``_C.rasterize_gaussians_backward`` / ``_C.mark_visible`` equivalents:
  :func:`native_rasterize_gaussians`, :func:`native_rasterize_gaussians_backward`, :func:`native_mark_visible`.

Host code is plumbing only (allocation, pointer passing, autograd bookkeeping); all arithmetic happens in the HIP
kernels.  There is no CPU or eager-PyTorch fallback: non-GPU tensors raise.
"""
from __future__ import annotations

import ctypes
from typing import NamedTuple, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib

NUM_CHANNELS = 3  # cuda_rasterizer/config.h:15


def _ptr(t: Optional[torch.Tensor]):
    """Device pointer of a tensor, or NULL for None / empty (the reference's 'not provided' torch.Tensor([]))."""
    if t is None or t.numel() == 0:
        return None
    return ctypes.c_void_p(t.data_ptr())


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype is torch.float32 and t.is_cuda and t.is_contiguous():      # the usual case, checked first: this runs ~25x per view
        return t
    if t.numel() == 0:
        return t
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (there is no CPU rasterizer in this package)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def _stream_ptr(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _on_device:
    """``torch.cuda.device(dev)`` only when ``dev`` is not already current (the context manager costs ~5 us per use)."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        self.ctx = None if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


_bytes_cache = {}


def _scratch_bytes(kind: str, *dims) -> int:
    """Scratch sizes from the library, memoised (pure functions of their arguments)."""
    k = (kind,) + dims
    v = _bytes_cache.get(k)
    if v is None:
        v = _bytes_cache[k] = int(getattr(_lib.lib(), "ag_raster_" + kind + "_bytes")(*dims))
    return v


# Instance capacity the next optimistic forward sizes its binning buffer for, per (P, W, H, device): 1.25 x the largest
# count seen so far.  The first frame of a configuration (and any frame that outgrows the capacity) takes the two-stage path.
_capacity = {}


def native_rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                               viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                               campos, prefiltered, debug, *, _optimistic=False):
    """``_C.rasterize_gaussians`` (``rasterize_points.cu:35-119``).

    Returns ``(num_rendered, color, depth, alpha, radii, geomBuffer, binningBuffer, imgBuffer)``.

    ``_optimistic`` (the autograd node's path): all forward stages are enqueued before the host reads the instance count,
    against a binning buffer sized from earlier frames (``ag_raster_forward_optimistic``), so the GPU does not idle over the
    host round trip the reference has between its scan and its sort (``rasterizer_impl.cu:282``).  The first element of the
    result is then a pair ``(num_rendered, layout_R)``: the scratch is laid out for ``layout_R`` instances and THAT is what the
    backward takes as ``R``.
    """
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    L = _lib.lib()
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    dev = means3D.device
    f_opts = dict(dtype=torch.float32, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    byte = dict(dtype=torch.uint8, device=dev)
    if P == 0:
        # the reference skips the native call and returns its zero-filled outputs and empty buffers
        z = lambda c: torch.zeros((c, H, W), **f_opts)  # noqa: E731
        e = lambda: torch.empty((0,), **byte)           # noqa: E731
        return ((0, 0) if _optimistic else 0), z(NUM_CHANNELS), z(1), z(1), radii, e(), e(), e()

    out_color = torch.empty((NUM_CHANNELS, H, W), **f_opts)
    out_depth = torch.empty((1, H, W), **f_opts)
    out_alpha = torch.empty((1, H, W), **f_opts)
    means3D = _f32c(means3D, "means3D")
    colors = _f32c(colors, "colors_precomp")
    opacity = _f32c(opacity, "opacities")
    scales = _f32c(scales, "scales")
    rotations = _f32c(rotations, "rotations")
    cov3D_precomp = _f32c(cov3D_precomp, "cov3D_precomp")
    sh = _f32c(sh, "shs")
    background = _f32c(background, "bg")
    viewmatrix = _f32c(viewmatrix, "viewmatrix")
    projmatrix = _f32c(projmatrix, "projmatrix")
    campos = _f32c(campos, "campos")

    geom = torch.empty((_scratch_bytes("geom", P),), **byte)
    img = torch.empty((_scratch_bytes("image", W, H),), **byte)

    a = _lib.AgRasterForwardArgs()
    a.P, a.W, a.H = P, W, H
    a.sh_degree = int(degree)
    a.sh_coeffs = int(sh.size(1)) if sh.numel() else 0
    a.prefiltered = int(bool(prefiltered))
    a.tan_fovx, a.tan_fovy, a.scale_modifier = float(tan_fovx), float(tan_fovy), float(scale_modifier)
    a.bg = _ptr(background); a.means3D = _ptr(means3D); a.colors_precomp = _ptr(colors); a.shs = _ptr(sh)
    a.opacities = _ptr(opacity); a.scales = _ptr(scales); a.rotations = _ptr(rotations)
    a.cov3D_precomp = _ptr(cov3D_precomp)
    a.viewmatrix = _ptr(viewmatrix); a.projmatrix = _ptr(projmatrix); a.campos = _ptr(campos)
    a.out_color = _ptr(out_color); a.out_depth = _ptr(out_depth); a.out_alpha = _ptr(out_alpha); a.radii = _ptr(radii)
    a.geom_buffer = _ptr(geom); a.geom_bytes = geom.numel()
    a.image_buffer = _ptr(img); a.image_bytes = img.numel()
    a.binning_buffer = None; a.binning_bytes = 0

    with _on_device(dev):
        stream = _stream_ptr(dev)
        R = ctypes.c_int32(0)
        key = (P, W, H, dev.index)
        cap = _capacity.get(key) if _optimistic else None
        layout_R = None
        if cap is not None:
            binning = torch.empty((_scratch_bytes("binning", cap),), **byte)
            a.binning_buffer = _ptr(binning); a.binning_bytes = binning.numel()
            rc = L.ag_raster_forward_optimistic(ctypes.byref(a), cap, stream, ctypes.byref(R))
            if rc == 0:
                layout_R = cap
            elif rc != _lib.AG_ERR_SCRATCH_TOO_SMALL:
                _lib.check(rc, "ag_raster_forward_optimistic")
        if layout_R is None:                                  # first frame of this configuration, or more instances than planned for
            _lib.check(L.ag_raster_forward_plan(ctypes.byref(a), stream, ctypes.byref(R)), "ag_raster_forward_plan")
            layout_R = int(R.value)
            binning = torch.empty((L.ag_raster_binning_bytes(layout_R),), **byte)
            a.binning_buffer = _ptr(binning); a.binning_bytes = binning.numel()
            _lib.check(L.ag_raster_forward_render(ctypes.byref(a), layout_R, stream), "ag_raster_forward_render")
        num_rendered = int(R.value)
        if _optimistic:
            want = num_rendered + num_rendered // 4 + 1024
            if want > _capacity.get(key, 0):
                _capacity[key] = want
        if debug:
            torch.cuda.synchronize(dev)
    if _optimistic:
        return (num_rendered, layout_R), out_color, out_depth, out_alpha, radii, geom, binning, img
    return num_rendered, out_color, out_depth, out_alpha, radii, geom, binning, img


def native_rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                        cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                        dL_dout_depth, dL_dout_alpha, sh, degree, campos, geomBuffer, R,
                                        binningBuffer, imageBuffer, alphas, debug, *, _accum_buffer=None):
    """``_C.rasterize_gaussians_backward`` (``rasterize_points.cu:121-208``).

    Returns ``(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)``.
    ``_accum_buffer`` (tests only) lets the caller own the per-Gaussian accumulator scratch so the internal
    ``dL_dconic`` / ``dL_ddepths`` sums can be inspected.
    """
    L = _lib.lib()
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if sh.numel() else 0
    dev = means3D.device
    f = dict(dtype=torch.float32, device=dev)
    dL_dmeans3D = torch.empty((P, 3), **f)
    dL_dmeans2D = torch.empty((P, 3), **f)
    dL_dcolors = torch.empty((P, NUM_CHANNELS), **f)
    dL_dopacity = torch.empty((P, 1), **f)
    dL_dcov3D = torch.empty((P, 6), **f)
    dL_dsh = torch.zeros((P, M, 3), **f)
    dL_dscales = torch.empty((P, 3), **f)
    dL_drotations = torch.empty((P, 4), **f)
    if P == 0:
        return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations

    accum = _accum_buffer if _accum_buffer is not None else torch.empty((_scratch_bytes("accum", P),), dtype=torch.uint8, device=dev)
    keep = [_f32c(t, n) for t, n in ((background, "bg"), (means3D, "means3D"), (colors, "colors_precomp"),
                                     (scales, "scales"), (rotations, "rotations"), (cov3D_precomp, "cov3D_precomp"),
                                     (viewmatrix, "viewmatrix"), (projmatrix, "projmatrix"), (campos, "campos"),
                                     (alphas, "alphas"), (dL_dout_color, "dL_dout_color"),
                                     (dL_dout_depth, "dL_dout_depth"), (dL_dout_alpha, "dL_dout_alpha"), (sh, "shs"))]
    (background, means3D, colors, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, alphas,
     dL_dout_color, dL_dout_depth, dL_dout_alpha, sh) = keep
    radii = radii.contiguous()

    b = _lib.AgRasterBackwardArgs()
    b.P, b.W, b.H = P, W, H
    b.sh_degree, b.sh_coeffs, b.num_rendered = int(degree), M, int(R)
    b.tan_fovx, b.tan_fovy, b.scale_modifier = float(tan_fovx), float(tan_fovy), float(scale_modifier)
    b.bg = _ptr(background); b.means3D = _ptr(means3D); b.radii = _ptr(radii); b.colors_precomp = _ptr(colors)
    b.shs = _ptr(sh); b.scales = _ptr(scales); b.rotations = _ptr(rotations); b.cov3D_precomp = _ptr(cov3D_precomp)
    b.viewmatrix = _ptr(viewmatrix); b.projmatrix = _ptr(projmatrix); b.campos = _ptr(campos); b.alphas = _ptr(alphas)
    b.dL_dout_color = _ptr(dL_dout_color); b.dL_dout_depth = _ptr(dL_dout_depth); b.dL_dout_alpha = _ptr(dL_dout_alpha)
    b.geom_buffer = _ptr(geomBuffer); b.image_buffer = _ptr(imageBuffer); b.binning_buffer = _ptr(binningBuffer)
    b.dL_dmeans2D = _ptr(dL_dmeans2D); b.dL_dcolors = _ptr(dL_dcolors); b.dL_dopacity = _ptr(dL_dopacity)
    b.dL_dmeans3D = _ptr(dL_dmeans3D); b.dL_dcov3D = _ptr(dL_dcov3D); b.dL_dsh = _ptr(dL_dsh)
    b.dL_dscales = _ptr(dL_dscales); b.dL_drotations = _ptr(dL_drotations)
    b.accum_buffer = _ptr(accum); b.accum_bytes = accum.numel()
    with _on_device(dev):
        _lib.check(L.ag_raster_backward(ctypes.byref(b), _stream_ptr(dev)), "ag_raster_backward")
        if debug:
            torch.cuda.synchronize(dev)
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def native_mark_visible(means3D, viewmatrix, projmatrix):
    """``_C.mark_visible`` (``rasterize_points.cu:210-229``)."""
    L = _lib.lib()
    P = int(means3D.size(0))
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P:
        means3D = _f32c(means3D, "means3D")
        viewmatrix = _f32c(viewmatrix, "viewmatrix")
        projmatrix = _f32c(projmatrix, "projmatrix")
        with _lib.on_device(means3D.device):
            _lib.check(L.ag_raster_mark_visible(P, _ptr(means3D), _ptr(viewmatrix), _ptr(projmatrix), _ptr(present),
                                                _stream_ptr(means3D.device)), "ag_raster_mark_visible")
    return present


class _RasterizeGaussians(torch.autograd.Function):
    """Autograd node with the reference's input order (``__init__.py:44-158``)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        rs = raster_settings
        (num_rendered, layout_R), color, depth, alpha, radii, geom, binning, img = native_rasterize_gaussians(
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree,
            rs.campos, rs.prefiltered, rs.debug, _optimistic=True)
        ctx.raster_settings = rs
        ctx.num_rendered = layout_R          # what the scratch buffers are laid out for (>= the true count)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning,
                              img, alpha)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img, alpha = ctx.saved_tensors
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = native_rasterize_gaussians_backward(
            rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
            rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_color, grad_depth, grad_alpha, sh, rs.sh_degree, rs.campos,
            geom, ctx.num_rendered, binning, img, alpha, rs.debug)
        # gradients of inputs that were "not provided" (empty tensors) are dropped, as autograd would ignore them
        return (grad_means3D, grad_means2D,
                grad_sh if sh.numel() else None,
                grad_colors_precomp if colors_precomp.numel() else None,
                grad_opacities,
                grad_scales if scales.numel() else None,
                grad_rotations if rotations.numel() else None,
                grad_cov3Ds_precomp if cov3Ds_precomp.numel() else None,
                None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return native_mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        # same argument contract and messages as the reference (__init__.py:194-198)
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        has_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (has_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)


class _PreparedView:
    """Argument structures + owned outputs of one (camera, slot) pair: see FusedRasterStep.prepare."""
    __slots__ = ("a", "b", "slot", "color", "depth", "alpha", "radii", "keep", "inputs", "in_ptrs", "checked", "image_grads")


class FusedRasterStep:
    """Forward AND backward of camera views in one native call each (``ag_raster_forward_backward``), for callers that hold the upstream
    image gradients when they render: the inner loop of a multi-view trainer, the throughput benchmark.  No autograd node, no Python
    between the two halves, and the library owns what ``bench.py`` used to do by hand in round 1:

    * consecutive views go to ``n_streams`` internal HIP streams in turn (views are independent; the one host wait per view -- the
      instance count -- then overlaps the other stream's kernels);
    * scratch (geometry / image / binning / accumulator buffers) is allocated once per stream and reused;
    * ``accumulate=True`` sums the per-Gaussian gradients over the views of a step on the device (per stream, ``join()`` adds the
      streams): the quantity view-sharded training exchanges.

    Colours-precomp path only (what the avatar uses, ``gaussians/gaussian_renderer.py:84``).  Results equal
    ``GaussianRasterizer`` + autograd on the same inputs (tests/test_raster_gpu.py)."""

    _GRADS = (("dL_dmeans2D", 3), ("dL_dcolors", 3), ("dL_dopacity", 1), ("dL_dmeans3D", 3), ("dL_dcov3D", 6), ("dL_dscales", 3),
              ("dL_drotations", 4))

    def __init__(self, P: int, W: int, H: int, device, n_streams: int = 2):
        self.P, self.W, self.H = int(P), int(W), int(H)
        self.dev = torch.device(device)
        if self.dev.index is None:
            self.dev = torch.device("cuda", torch.cuda.current_device())
        self.key = (self.P, self.W, self.H, self.dev.index)
        byte = dict(dtype=torch.uint8, device=self.dev)
        self.slots = []
        for _ in range(max(1, n_streams)):
            grads = {name: torch.zeros((self.P, c), dtype=torch.float32, device=self.dev) for name, c in self._GRADS}
            self.slots.append(dict(stream=torch.cuda.Stream(self.dev), grads=grads, used=False,
                                   geom=torch.empty((_scratch_bytes("geom", self.P),), **byte),
                                   img=torch.empty((_scratch_bytes("image", self.W, self.H),), **byte),
                                   accum=torch.empty((_scratch_bytes("accum", self.P),), **byte), binning=None, cap=0, pending=None))
        self._next = 0

    def _binning(self, slot, cap):
        if slot["binning"] is None or slot["cap"] < cap:
            old = slot["binning"]
            if old is not None:
                # allocated on the caller's stream, used only on the slot's: the previous view's backward may still be reading it
                old.record_stream(slot["stream"])
            slot["binning"] = torch.empty((_scratch_bytes("binning", cap),), dtype=torch.uint8, device=self.dev)
            slot["cap"] = cap
        return slot["binning"]

    def prepare(self, rs: GaussianRasterizationSettings, g_color, g_depth, g_alpha, slot: int) -> "_PreparedView":
        """Everything of a view that does not change from one iteration to the next, built once: the camera, the output images, the slot's
        scratch and gradient arrays, the upstream image gradients, all as filled-in argument structures.  A trainer's cameras are fixed, so
        it prepares one handle per (camera, slot) and pays per iteration only for :meth:`run` (five pointer updates and the native call:
        ~15 us of Python against ~60 for :meth:`view`).  The handle owns its output images: a later ``run`` of the same handle overwrites them."""
        W, H, dev, P = self.W, self.H, self.dev, self.P
        if int(rs.image_width) != W or int(rs.image_height) != H:
            raise RuntimeError("FusedRasterStep: sizes differ from the ones it was built for")
        sl = self.slots[int(slot)]
        f32 = dict(dtype=torch.float32, device=dev)
        h = _PreparedView()
        h.slot = int(slot)
        h.color, h.depth, h.alpha = torch.empty((NUM_CHANNELS, H, W), **f32), torch.empty((1, H, W), **f32), torch.empty((1, H, W), **f32)
        h.radii = torch.empty((P,), dtype=torch.int32, device=dev)
        cam = [_f32c(t, n) for t, n in ((rs.bg, "bg"), (rs.viewmatrix, "viewmatrix"), (rs.projmatrix, "projmatrix"), (rs.campos, "campos"))]
        h.keep = cam                                        # the structures hold raw pointers: keep their owners alive
        a = h.a = _lib.AgRasterForwardArgs()
        a.P, a.W, a.H = P, W, H
        a.sh_degree = a.sh_coeffs = 0
        a.prefiltered = int(bool(rs.prefiltered))
        a.tan_fovx, a.tan_fovy, a.scale_modifier = float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier)
        a.bg, a.viewmatrix, a.projmatrix, a.campos = (_ptr(t) for t in cam)
        a.out_color = _ptr(h.color); a.out_depth = _ptr(h.depth); a.out_alpha = _ptr(h.alpha); a.radii = _ptr(h.radii)
        a.geom_buffer = _ptr(sl["geom"]); a.geom_bytes = sl["geom"].numel()
        a.image_buffer = _ptr(sl["img"]); a.image_bytes = sl["img"].numel()
        b = h.b = _lib.AgRasterBackwardArgs()
        g = sl["grads"]
        b.dL_dmeans2D = _ptr(g["dL_dmeans2D"]); b.dL_dcolors = _ptr(g["dL_dcolors"]); b.dL_dopacity = _ptr(g["dL_dopacity"])
        b.dL_dmeans3D = _ptr(g["dL_dmeans3D"]); b.dL_dcov3D = _ptr(g["dL_dcov3D"]); b.dL_dscales = _ptr(g["dL_dscales"])
        b.dL_drotations = _ptr(g["dL_drotations"])
        b.accum_buffer = _ptr(sl["accum"]); b.accum_bytes = sl["accum"].numel()
        h.inputs = h.in_ptrs = None
        self._set_image_grads(h, g_color, g_depth, g_alpha)
        return h

    @staticmethod
    def _set_image_grads(h, g_color, g_depth, g_alpha):
        gs = [_f32c(t, n) for t, n in ((g_color, "dL_dout_color"), (g_depth, "dL_dout_depth"), (g_alpha, "dL_dout_alpha"))]
        h.image_grads = gs
        h.b.dL_dout_color, h.b.dL_dout_depth, h.b.dL_dout_alpha = (_ptr(t) for t in gs)

    def run(self, h: "_PreparedView", means3D, colors, opacities, scales, rotations, image_grads=None, accumulate: bool = False,
            inputs_outlive_join: bool = False):
        """Enqueue forward + backward of a prepared view on its slot's stream.  ``image_grads``: new ``(g_color, g_depth, g_alpha)`` of this
        iteration (None = the ones the handle holds).  ``inputs_outlive_join``: the caller keeps the five input arrays and the image
        gradients alive until :meth:`join` (then their ``record_stream`` bookkeeping, ~1 us each, is skipped).
        Returns ``(color, depth, alpha, radii, grads)`` like :meth:`view`.

        The host does not wait for the view's instance count here (``ag_raster_forward_backward_enqueue``): it is read when the slot is
        used next or at :meth:`join` (``ag_raster_collect``).  Should the view turn out to have more instances than the slot's binning
        buffer holds, it is redone at that point -- until then its images are not valid; the gradient sums are never touched by a view
        that does not fit.  So: consume the returned images only after :meth:`join` (or after the next ``run`` on the same slot).
        While the instance count of this configuration has not been learned yet (the first views of a (P, W, H) on this device) the view
        is collected before returning, so a first frame is never handed out unfinished."""
        L = _lib.lib()
        P, dev = self.P, self.dev
        sl = self.slots[h.slot]
        self._collect(sl)                                         # the slot's previous view: its count is known by now (or soon)
        ins = (means3D, colors, opacities, scales, rotations)
        if h.inputs is None or any(x is not y for x, y in zip(ins, h.inputs)) or any(t.data_ptr() != q for t, q in zip(ins, h.in_ptrs)):
            if int(means3D.size(0)) != P:
                raise RuntimeError("FusedRasterStep: sizes differ from the ones it was built for")
            chk = [_f32c(t, n) for t, n in zip(ins, ("means3D", "colors_precomp", "opacities", "scales", "rotations"))]
            a = h.a
            a.means3D, a.colors_precomp, a.opacities, a.scales, a.rotations = (_ptr(t) for t in chk)
            h.inputs, h.in_ptrs, h.checked = ins, [t.data_ptr() for t in ins], chk
        if image_grads is not None:
            self._set_image_grads(h, *image_grads)
        a, b = h.a, h.b
        b.accumulate = int(bool(accumulate and sl["used"]))       # the slot's first view of a step writes, later ones add
        st = sl["stream"]
        st.wait_stream(torch.cuda.current_stream(dev))            # inputs were produced on the caller's stream
        with _on_device(dev):
            cap = _capacity.get(self.key) or (4 * P + 4096)
            binning = self._binning(sl, cap)
            a.binning_buffer = binning.data_ptr(); a.binning_bytes = binning.numel()
            tk = ctypes.c_int32(-1)
            _lib.check(L.ag_raster_forward_backward_enqueue(ctypes.byref(a), ctypes.byref(b), cap, ctypes.c_void_p(st.cuda_stream), ctypes.byref(tk)),
                       "ag_raster_forward_backward_enqueue")
        if tk.value >= 0:
            sl["pending"] = (tk.value, h, cap)
            if not _capacity.get(self.key):
                self._collect(sl)                                 # capacity not learned yet: synchronous until it is
        sl["used"] = True
        if not inputs_outlive_join:
            for t in h.checked + h.image_grads:
                t.record_stream(st)                               # allocator: still in use on the internal stream
        return h.color, h.depth, h.alpha, h.radii, sl["grads"]

    def _collect(self, sl):
        """Read the instance count of the slot's pending view (waits for its preprocess + scan only).  More instances than the binning
        buffer was sized for: the view did not touch its outputs or the sums; it is redone here, synchronously, with a larger buffer."""
        pend = sl.get("pending")
        if pend is None:
            return
        sl["pending"] = None
        tk, h, cap = pend
        L = _lib.lib()
        R = ctypes.c_int32(0)
        with _on_device(self.dev):
            rc = L.ag_raster_collect(tk, ctypes.byref(R))
            while rc == _lib.AG_ERR_SCRATCH_TOO_SMALL:
                cap = int(R.value) + int(R.value) // 4 + 1024
                binning = self._binning(sl, cap)
                h.a.binning_buffer = binning.data_ptr(); h.a.binning_bytes = binning.numel()
                rc = L.ag_raster_forward_backward(ctypes.byref(h.a), ctypes.byref(h.b), cap, ctypes.c_void_p(sl["stream"].cuda_stream), ctypes.byref(R))
            _lib.check(rc, "ag_raster_forward_backward")
        want = int(R.value) + int(R.value) // 4 + 1024
        if want > _capacity.get(self.key, 0):
            _capacity[self.key] = want

    def view(self, rs: GaussianRasterizationSettings, means3D, colors, opacities, scales, rotations, g_color, g_depth, g_alpha,
             accumulate: bool = False, slot: Optional[int] = None):
        """Enqueue forward + backward of one view.  Returns ``(color, depth, alpha, radii, grads)``; ``grads`` is the slot's dict of
        gradient arrays (valid on the slot's stream; ``join()`` before reading them from another stream).  One-off form of
        :meth:`prepare` + :meth:`run`: fresh output images per call."""
        P, W, H, dev = self.P, self.W, self.H, self.dev
        if int(means3D.size(0)) != P or int(rs.image_width) != W or int(rs.image_height) != H:
            raise RuntimeError("FusedRasterStep: sizes differ from the ones it was built for")
        k = self._next % len(self.slots) if slot is None else int(slot)
        self._next += 1
        if P == 0:
            f32 = dict(dtype=torch.float32, device=dev)
            return (torch.zeros((NUM_CHANNELS, H, W), **f32), torch.zeros((1, H, W), **f32), torch.zeros((1, H, W), **f32),
                    torch.empty((0,), dtype=torch.int32, device=dev), self.slots[k]["grads"])
        h = self.prepare(rs, g_color, g_depth, g_alpha, k)
        out = self.run(h, means3D, colors, opacities, scales, rotations, accumulate=accumulate)
        self._collect(self.slots[k])                              # one-off form: the count (and a redo, if needed) before returning
        st = self.slots[k]["stream"]
        for t in h.keep + [h.color, h.depth, h.alpha, h.radii]:
            t.record_stream(st)
        return out

    def close(self):
        """Collect every view still pending on a slot: releases their entries of the library's ticket table (64 per process -- an object
        dropped with pending views, e.g. by an exception in the middle of a step, would otherwise leak them until every enqueue fails).
        Called by ``__del__``; safe to call more than once."""
        for sl in getattr(self, "slots", []):
            try:
                self._collect(sl)
            except Exception:                                    # interpreter shutdown / a library that is already gone
                sl["pending"] = None

    def __del__(self):
        self.close()

    def join(self):
        """Order the caller's stream after every internal stream and return the gradients summed over the slots that were used since
        the last ``join()`` (``{name: [P, c] tensor}``; the tensors belong to slot 0 and are overwritten by its next view)."""
        cur = torch.cuda.current_stream(self.dev)
        for sl in self.slots:
            self._collect(sl)
            cur.wait_stream(sl["stream"])
        used = [sl for sl in self.slots if sl["used"]]
        for sl in self.slots:
            sl["used"] = False
        if not used:
            return None
        base = used[0]["grads"]
        for sl in used[1:]:
            for name, _ in self._GRADS:
                base[name].add_(sl["grads"][name])
        if used[0] is not self.slots[0]:
            for name, _ in self._GRADS:
                self.slots[0]["grads"][name].copy_(base[name])
        return self.slots[0]["grads"]
