"""ctypes front-end of the CPU rasterizer oracle (``oracle/raster_oracle.c``).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product package never does.  Every entry point mirrors one stage of the reference's
``CudaRasterizer::Rasterizer::forward/backward`` (``cuda_rasterizer/rasterizer_impl.cu:197-446``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_raster.so")
_lib = None

c_f = ctypes.c_float
c_i = ctypes.c_int
P_ = ctypes.c_void_p


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "raster_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_raster.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.ago_preprocess.restype = c_i
        _lib.ago_get_higher_msb.restype = ctypes.c_uint32
    return _lib


def set_exp_scale(s: float) -> None:
    """Conditioning probe: multiply every exp() of the blend loops by ``s`` (1.0 = bit-identical to the reference build).  See
    ``ago_set_exp_scale`` in raster_oracle.c; callers must reset it to 1.0."""
    lib().ago_set_exp_scale(c_f(s))


def _p(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(P_)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def forward(means3D, colors, opacities, scales, rotations, bg, viewmatrix, projmatrix, tanfovx, tanfovy,
            img_w, img_h, scale_modifier: float = 1.0, cov3D_precomp=None, want_fragile: bool = True,
            shs=None, sh_degree: int = 0, campos=None) -> Dict[str, np.ndarray]:
    """Full forward; returns every intermediate state the parity tests compare.  With ``colors=None`` the colours come
    from the spherical harmonics ``shs`` [P, M, 3] at ``sh_degree`` seen from ``campos`` (state keys ``rgb``, ``clamped``)."""
    L = lib()
    means3D, opacities = _f32(means3D), _f32(opacities)
    colors = None if colors is None else _f32(colors)
    bg, viewmatrix, projmatrix = _f32(bg), _f32(viewmatrix), _f32(projmatrix)
    scales = None if scales is None else _f32(scales)
    rotations = None if rotations is None else _f32(rotations)
    cov3D_precomp = None if cov3D_precomp is None else _f32(cov3D_precomp)
    P, W, H = int(means3D.shape[0]), int(img_w), int(img_h)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    st = {
        "radii": np.zeros(P, np.int32), "means2D": np.zeros((P, 2), np.float32), "depths": np.zeros(P, np.float32),
        "cov3D": np.zeros((P, 6), np.float32), "conic_opacity": np.zeros((P, 4), np.float32),
        "tiles_touched": np.zeros(P, np.uint32), "point_offsets": np.zeros(P, np.uint32),
    }
    R = 0
    if P > 0:
        R = L.ago_preprocess(c_i(P), c_i(W), c_i(H), _p(means3D), _p(scales), c_f(scale_modifier), _p(rotations),
                             _p(opacities), _p(cov3D_precomp), _p(viewmatrix), _p(projmatrix), c_f(tanfovx), c_f(tanfovy),
                             _p(st["radii"]), _p(st["means2D"]), _p(st["depths"]), _p(st["cov3D"]),
                             _p(st["conic_opacity"]), _p(st["tiles_touched"]), _p(st["point_offsets"]))
    st["num_rendered"] = R
    st["keys_unsorted"] = np.zeros(R, np.uint64)
    st["vals_unsorted"] = np.zeros(R, np.uint32)
    st["keys_sorted"] = np.zeros(R, np.uint64)
    st["point_list"] = np.zeros(R, np.uint32)
    st["ranges"] = np.zeros((T, 2), np.uint32)
    st["n_contrib"] = np.zeros((H, W), np.uint32)
    st["color"] = np.zeros((3, H, W), np.float32)
    st["depth"] = np.zeros((1, H, W), np.float32)
    st["alpha"] = np.zeros((1, H, W), np.float32)
    st["fragile"] = np.zeros((H, W), np.uint8)
    if P == 0:
        # rasterize_points.cu:68-83: the native call is skipped entirely, outputs stay all-zero
        return st
    if colors is None:
        shs, campos = _f32(shs), _f32(campos)
        st["rgb"] = np.zeros((P, 3), np.float32)
        st["clamped"] = np.zeros((P, 3), np.uint8)
        L.ago_sh_forward(c_i(P), c_i(sh_degree), c_i(shs.shape[1]), _p(means3D), _p(campos), _p(shs), _p(st["radii"]),
                         _p(st["rgb"]), _p(st["clamped"]))
        colors = st["rgb"]
    L.ago_bin(c_i(P), c_i(W), c_i(H), c_i(R), _p(st["means2D"]), _p(st["depths"]), _p(st["point_offsets"]),
              _p(st["radii"]), _p(st["keys_unsorted"]), _p(st["vals_unsorted"]), _p(st["keys_sorted"]),
              _p(st["point_list"]), _p(st["ranges"]))
    L.ago_render_forward(c_i(W), c_i(H), _p(st["ranges"]), _p(st["point_list"]), _p(st["means2D"]), _p(colors),
                         _p(st["depths"]), _p(st["conic_opacity"]), _p(bg), _p(st["color"]), _p(st["depth"]),
                         _p(st["alpha"]), _p(st["n_contrib"]), _p(st["fragile"]) if want_fragile else None)
    return st


ACCUM_SLOTS = ("m2x", "m2y", "conx", "cony", "conw", "opac", "r", "g", "b", "depth")


def backward_blend(st: Dict[str, np.ndarray], colors, bg, dL_dcolor, dL_ddepth, dL_dalpha,
                   f32_accum: bool = False) -> Dict[str, np.ndarray]:
    """Per-Gaussian accumulators of the blend backward (``backward.cu:415-601``) plus ``abs_sum`` [P,10]."""
    L = lib()
    colors, bg = _f32(colors), _f32(bg)
    dL_dcolor, dL_ddepth, dL_dalpha = _f32(dL_dcolor), _f32(dL_ddepth), _f32(dL_dalpha)
    P = int(colors.shape[0])
    H, W = st["n_contrib"].shape
    g = {
        "dL_dmeans2D": np.zeros((P, 3), np.float32), "dL_dconic": np.zeros((P, 4), np.float32),
        "dL_dopacity": np.zeros((P, 1), np.float32), "dL_dcolors": np.zeros((P, 3), np.float32),
        "dL_ddepths": np.zeros((P, 1), np.float32), "abs_sum": np.zeros((P, 10), np.float32),
    }
    if P:
        L.ago_render_backward(c_i(P), c_i(W), c_i(H), _p(st["ranges"]), _p(st["point_list"]), _p(bg), _p(st["means2D"]),
                              _p(st["conic_opacity"]), _p(colors), _p(st["depths"]), _p(st["alpha"]), _p(st["n_contrib"]),
                              _p(dL_dcolor), _p(dL_ddepth), _p(dL_dalpha), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]),
                              _p(g["dL_dopacity"]), _p(g["dL_dcolors"]), _p(g["dL_ddepths"]), _p(g["abs_sum"]),
                              c_i(1 if f32_accum else 0))
    return g


def backward_preprocess(st: Dict[str, np.ndarray], acc: Dict[str, np.ndarray], means3D, scales, rotations,
                        viewmatrix, projmatrix, tanfovx, tanfovy, scale_modifier: float = 1.0,
                        cov3D_precomp=None) -> Dict[str, np.ndarray]:
    """computeCov2DCUDA + preprocessCUDA backward (``backward.cu:144-412``) on given accumulators."""
    L = lib()
    means3D, viewmatrix, projmatrix = _f32(means3D), _f32(viewmatrix), _f32(projmatrix)
    scales = None if scales is None else _f32(scales)
    rotations = None if rotations is None else _f32(rotations)
    P = int(means3D.shape[0])
    H, W = st["n_contrib"].shape
    g = {"dL_dmeans3D": np.zeros((P, 3), np.float32), "dL_dcov3D": np.zeros((P, 6), np.float32),
         "dL_dscales": np.zeros((P, 3), np.float32), "dL_drotations": np.zeros((P, 4), np.float32)}
    if P:
        cov3D = st["cov3D"] if cov3D_precomp is None else _f32(cov3D_precomp)
        L.ago_preprocess_backward(c_i(P), c_i(W), c_i(H), _p(means3D), _p(st["radii"]), _p(scales), _p(rotations),
                                  c_f(scale_modifier), _p(cov3D), _p(viewmatrix), _p(projmatrix), c_f(tanfovx),
                                  c_f(tanfovy), _p(_f32(acc["dL_dmeans2D"])), _p(_f32(acc["dL_dconic"])),
                                  _p(_f32(acc["dL_ddepths"])), _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]),
                                  _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


def backward_sh(st: Dict[str, np.ndarray], g: Dict[str, np.ndarray], means3D, shs, sh_degree: int, campos) -> None:
    """SH part of the preprocess backward (``backward.cu:20-139``, called at ``:406-407``): writes ``g['dL_dsh']`` and adds
    the view-direction term to ``g['dL_dmeans3D']`` (after the projection and depth terms, as the reference orders them)."""
    L = lib()
    means3D, shs, campos = _f32(means3D), _f32(shs), _f32(campos)
    P, M = int(shs.shape[0]), int(shs.shape[1])
    g["dL_dsh"] = np.zeros((P, M, 3), np.float32)
    if P:
        L.ago_sh_backward(c_i(P), c_i(sh_degree), c_i(M), _p(means3D), _p(campos), _p(shs), _p(st["clamped"]),
                          _p(st["radii"]), _p(_f32(g["dL_dcolors"])), _p(g["dL_dmeans3D"]), _p(g["dL_dsh"]))


def backward(st: Dict[str, np.ndarray], means3D, colors, scales, rotations, bg, viewmatrix, projmatrix,
             tanfovx, tanfovy, dL_dcolor, dL_ddepth, dL_dalpha, scale_modifier: float = 1.0,
             cov3D_precomp=None, f32_accum: bool = False, shs=None, sh_degree: int = 0, campos=None) -> Dict[str, np.ndarray]:
    """Backward given the forward state ``st`` (what the reference keeps in geom/binning/img buffers)."""
    if colors is None:
        colors = st["rgb"]
    g = backward_blend(st, colors, bg, dL_dcolor, dL_ddepth, dL_dalpha, f32_accum)
    g.update(backward_preprocess(st, g, means3D, scales, rotations, viewmatrix, projmatrix, tanfovx, tanfovy,
                                 scale_modifier, cov3D_precomp))
    if shs is not None:
        backward_sh(st, g, means3D, shs, sh_degree, campos)
    return g


def mark_visible(means3D, viewmatrix, projmatrix) -> np.ndarray:
    L = lib()
    means3D, viewmatrix, projmatrix = _f32(means3D), _f32(viewmatrix), _f32(projmatrix)
    P = int(means3D.shape[0])
    out = np.zeros(P, np.uint8)
    if P:
        L.ago_mark_visible(c_i(P), _p(means3D), _p(viewmatrix), _p(projmatrix), _p(out))
    return out.astype(bool)


def higher_msb(n: int) -> int:
    return int(lib().ago_get_higher_msb(ctypes.c_uint32(n)))
