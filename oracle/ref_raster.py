"""ctypes front-end of oracle/_ref/libref_raster.so: the reference's OWN rasterizer sources running on the CPU.

TEST INFRASTRUCTURE ONLY (see oracle/ref_build.py, oracle/cuda_cpu/cuda_runtime.h).  Returns the same dictionaries as
oracle/raster_oracle.py so the two can be compared key by key.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libref_raster.so")
_lib = None
c_f, c_i, P_ = ctypes.c_float, ctypes.c_int, ctypes.c_void_p


def available() -> bool:
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ref_create.restype = P_
        _lib.ref_forward.restype = c_i
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(P_)


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


class RefRasterizer:
    """One forward (+ optional backward) through CudaRasterizer::Rasterizer of the reference."""

    def __init__(self):
        self.h = P_(lib().ref_create())

    def __del__(self):
        try:
            lib().ref_destroy(self.h)
        except Exception:
            pass

    def forward(self, means3D, colors, opacities, scales, rotations, bg, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
                img_w, img_h, scale_modifier=1.0, cov3D_precomp=None, shs=None, sh_degree=0) -> Dict[str, np.ndarray]:
        L = lib()
        self.inp = dict(means3D=_f32(means3D), colors=_f32(colors), opacities=_f32(opacities), scales=_f32(scales),
                        rotations=_f32(rotations), bg=_f32(bg), view=_f32(viewmatrix), proj=_f32(projmatrix),
                        campos=_f32(campos), cov3D_precomp=_f32(cov3D_precomp), shs=_f32(shs))
        self.sh_degree = int(sh_degree)
        i = self.inp
        P, W, H = int(i["means3D"].shape[0]), int(img_w), int(img_h)
        T = ((W + 15) // 16) * ((H + 15) // 16)
        self.tan = (float(tanfovx), float(tanfovy), float(scale_modifier))
        st = {"color": np.zeros((3, H, W), np.float32), "depth": np.zeros((1, H, W), np.float32),
              "alpha": np.zeros((1, H, W), np.float32), "radii": np.zeros(P, np.int32)}
        if i["shs"] is None:
            R = L.ref_forward(self.h, c_i(P), c_i(W), c_i(H), _p(i["bg"]), _p(i["means3D"]), _p(i["colors"]),
                              _p(i["opacities"]), _p(i["scales"]), c_f(scale_modifier), _p(i["rotations"]),
                              _p(i["cov3D_precomp"]), _p(i["view"]), _p(i["proj"]), _p(i["campos"]), c_f(tanfovx),
                              c_f(tanfovy), _p(st["color"]), _p(st["depth"]), _p(st["alpha"]), _p(st["radii"]))
        else:
            R = L.ref_forward_sh(self.h, c_i(P), c_i(self.sh_degree), c_i(i["shs"].shape[1]), c_i(W), c_i(H), _p(i["bg"]),
                                 _p(i["means3D"]), _p(i["shs"]), _p(i["opacities"]), _p(i["scales"]), c_f(scale_modifier),
                                 _p(i["rotations"]), _p(i["cov3D_precomp"]), _p(i["view"]), _p(i["proj"]), _p(i["campos"]),
                                 c_f(tanfovx), c_f(tanfovy), _p(st["color"]), _p(st["depth"]), _p(st["alpha"]), _p(st["radii"]))
            st["rgb"] = np.zeros((P, 3), np.float32)
            st["clamped"] = np.zeros((P, 3), np.uint8)
            L.ref_get_colors(self.h, _p(st["rgb"]), _p(st["clamped"]))
        st["num_rendered"] = R
        st.update(depths=np.zeros(P, np.float32), means2D=np.zeros((P, 2), np.float32), cov3D=np.zeros((P, 6), np.float32),
                  conic_opacity=np.zeros((P, 4), np.float32), tiles_touched=np.zeros(P, np.uint32),
                  point_offsets=np.zeros(P, np.uint32), keys_sorted=np.zeros(R, np.uint64),
                  point_list=np.zeros(R, np.uint32), ranges=np.zeros((T, 2), np.uint32),
                  n_contrib=np.zeros((H, W), np.uint32))
        L.ref_get_state(self.h, _p(st["depths"]), _p(st["means2D"]), _p(st["cov3D"]), _p(st["conic_opacity"]),
                        _p(st["tiles_touched"]), _p(st["point_offsets"]), _p(st["keys_sorted"]), _p(st["point_list"]),
                        _p(st["ranges"]), _p(st["n_contrib"]))
        self.st = st
        return st

    def backward(self, dL_dcolor, dL_ddepth, dL_dalpha, alphas=None) -> Dict[str, np.ndarray]:
        L = lib()
        i, st = self.inp, self.st
        P = int(i["means3D"].shape[0])
        g = {"dL_dmeans2D": np.zeros((P, 3), np.float32), "dL_dconic": np.zeros((P, 4), np.float32),
             "dL_dopacity": np.zeros((P, 1), np.float32), "dL_dcolors": np.zeros((P, 3), np.float32),
             "dL_ddepths": np.zeros((P, 1), np.float32), "dL_dmeans3D": np.zeros((P, 3), np.float32),
             "dL_dcov3D": np.zeros((P, 6), np.float32), "dL_dscales": np.zeros((P, 3), np.float32),
             "dL_drotations": np.zeros((P, 4), np.float32)}
        al = _f32(st["alpha"] if alphas is None else alphas)
        a, b, c = _f32(dL_dcolor), _f32(dL_ddepth), _f32(dL_dalpha)
        if i["shs"] is None:
            L.ref_backward(self.h, _p(i["bg"]), _p(i["means3D"]), _p(i["colors"]), _p(al), _p(i["scales"]), c_f(self.tan[2]),
                           _p(i["rotations"]), _p(i["cov3D_precomp"]), _p(i["view"]), _p(i["proj"]), _p(i["campos"]),
                           c_f(self.tan[0]), c_f(self.tan[1]), _p(st["radii"]), _p(a), _p(b), _p(c), _p(g["dL_dmeans2D"]),
                           _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolors"]), _p(g["dL_ddepths"]),
                           _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
        else:
            M = int(i["shs"].shape[1])
            g["dL_dsh"] = np.zeros((P, M, 3), np.float32)
            L.ref_backward_sh(self.h, c_i(self.sh_degree), c_i(M), _p(i["bg"]), _p(i["means3D"]), _p(i["shs"]), _p(al),
                              _p(i["scales"]), c_f(self.tan[2]), _p(i["rotations"]), _p(i["cov3D_precomp"]), _p(i["view"]),
                              _p(i["proj"]), _p(i["campos"]), c_f(self.tan[0]), c_f(self.tan[1]), _p(st["radii"]), _p(a),
                              _p(b), _p(c), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]),
                              _p(g["dL_dcolors"]), _p(g["dL_ddepths"]), _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]),
                              _p(g["dL_dsh"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
        return g


def mark_visible(means3D, viewmatrix, projmatrix) -> np.ndarray:
    m, v, p = _f32(means3D).copy(), _f32(viewmatrix).copy(), _f32(projmatrix).copy()
    out = np.zeros(m.shape[0], np.bool_)
    lib().ref_mark_visible(c_i(m.shape[0]), _p(m), _p(v), _p(p), _p(out))
    return out
