"""Host-side camera set-up of the render path, in explicit float32.

Mirrors what ``render3`` does before it builds the rasterizer settings
(reference ``gaussians/gaussian_renderer.py:45-52`` with ``utils/graphics_utils.py:51-85``):

* ``FoV = focal2fov(f, pixels) = 2*atan(pixels / (2 f))``, ``tanfov = tan(FoV/2)``  (python doubles)
* ``viewmatrix = extr^T``  (row-major memory of the transposed extrinsic)
* ``projmatrix = extr^T @ P^T`` with ``P`` the off-centre frustum built from the intrinsics ``K`` in float32
* ``campos = inv(extr)[:3, 3]``

The reference evaluates ``P`` with float32 torch scalars and the 4x4 product with a device GEMM whose
summation order is unspecified; here the order is fixed (k = 0..3, left to right, no FMA) so that the oracle,
the tests and the HIP path all see bit-identical matrices.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

f32 = np.float32


def focal2fov(focal: float, pixels: int) -> float:
    """``utils/graphics_utils.py:84-85``."""
    return 2.0 * math.atan(pixels / (2.0 * focal))


def projection_matrix(znear: float, zfar: float, K: np.ndarray, img_w: int, img_h: int) -> np.ndarray:
    """Off-centre perspective matrix from intrinsics, float32 (``utils/graphics_utils.py:60-80``, K branch)."""
    K = np.asarray(K, dtype=f32)
    # `python_float / tensor` is evaluated by torch as reciprocal(tensor) * float32(python_float)
    near_fx = f32(f32(1.0) / K[0, 0]) * f32(znear)
    near_fy = f32(f32(1.0) / K[1, 1]) * f32(znear)
    left = -(f32(img_w) - K[0, 2]) * near_fx
    right = K[0, 2] * near_fx
    bottom = (K[1, 2] - f32(img_h)) * near_fy
    top = K[1, 2] * near_fy
    P = np.zeros((4, 4), dtype=f32)
    z_sign = 1.0
    P[0, 0] = f32(f32(1.0) / (right - left)) * f32(2.0 * znear)
    P[1, 1] = f32(f32(1.0) / (top - bottom)) * f32(2.0 * znear)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def matmul4_f32(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """4x4 float32 product with a fixed left-to-right k order and one rounding per op."""
    a = np.asarray(a, dtype=f32)
    b = np.asarray(b, dtype=f32)
    out = np.zeros((4, 4), dtype=f32)
    for i in range(4):
        for j in range(4):
            acc = f32(a[i, 0] * b[0, j])
            for k in range(1, 4):
                acc = f32(acc + f32(a[i, k] * b[k, j]))
            out[i, j] = acc
    return out


def camera_from_intr_extr(extr: np.ndarray, intr: np.ndarray, img_w: int, img_h: int,
                          znear: float = 0.1, zfar: float = 100.0) -> Dict[str, object]:
    """Everything ``GaussianRasterizationSettings`` needs, as host values.

    Returns ``tanfovx, tanfovy`` (python floats), ``viewmatrix, projmatrix`` ([4,4] float32, memory order as the
    rasterizer reads it: element ``[i, j]`` at ``4*i + j``) and ``campos`` ([3] float32).
    """
    extr = np.asarray(extr, dtype=f32)
    intr = np.asarray(intr, dtype=f32)
    fovx = focal2fov(float(intr[0, 0]), img_w)
    fovy = focal2fov(float(intr[1, 1]), img_h)
    world_view = np.ascontiguousarray(extr.T)
    proj = np.ascontiguousarray(projection_matrix(znear, zfar, intr, img_w, img_h).T)
    full_proj = matmul4_f32(world_view, proj)
    campos = np.linalg.inv(extr.astype(np.float64))[:3, 3].astype(f32)
    return {
        "tanfovx": math.tan(fovx * 0.5),
        "tanfovy": math.tan(fovy * 0.5),
        "viewmatrix": world_view,
        "projmatrix": full_proj,
        "campos": campos,
        "img_w": int(img_w),
        "img_h": int(img_h),
    }


def rodrigues(rvec) -> np.ndarray:
    """Axis-angle to rotation matrix (what ``cv.Rodrigues`` returns for a 3-vector), float64."""
    rvec = np.asarray(rvec, dtype=np.float64)
    theta = float(np.linalg.norm(rvec))
    if theta < 1e-12:
        return np.identity(3)
    k = rvec / theta
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.identity(3) + math.sin(theta) * Kx + (1.0 - math.cos(theta)) * (Kx @ Kx)


def calc_front_mv(object_center, tar_pos=(0.0, 0.0, 2.0)) -> np.ndarray:
    """Front-view extrinsic (``utils/visualize_util.py:88-106``)."""
    m0 = np.identity(4, f32)
    m0[:3, 3] = -np.asarray(object_center, f32)
    mx = np.identity(4, f32)
    mx[:3, :3] = rodrigues([math.pi, 0.0, 0.0])
    mt = np.identity(4, f32)
    mt[:3, 3] = np.asarray(tar_pos, f32)
    return (mt @ mx @ m0).astype(f32)


def calc_free_mv(object_center, tar_pos=(0.0, 0.0, 2.0), rot_Y: float = 0.0, rot_X: float = 0.0) -> np.ndarray:
    """Free-view extrinsic without global orientation (``utils/visualize_util.py:133-162``)."""
    m0 = np.identity(4, f32)
    m0[:3, 3] = -np.asarray(object_center, f32)
    mg = np.identity(4, f32)
    mg[:3, :3] = rodrigues([math.pi, 0.0, 0.0])
    my = np.identity(4, f32)
    my[:3, :3] = rodrigues([0.0, rot_Y, 0.0])
    mx = np.identity(4, f32)
    mx[:3, :3] = rodrigues([rot_X, 0.0, 0.0])
    mt = np.identity(4, f32)
    mt[:3, 3] = np.asarray(tar_pos, f32)
    return (mt @ mx @ my @ mg @ m0).astype(f32)
