"""``render3`` re-host: camera set-up + rasterizer call with the reference's signature and return dictionary
(reference ``gaussians/gaussian_renderer.py:19-106``).

Differences in mechanism, not in results:
  * the camera matrices are computed on the host in explicit float32 (``camera.py``) from ONE device->host copy of
    ``extr``/``intr`` and cached per (extr, intr, size); the reference does two ``.item()`` syncs plus device math;
  * the four small settings tensors are uploaded once per cached camera and reused.
"""
from __future__ import annotations

import os
import weakref
from typing import Dict

import numpy as np
import torch

from . import camera as cam
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

_SH_C0 = 0.28209479177387814
_SH_C1 = 0.4886025119029199
_SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435)

_camera_cache: Dict[bytes, dict] = {}
# The same camera TENSORS again (the cameras of a capture rig kept on the device, a multi-view trainer's fixed views): no device->host copy at
# all.  The copy below is a host wait for everything queued before it -- in a training step that is the whole StyleUNet forward, after which
# the host has lost its lead over the GPU and the small kernels that follow (loss, the first backward nodes) are issued into an idle device
# (profiles/r05c_step_gaps.txt).  Keyed on the two tensors' identity (weak references: a freed tensor's address can be handed out again) and
# version counters (an in-place change of the values misses).  AG_CAMERA_IDENT_CACHE=0: always the copy (the A/B).
_camera_ident: Dict[tuple, tuple] = {}
_IDENT_ON = os.environ.get("AG_CAMERA_IDENT_CACHE", "1") != "0"


def _camera_tensors(extr: torch.Tensor, intr: torch.Tensor, img_w: int, img_h: int, device) -> dict:
    ident = None
    # Only version-tracked in-place writes are seen by the identity key (a write through .data, a numpy alias or an external kernel is not: such
    # callers set AG_CAMERA_IDENT_CACHE=0 or pass fresh tensors); tensors without a version counter (torch.inference_mode) take the by-value path.
    if _IDENT_ON and isinstance(extr, torch.Tensor) and isinstance(intr, torch.Tensor) and not (extr.is_inference() or intr.is_inference()):
        ident = (id(extr), extr.data_ptr(), extr._version, id(intr), intr.data_ptr(), intr._version, img_w, img_h, str(device))
        got = _camera_ident.get(ident)
        if got is not None and got[0]() is extr and got[1]() is intr:
            return got[2]
    hit = _camera_tensors_by_value(extr, intr, img_w, img_h, device)
    if ident is not None:
        if len(_camera_ident) > 256:
            _camera_ident.clear()
        _camera_ident[ident] = (weakref.ref(extr), weakref.ref(intr), hit)
    return hit


def _camera_tensors_by_value(extr: torch.Tensor, intr: torch.Tensor, img_w: int, img_h: int, device) -> dict:
    host = torch.cat([extr.reshape(-1).float(), intr.reshape(-1).float()]).cpu().numpy()   # the one host sync
    key = host.tobytes() + np.array([img_w, img_h], np.int32).tobytes() + str(device).encode()
    hit = _camera_cache.get(key)
    if hit is None:
        c = cam.camera_from_intr_extr(host[:16].reshape(4, 4), host[16:25].reshape(3, 3), img_w, img_h)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
        hit = dict(tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], viewmatrix=t(c["viewmatrix"]),
                   projmatrix=t(c["projmatrix"]), campos=t(c["campos"]))
        if len(_camera_cache) > 256:
            _camera_cache.clear()
        _camera_cache[key] = hit
    return hit


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """Real spherical harmonics up to degree 3 (``utils/sh_utils.py:57``); sh [..., C, (deg+1)^2], dirs [..., 3]."""
    assert 0 <= deg <= 3 and sh.shape[-1] >= (deg + 1) ** 2
    res = _SH_C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        res = res - _SH_C1 * y * sh[..., 1] + _SH_C1 * z * sh[..., 2] - _SH_C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + _SH_C2[0] * xy * sh[..., 4] + _SH_C2[1] * yz * sh[..., 5] + _SH_C2[2] * (2.0 * zz - xx - yy) * sh[..., 6]
                   + _SH_C2[3] * xz * sh[..., 7] + _SH_C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                res = (res + _SH_C3[0] * y * (3 * xx - yy) * sh[..., 9] + _SH_C3[1] * xy * z * sh[..., 10]
                       + _SH_C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + _SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                       + _SH_C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + _SH_C3[5] * z * (xx - yy) * sh[..., 14]
                       + _SH_C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return res


def render3(gaussian_vals: dict, bg_color: torch.Tensor, extr: torch.Tensor, intr: torch.Tensor, img_w: int, img_h: int,
            scaling_modifier: float = 1.0) -> dict:
    means3D = gaussian_vals['positions']
    dev = means3D.device
    # zero tensor whose gradient is the screen-space mean gradient (gaussian_renderer.py:30-35)
    screenspace_points = torch.zeros_like(means3D, requires_grad=True) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    c = _camera_tensors(extr, intr, int(img_w), int(img_h), dev)
    settings = GaussianRasterizationSettings(
        image_height=int(img_h), image_width=int(img_w), tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=c["viewmatrix"], projmatrix=c["projmatrix"],
        sh_degree=gaussian_vals['max_sh_degree'], campos=c["campos"], prefiltered=False, debug=False)
    assert not ('colors' in gaussian_vals and 'shs' in gaussian_vals), "Cannot use both color and SH!"
    colors_precomp = gaussian_vals.get('colors')
    if 'shs' in gaussian_vals:   # SH -> RGB in Python, as the reference does (gaussian_renderer.py:76-82)
        dir_pp = means3D - c["campos"][None]
        dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        colors_precomp = torch.clamp_min(eval_sh(gaussian_vals['max_sh_degree'], gaussian_vals['shs'], dir_pp) + 0.5, 0.0)
    rendered_image, radii, rendered_depth, rendered_alpha = GaussianRasterizer(settings)(
        means3D=means3D, means2D=screenspace_points, shs=None, colors_precomp=colors_precomp,
        opacities=gaussian_vals['opacity'], scales=gaussian_vals['scales'], rotations=gaussian_vals['rotations'],
        cov3D_precomp=None)
    return {"render": rendered_image, "depth": rendered_depth, "mask": rendered_alpha,
            "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
