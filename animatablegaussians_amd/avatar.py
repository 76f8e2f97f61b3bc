"""The render path of ``AvatarNet`` (reference ``network/avatar.py:161-239``) on the fused MI355X operators.

``AvatarRenderCore`` owns the per-subject constant buffers (canvas mask -> pixel list, canonical Gaussian parameters,
LBS weights) and turns the three StyleUNet outputs + the joint matrices + a camera into an image:

    gather_activate (get_positions / get_others / get_colors)  ->  lbs_transform (transform_cano2live)  ->  render3

It is the piece ``AvatarNet.render`` delegates to once its ``position_net / other_net / color_net`` have produced their
``[1, 2C, S, S]`` maps; the networks themselves (``DualStyleUNet``) stay the reference's own modules, running on the
``fused`` / ``upfirdn2d`` drop-ins.  Buffers come either from the reference's on-disk assets (EXR /
NPY, loaded by the caller) or from ``synthetic`` for tests and benchmarks.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import avatar_ops as ops
from .gaussian_renderer import render3


class AvatarRenderCore(nn.Module):
    def __init__(self, cano_mask: torch.Tensor, xyz: torch.Tensor, opacity_raw: torch.Tensor, scaling_raw: torch.Tensor,
                 rotation_raw: torch.Tensor, lbs: torch.Tensor, max_sh_degree: int = 0):
        super().__init__()
        S2 = cano_mask.shape[1]
        assert cano_mask.shape[0] * 2 == S2, "canvas mask must be [S, 2S] (front | back)"
        self.max_sh_degree = max_sh_degree
        self.register_buffer("pix", ops.mask_to_pix(cano_mask), persistent=False)
        for name, t in (("xyz", xyz), ("opacity_raw", opacity_raw), ("scaling_raw", scaling_raw),
                        ("rotation_raw", rotation_raw), ("lbs", lbs)):
            self.register_buffer(name, t.float().contiguous(), persistent=False)
        assert self.pix.numel() == self.xyz.shape[0] == self.lbs.shape[0]

    @classmethod
    def synthetic(cls, S: int = 1024, J: int = 55, seed: int = 31359, device="cuda") -> "AvatarRenderCore":
        """Buffers of the synthetic avatar (``synth.avatar_map_gaussians``): canonical points on the front|back canvas,
        4-sparse LBS weights, create_from_pcd-style raw parameters."""
        from . import synth
        av = synth.avatar_map_gaussians(S)
        rs = np.random.RandomState(seed)
        N = av["means3D"].shape[0]
        w = rs.normal(0, 1, (N, J)) * 4
        w = np.exp(w - w.max(1, keepdims=True))
        idx = np.argsort(-w, axis=1)[:, :4]
        lbs = np.zeros_like(w)
        np.put_along_axis(lbs, idx, np.take_along_axis(w, idx, 1), 1)
        lbs /= lbs.sum(1, keepdims=True)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
        rot = np.zeros((N, 4), np.float32)
        rot[:, 0] = 1.0
        return cls(t(av["mask"]), t(av["means3D"]), t(np.full((N, 1), np.log(0.1 / 0.9), np.float32)),
                   t(np.log(av["scales"])), t(rot), t(lbs.astype(np.float32)))

    def assemble(self, position_map, other_map, color_map):
        """-> dict of canonical-space Gaussian attributes (what ``render()`` builds at avatar.py:202-209)."""
        pos, opa, sca, rot, col = ops.gather_activate(position_map, other_map, color_map, self.pix, self.xyz,
                                                      self.opacity_raw, self.scaling_raw, self.rotation_raw)
        return {'positions': pos, 'opacity': opa, 'scales': sca, 'rotations': rot, 'colors': col,
                'max_sh_degree': self.max_sh_degree}

    def forward(self, position_map, other_map, color_map, cano2live_jnt_mats, extr, intr, img_w, img_h,
                bg_color: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        g = self.assemble(position_map, other_map, color_map)
        offset = g['positions'] - self.xyz                       # nonrigid_offset, avatar.py:211
        g['positions'], g['rotations'] = ops.lbs_transform(g['positions'], g['rotations'], self.lbs, cano2live_jnt_mats)
        if bg_color is None:
            bg_color = torch.zeros(3, device=position_map.device)
        r = render3(g, bg_color, extr, intr, img_w, img_h)
        return {'rgb_map': r['render'].permute(1, 2, 0), 'mask_map': r['mask'].permute(1, 2, 0), 'offset': offset,
                'depth_map': r['depth'].permute(1, 2, 0), 'posed_gaussians': g}
