"""CPU oracle (torch fp32 + autograd) of the per-Gaussian assembly and skinning steps.  TEST INFRASTRUCTURE ONLY.

Restates, with plain torch ops on the CPU,
  * ``AvatarNet.get_positions / get_others / get_colors``  (reference ``network/avatar.py:93-124``): re-layout of the
    StyleUNet output ``[1, 2C, S, S]`` into the ``[S, 2S, C]`` front|back canvas, boolean-mask gather, and the
    activations ``0.05*delta + xyz``, ``sigmoid``, ``exp``, ``F.normalize`` (``gaussians/gaussian_model.py:53-61``);
  * ``AvatarNet.transform_cano2live`` (``network/avatar.py:84-91``): linear-blend skinning of positions and rotations.

The backward oracle is ``torch.autograd`` of exactly these ops -- that IS the reference's backward for this stage.

PARITY UNPINNED for the two pytorch3d helpers: ``pytorch3d.transforms.quaternion_to_matrix`` /
``matrix_to_quaternion`` (pytorch3d == 0.7.4, ``requirements.txt:9``) are third-party code that is absent from
/root/reference and not installable here; they are restated below from the published 0.7.4 algorithm
(``transforms/rotation_conversions.py``: ``two_s = 2/|q|^2``; four ``sqrt(max(0, 1 +- m00 +- m11 +- m22))`` candidates,
arg-max row, division by ``2*max(q_abs, 0.1)``, no sign standardisation).  Everything else in this file is pinned by
the reference's own Python, which these functions mirror line by line.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def quaternion_to_matrix(quaternions: torch.Tensor) -> torch.Tensor:
    r, i, j, k = torch.unbind(quaternions, -1)
    two_s = 2.0 / (quaternions * quaternions).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k),
            two_s * (i * j - k * r),
            two_s * (i * k + j * r),
            two_s * (i * j + k * r),
            1 - two_s * (i * i + k * k),
            two_s * (j * k - i * r),
            two_s * (i * k - j * r),
            two_s * (j * k + i * r),
            1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(quaternions.shape[:-1] + (3, 3))


def _sqrt_positive_part(x: torch.Tensor) -> torch.Tensor:
    ret = torch.zeros_like(x)
    positive_mask = x > 0
    ret[positive_mask] = torch.sqrt(x[positive_mask])
    return ret


def matrix_to_quaternion(matrix: torch.Tensor) -> torch.Tensor:
    batch_dim = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(matrix.reshape(batch_dim + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(
        torch.stack(
            [1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1))
    quat_by_rijk = torch.stack(
        [
            torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
            torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
            torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
            torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
        ],
        dim=-2,
    )
    flr = torch.tensor(0.1).to(dtype=q_abs.dtype, device=q_abs.device)
    quat_candidates = quat_by_rijk / (2.0 * q_abs[..., None].max(flr))
    return quat_candidates[F.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5, :].reshape(batch_dim + (4,))


def canvas(net_out: torch.Tensor, C: int) -> torch.Tensor:
    """[1, 2C, S, S] -> [S, 2S, C] (front | back along W), network/avatar.py:95-96,108-109,121-122."""
    front, back = torch.split(net_out, [C, C], 1)
    return torch.cat([front, back], 3)[0].permute(1, 2, 0)


def gather_activate(position_map, other_map, color_map, mask, xyz, opacity_raw, scaling_raw, rotation_raw):
    """get_positions + get_others + get_colors on already-computed network outputs.

    Returns (positions [N,3], opacity [N,1], scales [N,3], rotations [N,4], colors [N,3])."""
    positions = 0.05 * canvas(position_map, 3)[mask] + xyz
    others = canvas(other_map, 8)[mask]
    o, s, r = torch.split(others, [1, 3, 4], 1)
    opacity = torch.sigmoid(o + opacity_raw)
    scales = torch.exp(s + scaling_raw)
    rotations = F.normalize(r + rotation_raw)
    colors = canvas(color_map, 3)[mask]
    return positions, opacity, scales, rotations, colors


def transform_cano2live(positions, rotations, lbs, jnt_mats):
    """network/avatar.py:84-91."""
    pt_mats = torch.einsum('nj,jxy->nxy', lbs, jnt_mats)
    live_pos = torch.einsum('nxy,ny->nx', pt_mats[..., :3, :3], positions) + pt_mats[..., :3, 3]
    rot_mats = quaternion_to_matrix(rotations)
    rot_mats = torch.einsum('nxy,nyz->nxz', pt_mats[..., :3, :3], rot_mats)
    return live_pos, matrix_to_quaternion(rot_mats)


def get_pose_map(cano_smpl_map, mask, lbs, jnt_mats_wo_root):
    """network/avatar.py:149-159, literally (einsum blend, scatter into the canvas, F.interpolate 0.5 'nearest', split
    front|back and stack on the channel axis) -> [6, S/2, S/2]."""
    init_points = cano_smpl_map[mask]
    pt_mats = torch.einsum('nj,jxy->nxy', lbs, jnt_mats_wo_root)
    live_pts = torch.einsum('nxy,ny->nx', pt_mats[..., :3, :3], init_points) + pt_mats[..., :3, 3]
    live_pos_map = torch.zeros_like(cano_smpl_map)
    live_pos_map[mask] = live_pts
    live_pos_map = F.interpolate(live_pos_map.permute(2, 0, 1)[None], None, [0.5, 0.5], mode='nearest')[0]
    half = live_pos_map.shape[2] // 2
    return torch.cat(torch.split(live_pos_map, [half, half], 2), 0)


def get_viewdir_feat(cano_smpl_map, cano_nml_map, mask, lbs, jnt_mats, extr, w0, b0, w2, b2, weight_viewdirs=1.0):
    """network/avatar.py:126-147 in eval mode (no view-direction jitter); viewdir_net = Conv2d(1,64,4,2,1) +
    LeakyReLU(0.2) + Conv2d(64,128,4,2,1) (:46-50) with the given weights."""
    init_points, cano_nmls = cano_smpl_map[mask], cano_nml_map[mask]
    pt_mats = torch.einsum('nj,jxy->nxy', lbs, jnt_mats)
    live_pts = torch.einsum('nxy,ny->nx', pt_mats[..., :3, :3], init_points) + pt_mats[..., :3, 3]
    live_nmls = torch.einsum('nxy,ny->nx', pt_mats[..., :3, :3], cano_nmls)
    cam_pos = -torch.matmul(torch.linalg.inv(extr[:3, :3]), extr[:3, 3])
    viewdirs = F.normalize(cam_pos[None] - live_pts, dim=-1, eps=1e-3)
    viewdirs = F.normalize(viewdirs, dim=-1, eps=1e-3)
    viewdirs = (live_nmls * viewdirs).sum(-1)
    viewdirs_map = torch.zeros(*cano_nml_map.shape[:2]).to(viewdirs)
    viewdirs_map[mask] = viewdirs
    viewdirs_map = F.interpolate(viewdirs_map[None, None], None, 0.5, 'nearest')
    half = viewdirs_map.shape[-1] // 2
    outs = []
    for v in torch.split(viewdirs_map, [half, half], -1):
        h = F.leaky_relu(F.conv2d(v, w0, b0, stride=2, padding=1), 0.2)
        outs.append(weight_viewdirs * F.conv2d(h, w2, b2, stride=2, padding=1))
    return outs[0], outs[1]


def normalize_vert_bbox(verts, attris=None, dim=-1, per_axis=False):
    """utils/geo_util.py:104-114."""
    lo, hi = verts.min(dim=dim, keepdim=True)[0], verts.max(dim=dim, keepdim=True)[0]
    v = (verts if attris is None else attris) - 0.5 * (hi + lo)
    return 2 * v / (hi - lo) if per_axis else 2 * v / (hi - lo).max(dim=dim, keepdim=True)[0]


def hand_fuse(g, cano_xyz, left_mano_v, right_mano_v, centre, hand):
    """network/avatar.py:183-200 on dicts with positions / opacity / scales / rotations."""
    wl = torch.sigmoid(2.5 * (normalize_vert_bbox(left_mano_v, attris=cano_xyz, dim=0, per_axis=True)[..., 0:1] + 2.0))
    wr = torch.sigmoid(-2.5 * (normalize_vert_bbox(right_mano_v, attris=cano_xyz, dim=0, per_axis=True)[..., 0:1] - 2.0))
    wl[cano_xyz[..., 1] < centre[1]] = 0.
    wr[cano_xyz[..., 1] < centre[1]] = 0.
    s = torch.maximum(wl + wr, torch.ones_like(wl))
    wl, wr = wl / s, wr / s
    w = wl + wr
    return {k: w * hand[k] + (1.0 - w) * g[k] for k in ('positions', 'opacity', 'scales', 'rotations')}, w
