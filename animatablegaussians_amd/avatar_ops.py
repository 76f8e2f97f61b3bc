"""Autograd operators over the per-Gaussian assembly and skinning kernels (``include/ag_avatar.h``).

``gather_activate`` is the fused equivalent of ``AvatarNet.get_positions`` + ``get_others`` + ``get_colors`` applied to
already-computed StyleUNet outputs (reference ``network/avatar.py:93-124``); ``lbs_transform`` is
``AvatarNet.transform_cano2live`` (``network/avatar.py:84-91``).  Python only allocates and passes pointers.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _chk(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float32 GPU tensor")
    return t.contiguous()


def mask_to_pix(mask: torch.Tensor) -> torch.Tensor:
    """[S, 2S] bool canvas mask -> ascending int32 pixel list (the order ``canvas[mask]`` enumerates)."""
    return torch.nonzero(mask.reshape(-1), as_tuple=False).reshape(-1).to(torch.int32).contiguous()


class _GatherActivate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, position_map, other_map, color_map, pix, xyz, opacity_raw, scaling_raw, rotation_raw):
        L = _lib.lib()
        position_map, other_map, color_map = (_chk(position_map, "position_map"), _chk(other_map, "other_map"),
                                              _chk(color_map, "color_map"))
        xyz, opacity_raw = _chk(xyz, "xyz"), _chk(opacity_raw, "opacity_raw")
        scaling_raw, rotation_raw = _chk(scaling_raw, "scaling_raw"), _chk(rotation_raw, "rotation_raw")
        S = int(position_map.shape[-1])
        if tuple(position_map.shape) != (1, 6, S, S) or tuple(other_map.shape) != (1, 16, S, S) or tuple(color_map.shape) != (1, 6, S, S):
            raise RuntimeError("expected maps of shape [1,6,S,S], [1,16,S,S], [1,6,S,S]")
        N, dev = int(pix.numel()), position_map.device
        f = dict(dtype=torch.float32, device=dev)
        out = [torch.empty((N, c), **f) for c in (3, 1, 3, 4, 3)]
        a = _lib.AgGatherArgs()
        a.N, a.S = N, S
        a.pix = _p(pix)
        a.position_map, a.other_map, a.color_map = _p(position_map), _p(other_map), _p(color_map)
        a.xyz, a.opacity_raw, a.scaling_raw, a.rotation_raw = _p(xyz), _p(opacity_raw), _p(scaling_raw), _p(rotation_raw)
        a.positions, a.opacity, a.scales, a.rotations, a.colors = (_p(t) for t in out)
        with _lib.on_device(dev):
            _lib.check(L.ag_gather_activate_forward(ctypes.byref(a), _stream(dev)), "ag_gather_activate_forward")
        ctx.save_for_backward(position_map, other_map, color_map, pix, xyz, opacity_raw, scaling_raw, rotation_raw)
        return tuple(out)

    @staticmethod
    def backward(ctx, g_pos, g_opa, g_sca, g_rot, g_col):
        L = _lib.lib()
        position_map, other_map, color_map, pix, xyz, opacity_raw, scaling_raw, rotation_raw = ctx.saved_tensors
        dev = position_map.device
        N, S = int(pix.numel()), int(position_map.shape[-1])
        grads = [(_chk(g, "grad") if g is not None else torch.zeros((N, c), dtype=torch.float32, device=dev))
                 for g, c in ((g_pos, 3), (g_opa, 1), (g_sca, 3), (g_rot, 4), (g_col, 3))]
        a = _lib.AgGatherArgs()
        a.N, a.S = N, S
        a.pix = _p(pix)
        a.position_map, a.other_map, a.color_map = _p(position_map), _p(other_map), _p(color_map)
        a.xyz, a.opacity_raw, a.scaling_raw, a.rotation_raw = _p(xyz), _p(opacity_raw), _p(scaling_raw), _p(rotation_raw)
        a.positions, a.opacity, a.scales, a.rotations, a.colors = (_p(t) for t in grads)
        gp, go, gc = (torch.empty_like(m) for m in (position_map, other_map, color_map))
        with _lib.on_device(dev):
            _lib.check(L.ag_gather_activate_backward(ctypes.byref(a), _p(gp), _p(go), _p(gc), _stream(dev)),
                       "ag_gather_activate_backward")
        # the canonical Gaussian parameters are not optimised by the reference trainer (GaussianModel is not an
        # nn.Module, main_avatar.py:55-58 only collects avatar_net.parameters()): no gradient is produced for them
        return gp, go, gc, None, None, None, None, None


def gather_activate(position_map, other_map, color_map, pix, xyz, opacity_raw, scaling_raw, rotation_raw):
    """-> (positions [N,3], opacity [N,1], scales [N,3], rotations [N,4], colors [N,3])."""
    return _GatherActivate.apply(position_map, other_map, color_map, pix, xyz, opacity_raw, scaling_raw, rotation_raw)


class _GatherPart(torch.autograd.Function):
    """One part of the assembly on its own -- what ``AvatarNet.get_positions`` / ``get_others`` / ``get_colors`` return for ONE
    network output (network/avatar.py:93-124), the way the reference trainer's pre-training pass calls them
    (main_avatar.py:126-160).  ``part`` in {'position', 'other', 'color'}; ``raws`` = the canonical parameters that part adds."""

    _SHAPES = {"position": (6, (3,)), "other": (16, (1, 3, 4)), "color": (6, (3,))}

    @staticmethod
    def forward(ctx, part, net_map, pix, *raws):
        L = _lib.lib()
        ch, outs = _GatherPart._SHAPES[part]
        net_map = _chk(net_map, part + "_map")
        S = int(net_map.shape[-1])
        if tuple(net_map.shape) != (1, ch, S, S):
            raise RuntimeError(f"expected a {part} map of shape [1,{ch},S,S]")
        raws = tuple(_chk(r, "canonical parameter") for r in raws)
        N, dev = int(pix.numel()), net_map.device
        out = [torch.empty((N, c), dtype=torch.float32, device=dev) for c in outs]
        a = _lib.AgGatherArgs()
        a.N, a.S, a.pix = N, S, _p(pix)
        if part == "position":
            a.position_map, a.xyz, a.positions = _p(net_map), _p(raws[0]), _p(out[0])
        elif part == "other":
            a.other_map = _p(net_map)
            a.opacity_raw, a.scaling_raw, a.rotation_raw = (_p(r) for r in raws)
            a.opacity, a.scales, a.rotations = (_p(t) for t in out)
        else:
            a.color_map, a.colors = _p(net_map), _p(out[0])
        with _lib.on_device(dev):
            _lib.check(L.ag_gather_activate_forward(ctypes.byref(a), _stream(dev)), "ag_gather_activate_forward")
        ctx.part = part
        ctx.save_for_backward(net_map, pix, *raws)
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        L = _lib.lib()
        net_map, pix, *raws = ctx.saved_tensors
        part, dev = ctx.part, net_map.device
        N, S = int(pix.numel()), int(net_map.shape[-1])
        outs = _GatherPart._SHAPES[part][1]
        grads = [(_chk(g, "grad") if g is not None else torch.zeros((N, c), dtype=torch.float32, device=dev)) for g, c in zip(grads, outs)]
        gm = torch.empty_like(net_map)
        a = _lib.AgGatherArgs()
        a.N, a.S, a.pix = N, S, _p(pix)
        null = ctypes.c_void_p(None)
        gp = go = gc = null
        if part == "position":
            a.positions, gp = _p(grads[0]), _p(gm)
        elif part == "other":
            a.other_map = _p(net_map)
            a.opacity_raw, a.scaling_raw, a.rotation_raw = (_p(r) for r in raws)
            a.opacity, a.scales, a.rotations = (_p(t) for t in grads)
            go = _p(gm)
        else:
            a.colors, gc = _p(grads[0]), _p(gm)
        with _lib.on_device(dev):
            _lib.check(L.ag_gather_activate_backward(ctypes.byref(a), gp, go, gc, _stream(dev)), "ag_gather_activate_backward")
        return (None, gm, None) + (None,) * len(raws)


def gather_positions(position_map, pix, xyz):
    """``0.05 * position_map[mask] + xyz`` -> [N,3]  (network/avatar.py:93-104)."""
    return _GatherPart.apply("position", position_map, pix, xyz)[0]


def gather_others(other_map, pix, opacity_raw, scaling_raw, rotation_raw):
    """-> (opacity [N,1], scales [N,3], rotations [N,4]) = sigmoid / exp / normalize of map + raw  (network/avatar.py:106-117)."""
    return _GatherPart.apply("other", other_map, pix, opacity_raw, scaling_raw, rotation_raw)


def gather_colors(color_map, pix):
    """``color_map[mask]`` -> [N,3]  (network/avatar.py:119-124)."""
    return _GatherPart.apply("color", color_map, pix)[0]


@torch.no_grad()
def canonical_activations(pix, S, opacity_raw, scaling_raw, rotation_raw):
    """``GaussianModel.get_opacity / get_scaling / get_rotation`` (gaussians/gaussian_model.py:115-147): the activations of
    the canonical parameters alone (the assembly kernel with a null network map)."""
    N, dev = int(pix.numel()), opacity_raw.device
    out = [torch.empty((N, c), dtype=torch.float32, device=dev) for c in (1, 3, 4)]
    a = _lib.AgGatherArgs()
    a.N, a.S, a.pix = N, int(S), _p(pix)
    a.opacity_raw, a.scaling_raw, a.rotation_raw = (_p(_chk(r, "canonical parameter")) for r in (opacity_raw, scaling_raw, rotation_raw))
    a.opacity, a.scales, a.rotations = (_p(t) for t in out)
    with _lib.on_device(dev):
        _lib.check(_lib.lib().ag_gather_activate_forward(ctypes.byref(a), _stream(dev)), "ag_gather_activate_forward")
    return tuple(out)


class SparseLbs:
    """Sparse form of the blend-weight rows ``lbs [N, J]`` for the skinning kernels: K = the largest number of non-zeros in a row,
    ``idx [K, N]`` uint8 joints (ascending within a row) and ``w [K, N]`` weights, zero padded.  Rows sampled from the SMPL-X
    skinning weights by barycentric interpolation have <= 12 non-zeros of 55 (gen_data/gen_pos_maps.py:24-39,132): the kernels
    then read 5 K instead of 220 bytes per Gaussian; results equal the dense path's (exact zeros do not change a sum).
    ``SparseLbs.build`` returns None when a row has more than ``max_k`` non-zeros (weights sampled from the diffused weight volume,
    gen_pos_maps.py:129-130): the dense kernels are used then."""

    def __init__(self, idx, w):
        self.idx, self.w, self.K = idx, w, int(idx.shape[0])

    @staticmethod
    @torch.no_grad()
    def build(lbs: torch.Tensor, max_k: int = 16):
        N, J = lbs.shape
        if J > 256 or N == 0:
            return None
        nz = lbs != 0
        K = int(nz.sum(1).max())
        if K == 0 or K > max_k:
            return None
        # the K first non-zero columns of every row in ascending order: sort columns by (is zero, column index)
        key = (~nz).to(torch.int32) * J + torch.arange(J, device=lbs.device, dtype=torch.int32)[None]
        cols = torch.argsort(key, dim=1, stable=True)[:, :K]
        w = torch.gather(lbs, 1, cols)                      # zeros where a row has fewer than K non-zeros
        return SparseLbs(cols.to(torch.uint8).t().contiguous(), w.to(torch.float32).t().contiguous())


def _lbs_args(N, J, lbs, jnt_mats, positions, rotations, out_p, out_r, sparse):
    a = _lib.AgLbsArgs()
    a.N, a.J = N, J
    a.jnt_mats, a.positions, a.rotations = _p(jnt_mats), _p(positions), _p(rotations)
    a.out_positions, a.out_rotations = _p(out_p), _p(out_r)
    if sparse is not None:
        a.sp_idx, a.sp_w, a.K = _p(sparse.idx), _p(sparse.w), sparse.K
    else:
        a.lbs = _p(lbs)
    return a


class _LbsTransform(torch.autograd.Function):
    @staticmethod
    def forward(ctx, positions, rotations, lbs, jnt_mats, sparse):
        L = _lib.lib()
        positions, rotations = _chk(positions, "positions"), _chk(rotations, "rotations")
        lbs, jnt_mats = _chk(lbs, "lbs"), _chk(jnt_mats, "jnt_mats")
        N, J, dev = int(positions.shape[0]), int(lbs.shape[1]), positions.device
        if tuple(jnt_mats.shape) != (J, 4, 4) or lbs.shape[0] != N or tuple(rotations.shape) != (N, 4):
            raise RuntimeError("expected positions [N,3], rotations [N,4], lbs [N,J], jnt_mats [J,4,4]")
        if sparse is not None and tuple(sparse.idx.shape) != (sparse.K, N):
            raise RuntimeError("sparse blend weights do not match the number of Gaussians")
        out_p, out_r = torch.empty_like(positions), torch.empty_like(rotations)
        a = _lbs_args(N, J, lbs, jnt_mats, positions, rotations, out_p, out_r, sparse)
        with _lib.on_device(dev):
            _lib.check(L.ag_lbs_forward(ctypes.byref(a), _stream(dev)), "ag_lbs_forward")
        ctx.save_for_backward(positions, rotations, lbs, jnt_mats)
        ctx.sparse = sparse
        return out_p, out_r

    @staticmethod
    def backward(ctx, g_p, g_r):
        L = _lib.lib()
        positions, rotations, lbs, jnt_mats = ctx.saved_tensors
        N, J, dev = int(positions.shape[0]), int(lbs.shape[1]), positions.device
        g_p = _chk(g_p, "grad") if g_p is not None else torch.zeros_like(positions)
        g_r = _chk(g_r, "grad") if g_r is not None else torch.zeros_like(rotations)
        dp, dr = torch.empty_like(positions), torch.empty_like(rotations)
        a = _lbs_args(N, J, lbs, jnt_mats, positions, rotations, g_p, g_r, ctx.sparse)
        with _lib.on_device(dev):
            _lib.check(L.ag_lbs_backward(ctypes.byref(a), _p(dp), _p(dr), _stream(dev)), "ag_lbs_backward")
        return dp, dr, None, None, None   # lbs weights and joint matrices are data, not parameters


def lbs_transform(positions, rotations, lbs, jnt_mats, sparse=None):
    """-> (live positions [N,3], live rotations [N,4]).  ``sparse``: ``SparseLbs.build(lbs)`` (or None: dense rows)."""
    return _LbsTransform.apply(positions, rotations, lbs, jnt_mats, sparse)


def hand_fuse(positions, opacity, scales, rotations, xyz, left_mano_v, right_mano_v, centre, hand_positions, hand_opacity,
              hand_scales, hand_rotations):
    """Eval-time hand fusion, network/avatar.py:183-200 (no autograd: the reference runs it under ``torch.no_grad`` in test
    mode).  Returns new (positions, opacity, scales, rotations); the inputs are left untouched."""
    dev = xyz.device
    outs = [_chk(t.detach(), n).clone() for t, n in ((positions, "positions"), (opacity, "opacity"), (scales, "scales"),
                                                     (rotations, "rotations"))]
    boxes = []
    for v in (left_mano_v, right_mano_v):            # normalize_vert_bbox only needs the x extent of the hand vertices
        vx = _chk(v.to(dev), "mano vertices")[:, 0]
        boxes.append(torch.stack([vx.min(), vx.max()]))
    a = _lib.AgHandFuseArgs()
    a.N = int(xyz.shape[0])
    keep = [_chk(xyz, "xyz"), boxes[0], boxes[1], _chk(torch.as_tensor(centre, dtype=torch.float32).to(dev).reshape(3), "centre"),
            _chk(hand_positions, "hand_positions"), _chk(hand_opacity, "hand_opacity"), _chk(hand_scales, "hand_scales"),
            _chk(hand_rotations, "hand_rotations")]
    for name, t in zip(("xyz", "left_box", "right_box", "centre", "hand_positions", "hand_opacity", "hand_scales", "hand_rotations",
                        "positions", "opacity", "scales", "rotations"), keep + outs):
        if name in ("hand_positions", "hand_opacity", "hand_scales", "hand_rotations", "positions", "opacity", "scales", "rotations") \
                and t.shape[0] != a.N:
            raise RuntimeError(f"hand_fuse: {name} has {t.shape[0]} rows, expected {a.N}")
        setattr(a, name, t.data_ptr())
    with _lib.on_device(dev):
        _lib.check(_lib.lib().ag_hand_fuse(ctypes.byref(a), _stream(dev)), "ag_hand_fuse")
    return tuple(outs)
