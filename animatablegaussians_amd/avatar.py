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

import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import avatar_ops as ops
from .gaussian_renderer import render3


class AvatarRenderCore(nn.Module):
    def __init__(self, cano_mask: torch.Tensor, xyz: torch.Tensor, opacity_raw: torch.Tensor, scaling_raw: torch.Tensor,
                 rotation_raw: torch.Tensor, lbs: torch.Tensor, max_sh_degree: int = 0):
        super().__init__()
        S2 = cano_mask.shape[1]
        assert cano_mask.shape[0] * 2 == S2, "canvas mask must be [S, 2S] (front | back)"
        self.max_sh_degree = max_sh_degree
        self.map_side = int(cano_mask.shape[0])
        self.register_buffer("pix", ops.mask_to_pix(cano_mask), persistent=False)
        for name, t in (("xyz", xyz), ("opacity_raw", opacity_raw), ("scaling_raw", scaling_raw),
                        ("rotation_raw", rotation_raw), ("lbs", lbs)):
            self.register_buffer(name, t.float().contiguous(), persistent=False)
        assert self.pix.numel() == self.xyz.shape[0] == self.lbs.shape[0]
        self.lbs_sparse = ops.SparseLbs.build(self.lbs) if self.lbs.is_cuda else None      # None: dense rows (or a CPU-side construction)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.lbs_sparse = ops.SparseLbs.build(self.lbs) if self.lbs.is_cuda else None      # follows the buffers across .to() / .cuda()
        return out

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
        g['positions'], g['rotations'] = ops.lbs_transform(g['positions'], g['rotations'], self.lbs, cano2live_jnt_mats, self.lbs_sparse)
        if bg_color is None:
            bg_color = torch.zeros(3, device=position_map.device)
        r = render3(g, bg_color, extr, intr, img_w, img_h)
        return {'rgb_map': r['render'].permute(1, 2, 0), 'mask_map': r['mask'].permute(1, 2, 0), 'offset': offset,
                'depth_map': r['depth'].permute(1, 2, 0), 'posed_gaussians': g}


def _knn3_log_scale(points: np.ndarray) -> np.ndarray:
    """``log sqrt(clamp_min(mean squared distance to the 3 nearest neighbours, 1e-7))`` -- the scale initialiser of
    ``GaussianModel.create_from_pcd`` (gaussians/gaussian_model.py:170-171; pytorch3d ``knn_points(K=4)[..., 1:]``).
    Init-time only, on the host (scipy k-d tree)."""
    from scipy.spatial import cKDTree
    d, _ = cKDTree(points).query(points, k=4)
    dist2 = np.maximum((d[:, 1:].astype(np.float64) ** 2).mean(-1), 1e-7)
    return np.log(np.sqrt(dist2)).astype(np.float32)


class _Graphed:
    """A no-grad callable captured once in a hipGraph and replayed: ``fn(*tensors) -> tuple of tensors`` with static
    shapes.  Inputs are copied into the capture's static buffers; the returned tensors are the capture's static outputs
    (valid until the next call).  The ~1000 kernel launches of a StyleUNet forward become one graph launch."""

    def __init__(self, fn, example_inputs):
        self.static_in = [t.clone() if t is not None else None for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):                      # warm-up outside the capture (allocator, lazy module state)
                fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            out = fn(*self.static_in)
        self.static_out = out if isinstance(out, (tuple, list)) else (out,)

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            if dst is not None and dst is not src:
                dst.copy_(src)
        self.graph.replay()
        return self.static_out


class CanoGaussianModel:
    """The read surface of the reference's canonical ``GaussianModel`` (``gaussians/gaussian_model.py:53-61,115-147``) that
    ``AvatarNet`` and the trainer's pre-training pass use (``network/avatar.py:100,113-115``, ``main_avatar.py:134-153``):
    ``get_xyz``, ``get_opacity``, ``get_scaling``, ``get_rotation``, their ``*_raw`` forms and the three activation callables.
    A view over ``AvatarRenderCore``'s buffers -- as in the reference it is not an ``nn.Module`` and nothing in it is trained
    (main_avatar.py:55-58 only collects ``avatar_net.parameters()``).  The activated getters run on the assembly kernel
    (null network map) and are cached until a raw buffer is modified in place."""

    def __init__(self, core: AvatarRenderCore, sh_degree: int = 0):
        self._core = core
        self.max_sh_degree = sh_degree
        self.active_sh_degree = 0
        self.spatial_lr_scale = 2.5                                 # network/avatar.py:32
        self.opacity_activation = torch.sigmoid                     # gaussian_model.py:53-61
        self.scaling_activation = torch.exp
        self.scaling_inverse_activation = torch.log
        self.rotation_activation = F.normalize
        self.inverse_opacity_activation = lambda x: torch.log(x / (1 - x))
        self._act = None

    def _activated(self):
        c = self._core
        key = (c.opacity_raw.data_ptr(), c.opacity_raw._version, c.scaling_raw.data_ptr(), c.scaling_raw._version,
               c.rotation_raw.data_ptr(), c.rotation_raw._version)
        if self._act is None or self._act[0] != key:
            self._act = (key, ops.canonical_activations(c.pix, c.map_side, c.opacity_raw, c.scaling_raw, c.rotation_raw))
        return self._act[1]

    get_xyz = property(lambda self: self._core.xyz)
    get_opacity_raw = property(lambda self: self._core.opacity_raw)
    get_scaling_raw = property(lambda self: self._core.scaling_raw)
    get_rotation_raw = property(lambda self: self._core.rotation_raw)
    get_opacity = property(lambda self: self._activated()[0])
    get_scaling = property(lambda self: self._activated()[1])
    get_rotation = property(lambda self: self._activated()[2])


_BG_CACHE = {}


def _bg_tensor(bg_color, dev):
    """The background colour on the device, uploaded once per (colour, device): a pageable host-to-device copy at the head of every render call
    is a host wait for everything queued before it (the previous step's backward and optimizer)."""
    if isinstance(bg_color, torch.Tensor):
        return bg_color.to(device=dev, dtype=torch.float32)
    key = (tuple(float(c) for c in np.asarray(bg_color, dtype=np.float64).reshape(-1)), str(dev))
    t = _BG_CACHE.get(key)
    if t is None:
        if len(_BG_CACHE) > 64:
            _BG_CACHE.clear()
        t = _BG_CACHE[key] = torch.as_tensor(np.asarray(bg_color), dtype=torch.float32).to(dev)
    return t


class AvatarNet(nn.Module):
    """Re-host of the reference's ``network.avatar.AvatarNet`` (``network/avatar.py:16-239``) on this package's kernels:
    three ``DualStyleUNet`` (position / other / colour), the view-direction encoder, the fused per-Gaussian assembly,
    LBS and the rasterizer.  ``render(items, bg_color)`` takes the same ``items`` dict (``smpl_pos_map``,
    ``cano2live_jnt_mats``, ``extr``, ``intr``, ``img_w``, ``img_h``) and returns the same dict (``rgb_map``,
    ``mask_map``, ``offset``, ``pos_map`` [+ ``cano_tex_map``, ``posed_gaussians`` in eval mode]).

    Differences, all deliberate: the per-subject assets are passed in as tensors (``cano_smpl_map`` [S, 2S, 3],
    ``lbs`` [N, J], optional ``cano_nml_map``) or read by ``from_data_dir`` with the OpenCV-free EXR reader (``exr.py``)
    instead of through ``config.opt`` (``dropin/avatar_module.py`` is the ``config``-reading constructor the reference trainer
    imports).  ``state_dict()`` / ``load_state_dict()`` / ``parameters()`` have the reference module's keys and order, so ``net.pt``
    and ``optm.pt`` written by either implementation load in the other (main_avatar.py:777-813).
    """

    def __init__(self, opt: Optional[dict] = None, *, cano_smpl_map: torch.Tensor, lbs: torch.Tensor,
                 cano_nml_map: Optional[torch.Tensor] = None, device="cuda"):
        super().__init__()
        from .styleunet import DualStyleUNet
        opt = dict(opt or {})
        self.opt = opt
        self.hand_mask = self.hand_positions = self.hand_opacity = self.hand_scales = self.hand_rotations = self.hand_colors = None
        self.random_style = opt.get('random_style', False)
        self.with_viewdirs = opt.get('with_viewdirs', True) and cano_nml_map is not None
        self.max_sh_degree = 0
        dev = torch.device(device)

        cano_smpl_map = cano_smpl_map.to(torch.float32)
        mask = torch.linalg.norm(cano_smpl_map, dim=-1) > 0.                                  # :28
        init_points = cano_smpl_map[mask]                                                       # :29
        N = init_points.shape[0]
        if lbs.shape[0] != N:
            raise ValueError(f"lbs has {lbs.shape[0]} rows, the canonical map has {N} valid pixels")
        log_scale = torch.from_numpy(_knn3_log_scale(init_points.cpu().numpy()))
        rot = torch.zeros(N, 4)
        rot[:, 0] = 1.0
        self.core = AvatarRenderCore(mask.to(dev), init_points.to(dev), torch.full((N, 1), float(np.log(0.1 / 0.9))).to(dev),
                                     log_scale[:, None].repeat(1, 3).to(dev), rot.to(dev), lbs.to(dev), self.max_sh_degree)
        self.register_buffer("cano_smpl_mask", mask.to(dev), persistent=False)
        self.map_shape = tuple(cano_smpl_map.shape[:2])

        self.color_net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2)     # :34-36
        self.position_net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2)
        self.other_net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=8, out_size=1024, style_dim=512, n_mlp=2)
        style = torch.ones(1, 512, dtype=torch.float32) / np.sqrt(512)                                             # :38-40
        for n in ("color_style", "position_style", "other_style"):
            self.register_buffer(n, style.clone(), persistent=False)

        if self.with_viewdirs:
            nml = cano_nml_map.to(torch.float32)
            self.register_buffer("cano_nmls", nml[mask].to(dev), persistent=False)                                 # :45
            # :46-50.  The torch modules only HOLD the parameters (reference names, initialisation and registration order:
            # ``viewdir_net.{0,2}.{weight,bias}`` after the three networks); get_viewdir_feat runs them on ag_conv.
            self.viewdir_net = nn.Sequential(nn.Conv2d(1, 64, 4, 2, 1), nn.LeakyReLU(0.2, inplace=True), nn.Conv2d(64, 128, 4, 2, 1))
        self.to(dev)
        self.cano_gaussian_model = CanoGaussianModel(self.core, self.max_sh_degree)

    # ---- construction helpers -------------------------------------------------------------------------------------
    @classmethod
    def from_data_dir(cls, opt: Optional[dict], data_dir: str, device="cuda") -> "AvatarNet":
        """The constructor's file reads of network/avatar.py:27-43: ``<data_dir>/smpl_pos_map/cano_smpl_pos_map.exr``,
        ``init_pts_lbs.npy`` and (with view directions) ``cano_smpl_nml_map.exr``."""
        from . import exr
        d = os.path.join(data_dir, 'smpl_pos_map')
        cano = torch.from_numpy(exr.imread(os.path.join(d, 'cano_smpl_pos_map.exr')))
        lbs = torch.from_numpy(np.load(os.path.join(d, 'init_pts_lbs.npy'))).to(torch.float32)
        nml = None
        if (opt or {}).get('with_viewdirs', True):
            nml = torch.from_numpy(exr.imread(os.path.join(d, 'cano_smpl_nml_map.exr')))
        return cls(opt, cano_smpl_map=cano, lbs=lbs, cano_nml_map=nml, device=device)

    @classmethod
    def synthetic(cls, opt: Optional[dict] = None, S: int = 1024, J: int = 55, seed: int = 31359, device="cuda") -> "AvatarNet":
        """The synthetic subject of SURVEY.md 8d config 3 (``synth.avatar_map_gaussians`` canvas, 4-sparse LBS weights)."""
        from . import synth
        av = synth.avatar_map_gaussians(S)
        mask = av["mask"]
        cano = np.zeros(mask.shape + (3,), np.float32)
        pts = av["means3D"].astype(np.float32).copy()
        pts[np.linalg.norm(pts, axis=-1) == 0] += 1e-6          # the mask is recovered from |xyz| > 0
        cano[mask] = pts
        nml = np.zeros_like(cano)
        nml[..., 2] = 1.0
        nml[:, S:, 2] = -1.0                                    # front canvas faces +z, back canvas -z
        nml *= mask[..., None]
        rs = np.random.RandomState(seed)
        N = pts.shape[0]
        w = rs.normal(0, 1, (N, J)) * 4
        w = np.exp(w - w.max(1, keepdims=True))
        idx = np.argsort(-w, axis=1)[:, :4]
        lbs = np.zeros_like(w)
        np.put_along_axis(lbs, idx, np.take_along_axis(w, idx, 1), 1)
        lbs /= lbs.sum(1, keepdims=True)
        return cls(opt, cano_smpl_map=torch.from_numpy(cano), lbs=torch.from_numpy(lbs.astype(np.float32)),
                   cano_nml_map=torch.from_numpy(nml), device=device)

    @torch.no_grad()
    def load_reference_state_dict(self, sd, strict=True):
        """``sd`` = the reference's ``AvatarNet.state_dict()`` (``net.pt['avatar_net']``, main_avatar.py:778-786), with or without
        its constant FIR / Haar buffers.  ``load_state_dict`` / ``state_dict`` themselves use the reference's exact keys."""
        per_net = {"color_net": {}, "position_net": {}, "other_net": {}}
        vd = {}
        for key, value in sd.items():
            head, _, rest = key.partition(".")
            if head in per_net:
                per_net[head][rest] = value
            elif head == "viewdir_net":
                vd[rest] = value
            elif strict:
                raise RuntimeError(f"unexpected key in reference state_dict: {key}")
        for name, part in per_net.items():
            getattr(self, name).load_reference_state_dict(part, strict=strict)
        if self.with_viewdirs:
            self.viewdir_net.load_state_dict(vd, strict=strict)

    # ---- the reference's methods ----------------------------------------------------------------------------------
    @property
    def init_points(self):
        return self.core.xyz

    @property
    def lbs(self):
        return self.core.lbs

    @torch.no_grad()
    def _blend_points(self, jnt_mats, vectors=None):
        """(no grad) LBS of the canonical points, optionally of direction vectors with the rotation part only
        (:127-129,151-152) -- on the skinning kernel: the reference's einsum over [N, 55] x [55, 16] lands on a skinny
        library GEMM that takes 3.5 ms per call at N = 268 k (profiles/r01f), the fused kernel ~20 us."""
        N = self.core.xyz.shape[0]
        if not hasattr(self, "_unit_quat") or self._unit_quat.shape[0] != N:
            q = torch.zeros(N, 4, device=self.core.xyz.device)
            q[:, 0] = 1.0
            self._unit_quat = q
        mats = jnt_mats.to(torch.float32).contiguous()
        pts, _ = ops.lbs_transform(self.core.xyz, self._unit_quat, self.core.lbs, mats, self.core.lbs_sparse)
        if vectors is None:
            return pts, None
        rot_only = mats.clone()
        rot_only[:, :3, 3] = 0.0
        vec, _ = ops.lbs_transform(vectors.contiguous(), self._unit_quat, self.core.lbs, rot_only, self.core.lbs_sparse)
        return pts, vec

    def _mask_index(self):
        """Flat int64 positions of the canvas mask in ``canvas[mask]`` order (= core.pix): a boolean-mask assignment makes torch
        compute them with nonzero() on every call, i.e. a device-to-host synchronisation per pose map and per view-direction map."""
        idx = getattr(self, "_mask_index64", None)
        if idx is None or idx.device != self.core.pix.device:
            idx = self._mask_index64 = self.core.pix.long()
        return idx

    @torch.no_grad()
    def get_pose_map(self, items):
        """Live position map [6, S/2, S/2] from ``cano2live_jnt_mats_woRoot``  (:149-159)."""
        live_pts, _ = self._blend_points(items['cano2live_jnt_mats_woRoot'])
        H, W = self.map_shape
        live = torch.zeros(H, W, 3, device=live_pts.device)
        live.view(-1, 3).index_copy_(0, self._mask_index(), live_pts)   # == live[self.cano_smpl_mask] = live_pts without the host sync
        live = live.permute(2, 0, 1)[:, ::2, ::2]                  # F.interpolate(scale 0.5, 'nearest') == every 2nd sample
        live = torch.cat(torch.split(live, [W // 4, W // 4], 2), 0).contiguous()
        items.update({'smpl_pos_map': live})
        return live

    def get_viewdir_feat(self, items):
        """Cosine between surface normal and view direction, as two [1, 128, S/8, S/8] feature maps  (:126-147)."""
        from . import conv as agc
        from .styleunet_ops import fused_leaky_relu
        with torch.no_grad():
            mats = items['cano2live_jnt_mats']
            cache = getattr(self, "_pose_cache", None)          # camera-independent: reused across the views of one pose
            if cache is None or cache[0] is not mats or cache[1] != mats._version:
                cache = (mats, mats._version) + self._blend_points(mats, self.cano_nmls)
                self._pose_cache = cache
            live_pts, live_nmls = cache[2], cache[3]
            extr = items['extr']
            cam_pos = -torch.matmul(torch.linalg.inv_ex(extr[:3, :3])[0], extr[:3, 3])   # inv without the singularity read-back
            viewdirs = F.normalize(cam_pos[None] - live_pts, dim=-1, eps=1e-3)
            if self.training:
                viewdirs = viewdirs + torch.randn_like(viewdirs) * 0.1
            viewdirs = F.normalize(viewdirs, dim=-1, eps=1e-3)
            cosine = (live_nmls * viewdirs).sum(-1)
            H, W = self.map_shape
            vmap = torch.zeros(H, W, device=cosine.device)
            vmap.view(-1).index_copy_(0, self._mask_index(), cosine)     # == vmap[self.cano_smpl_mask] = cosine, no host sync
            vmap = vmap[None, None, ::2, ::2]
            front, back = (t.contiguous() for t in torch.split(vmap, [W // 4, W // 4], -1))
        weight = self.opt.get('weight_viewdirs', 1.)
        feats = []
        for v in (front, back):
            c0, c2 = self.viewdir_net[0], self.viewdir_net[2]
            h = agc.conv2d(v, c0.weight, bias=c0.bias, stride=2, padding=1)
            h = fused_leaky_relu(h, None, 0.2, 1.0)
            h = agc.conv2d(h, c2.weight, bias=c2.bias, stride=2, padding=1)
            feats.append(h if weight == 1. else weight * h)        # (the default weight 1: no scaling pass forward or backward)
        return feats[0], feats[1]

    def _concurrently(self, fns, shared=()):
        """Run independent sub-networks on their own HIP streams (the first on the current one) and join: their small
        layers interleave on the CUs.  Autograd replays each node on its recording stream, so the backward overlaps too.
        ``shared``: tensors of the current stream the side streams read (registered with them for the allocator)."""
        import os
        cur = torch.cuda.current_stream()
        if os.environ.get("AG_SINGLE_STREAM") == "1" or len(fns) == 1 or torch.cuda.is_current_stream_capturing():
            return [f() for f in fns]
        pool = getattr(self, "_net_streams", None)
        if pool is None or len(pool) < len(fns) - 1 or pool[0].device != cur.device:
            pool = self._net_streams = [torch.cuda.Stream(cur.device) for _ in range(len(fns) - 1)]
        outs = [None] * len(fns)
        for i, f in enumerate(fns[1:]):
            pool[i].wait_stream(cur)
            for t in shared:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(pool[i])
            with torch.cuda.stream(pool[i]):
                outs[i + 1] = f()
        outs[0] = fns[0]()
        for i in range(len(fns) - 1):
            cur.wait_stream(pool[i])
            o = outs[i + 1]
            for t in (o if isinstance(o, (tuple, list)) else (o,)):
                if isinstance(t, torch.Tensor):
                    t.record_stream(cur)
        return outs

    def enable_graphs(self, on: bool = True):
        """Eval-mode option: run the three networks from captured hipGraphs (one launch per network instead of ~1000).
        Results are bit-identical to eager.  In steady state the eager path is already GPU-bound (35.6 vs 35.5 ms per view,
        bench_avatar.py --infer [--graphs]); the capture pays when the host is the bottleneck (a cold or busy CPU:
        15.3 -> 12.4 ms per network in profiles/graph_probe.py).  Captures lazily on the next eval
        call with gradients disabled; the captures hold parameter ADDRESSES, so call ``enable_graphs(False)`` before moving
        the module or swapping parameter tensors (in-place updates such as ``load_reference_state_dict`` are fine)."""
        self._use_graphs = bool(on)
        self._graphs = {}

    def _graphs_active(self):
        # eval mode only: the captures bake ``self.color_style`` in, and ``random_style`` (a fresh colour style per call, :469) applies in training
        # mode only -- so a captured pass and an eager one of the same mode always use the same style
        return getattr(self, "_use_graphs", False) and not self.training and not torch.is_grad_enabled()

    def _graphed(self, key, fn, inputs):
        g = self._graphs.get(key)
        if g is None:
            g = self._graphs[key] = _Graphed(fn, inputs)
        return g(*inputs)

    def _grouped_nets(self):
        """The grouped executor over (position, colour, other) -- networks with equal ToRGB widths next to each other -- or None when
        AG_GROUPED=0 / ``set_grouped(False)`` selects the one-network-at-a-time path (A/B measurements, the equality test)."""
        if not getattr(self, "_use_grouped", os.environ.get("AG_GROUPED") != "0"):
            return None
        g = getattr(self, "_grouped", None)
        if g is None:
            from .grouped import GroupedStyleUNets
            nets = [self.position_net, self.color_net, self.other_net]
            g = self._grouped = GroupedStyleUNets(nets) if GroupedStyleUNets.supported(nets) else False
        return g or None

    def set_grouped(self, on: bool) -> bool:
        prev = getattr(self, "_use_grouped", os.environ.get("AG_GROUPED") != "0")
        self._use_grouped = bool(on)
        return prev

    def get_maps(self, pose_map, front_viewdirs=None, back_viewdirs=None):
        """The three StyleUNet evaluations of ``get_positions`` / ``get_others`` / ``get_colors``  (:93-124), raw maps."""
        x = pose_map[None].contiguous()
        if self._graphs_active() and self._grouped_nets() is not None:        # ONE capture of the grouped chain of all three networks
            def all_fn(p, f, b):
                pm, cm, om = self._grouped_nets().forward([self.position_style, self.color_style, self.other_style], p,
                                                          {1: (f, b)} if f is not None else None)
                return pm, om, cm
            return tuple(self._graphed(("grouped", front_viewdirs is not None), all_fn, [x, front_viewdirs, back_viewdirs]))
        if self._graphs_active():
            (position_map,) = self._graphed("position", lambda p: self.position_net([self.position_style], p, randomize_noise=False)[0], [x])
            (other_map,) = self._graphed("other", lambda p: self.other_net([self.other_style], p, randomize_noise=False)[0], [x])
            (color_map,) = self._graphed(
                ("color", front_viewdirs is not None),
                lambda p, f, b: self.color_net([self.color_style], p, randomize_noise=False, view_feature1=f, view_feature2=b)[0],
                [x, front_viewdirs, back_viewdirs])
            return position_map, other_map, color_map
        color_style = torch.rand_like(self.color_style) if self.random_style and self.training else self.color_style
        grouped = self._grouped_nets()
        if grouped is not None:                      # the three networks as ONE launch chain (grouped.py): G = 3 encoders, G = 6 decoders
            vf = {1: (front_viewdirs, back_viewdirs)} if front_viewdirs is not None else None
            # (frozen: in eval mode the three styles are this module's buffers -- the weights' modulation / maxima / packed images persist between frames)
            position_map, color_map, other_map = grouped.forward([self.position_style, color_style, self.other_style], x, vf,
                                                                 frozen=color_style is self.color_style and not self.training)
            return position_map, other_map, color_map
        position_map, other_map, color_map = self._concurrently([
            lambda: self.position_net([self.position_style], x, randomize_noise=False)[0],
            lambda: self.other_net([self.other_style], x, randomize_noise=False)[0],
            lambda: self.color_net([color_style], x, randomize_noise=False, view_feature1=front_viewdirs,
                                   view_feature2=back_viewdirs)[0]], shared=[x, color_style, front_viewdirs, back_viewdirs])
        return position_map, other_map, color_map

    @staticmethod
    def _canvas(m):
        """[1, 2C, S, S] network output -> [S, 2S, C] front|back canvas (the layout ``pos_map`` / ``cano_tex_map`` have)."""
        c = m.shape[1] // 2
        return torch.cat([m[:, :c], m[:, c:]], 3)[0].permute(1, 2, 0)

    # ---- the per-network accessors the trainer's pre-training pass calls one at a time (main_avatar.py:126-160) ----------------
    def get_positions(self, pose_map, return_map=False):
        """:93-104 -> positions [N,3] (= 0.05 * position_map[mask] + cano xyz) [, position_map [S, 2S, 3]]."""
        position_map = self.position_net([self.position_style], pose_map[None].contiguous(), randomize_noise=False)[0]
        positions = ops.gather_positions(position_map, self.core.pix, self.core.xyz)
        return (positions, self._canvas(position_map)) if return_map else positions

    def get_others(self, pose_map):
        """:106-117 -> (opacity [N,1], scales [N,3], rotations [N,4])."""
        other_map = self.other_net([self.other_style], pose_map[None].contiguous(), randomize_noise=False)[0]
        return ops.gather_others(other_map, self.core.pix, self.core.opacity_raw, self.core.scaling_raw, self.core.rotation_raw)

    def get_colors(self, pose_map, front_viewdirs=None, back_viewdirs=None):
        """:119-124 -> (colors [N,3], color_map [S, 2S, 3])."""
        color_style = torch.rand_like(self.color_style) if self.random_style and self.training else self.color_style
        color_map = self.color_net([color_style], pose_map[None].contiguous(), randomize_noise=False, view_feature1=front_viewdirs,
                                   view_feature2=back_viewdirs)[0]
        return ops.gather_colors(color_map, self.core.pix), self._canvas(color_map)

    def transform_cano2live(self, gaussian_vals, items):
        """:84-91: skin ``positions`` and ``rotations`` of ``gaussian_vals`` (in place, as the reference) with the blended joint
        matrices of ``items['cano2live_jnt_mats']``."""
        gaussian_vals['positions'], gaussian_vals['rotations'] = ops.lbs_transform(
            gaussian_vals['positions'], gaussian_vals['rotations'], self.core.lbs, items['cano2live_jnt_mats'], self.core.lbs_sparse)
        return gaussian_vals

    def _fix_hand_pose_map(self):
        """The fixed frame's position map for ``generate_mean_hands()`` (the reference reads ``smpl_pos_map/%08d.exr %
        config.opt['test']['fix_hand_id']``, :61-67); the stand-alone class has no global config: pass it in, or use the drop-in."""
        raise RuntimeError("generate_mean_hands(): pass the fixed frame's pose map [3 or 6, S, S] (dropin/avatar_module.py reads "
                           "it from config.opt like the reference)")

    def _fix_hand_enabled(self):
        return bool(self.opt.get('fix_hand', False))

    @torch.no_grad()
    def generate_mean_hands(self, pose_map=None):
        """network/avatar.py:52-77: the Gaussians of one fixed frame (``opt['test']['fix_hand_id']``) that ``render`` fades the hands
        into at test time when ``fix_hand`` is set.  Also records ``hand_mask`` (Gaussians skinned mostly to wrist / finger
        joints)."""
        if pose_map is None:
            pose_map = self._fix_hand_pose_map()
        am = self.core.lbs.argmax(1)
        self.hand_mask = (am == 20) | (am == 21) | (am >= 25)
        position_map, other_map, color_map = self.get_maps(pose_map[:3])
        g = self.core.assemble(position_map, other_map, color_map)
        self.hand_positions, self.hand_opacity, self.hand_scales = g['positions'], g['opacity'], g['scales']
        self.hand_rotations, self.hand_colors = g['rotations'], g['colors']

    def render(self, items, bg_color=(0., 0., 0.), use_pca=False, use_vae=False):
        dev = self.core.xyz.device
        bg = _bg_tensor(bg_color, dev)
        assert not (use_pca and use_vae), "Cannot use both PCA and VAE!"
        key = 'smpl_pos_map_pca' if use_pca else 'smpl_pos_map_vae' if use_vae else 'smpl_pos_map'
        pose_map = items[key][:3]
        front_vd, back_vd = self.get_viewdir_feat(items) if self.with_viewdirs else (None, None)
        position_map, other_map, color_map = self.get_maps(pose_map, front_vd, back_vd)
        g = self.core.assemble(position_map, other_map, color_map)
        if (not self.training) and self._fix_hand_enabled():                                                       # :183-200
            if self.hand_positions is None:
                raise RuntimeError("fix_hand: call generate_mean_hands(pose_map) first (main_avatar.py:584)")
            g['positions'], g['opacity'], g['scales'], g['rotations'] = ops.hand_fuse(
                g['positions'], g['opacity'], g['scales'], g['rotations'], self.core.xyz, items['left_cano_mano_v'],
                items['right_cano_mano_v'], items['cano_smpl_center'], self.hand_positions, self.hand_opacity, self.hand_scales,
                self.hand_rotations)
        offset = g['positions'] - self.core.xyz                                                                   # :211
        g['positions'], g['rotations'] = ops.lbs_transform(g['positions'], g['rotations'], self.core.lbs,
                                                           items['cano2live_jnt_mats'], self.core.lbs_sparse)
        r = render3(g, bg, items['extr'], items['intr'], items['img_w'], items['img_h'])
        ret = {'rgb_map': r['render'].permute(1, 2, 0), 'mask_map': r['mask'].permute(1, 2, 0), 'offset': offset,
               'pos_map': self._canvas(position_map)}
        if not self.training:
            ret.update({'cano_tex_map': self._canvas(color_map), 'posed_gaussians': g})
        return ret

    def render_views(self, items, views, bg_color=(0., 0., 0.)):
        """Render V cameras of ONE pose (multi-view training step / free-view synthesis of a frame).

        ``items`` carries the pose (``smpl_pos_map``, ``cano2live_jnt_mats``); ``views`` is a list of dicts with ``extr``,
        ``intr``, ``img_w``, ``img_h``.  Everything that does not depend on the camera is evaluated once: the position and
        the other networks, the encoder and decoder stages 0..4 of the colour network (77 % of its FLOPs), the assembly
        of positions / opacities / scales / rotations and the LBS.  Per view: the view-direction features, stage 5 of the
        colour decoders, the colour gather and the rasterizer.  Each returned dict equals what ``render`` returns for
        ``{**items, **view}`` (in training mode up to the view-direction jitter's random draw); under autograd the
        shared part is back-propagated once with the gradients of all views summed."""
        dev = self.core.xyz.device
        bg = _bg_tensor(bg_color, dev)
        pose_map = items['smpl_pos_map'][:3]
        x = pose_map[None].contiguous()
        feats = [self.get_viewdir_feat({**items, **v}) if self.with_viewdirs else (None, None) for v in views]
        if self._graphs_active() and self._grouped_nets() is not None:        # one capture per view count: shared stages once, the tail per view
            nv = len(views)

            def views_fn(p, *fb):
                vf = {1: [(fb[2 * i], fb[2 * i + 1]) for i in range(nv)]}
                pm, cms, om = self._grouped_nets().forward([self.position_style, self.color_style, self.other_style], p, vf)
                return (pm, om, *cms)
            outs = self._graphed(("grouped_views", nv, self.with_viewdirs), views_fn, [x] + [t for fb in feats for t in fb])
            position_map, other_map, color_maps = outs[0], outs[1], [c.clone() for c in outs[2:]]
        elif self._graphs_active():
            (position_map,) = self._graphed("position", lambda p: self.position_net([self.position_style], p, randomize_noise=False)[0], [x])
            (other_map,) = self._graphed("other", lambda p: self.other_net([self.other_style], p, randomize_noise=False)[0], [x])
            cn = self.color_net

            def shared_fn(p):
                w_latent, noise = cn._latent_and_noise([self.color_style], False, None, False)
                levels = cn.encode(p)
                (o1, s1), (o2, s2) = (cn.decode_shared(b, levels, w_latent, noise) for b in (1, 2))
                return o1, s1, o2, s2, levels[0], w_latent

            o1, s1, o2, s2, level0, w_latent = self._graphed("color_shared", shared_fn, [x])

            def view_fn(f, b):        # closes over the static outputs of the shared capture
                noise = [cn._p(f"noises.noise_{i}") for i in range(cn.num_layers)]
                lv = [level0] * (len(cn.enc) + 1)     # the view-dependent stages only read the finest level, levels[0]
                parts = [cn.decode_view(br, lv, w_latent, noise, o, sk, vf) for br, o, sk, vf in ((1, o1, s1, f), (2, o2, s2, b))]
                return torch.cat(parts, 1)

            color_maps = []
            for f, b in feats:
                (cm,) = self._graphed(("color_view", f is not None), view_fn, [f, b])
                color_maps.append(cm.clone())     # the capture's output buffer is reused by the next view
        elif self._grouped_nets() is not None:
            color_style = torch.rand_like(self.color_style) if self.random_style and self.training else self.color_style
            vf = {1: [fb for fb in feats]} if self.with_viewdirs else {1: [(None, None)] * len(feats)}
            position_map, color_maps, other_map = self._grouped_nets().forward([self.position_style, color_style, self.other_style], x, vf,
                                                                               frozen=color_style is self.color_style and not self.training)
        else:
            color_style = torch.rand_like(self.color_style) if self.random_style and self.training else self.color_style
            position_map, other_map, color_maps = self._concurrently([
                lambda: self.position_net([self.position_style], x, randomize_noise=False)[0],
                lambda: self.other_net([self.other_style], x, randomize_noise=False)[0],
                lambda: self.color_net.forward_views([color_style], x, feats, randomize_noise=False)],
                shared=[x, color_style] + [t for fb in feats for t in fb])
        rets = []
        base = offset = None
        for v, color_map in zip(views, color_maps):
            if base is None:
                # camera-independent: positions / opacity / scales / rotations are gathered and skinned ONCE (round 5, second session: every view
                # used to gather all three maps again -- and back-propagate a zero-filled [1, 16 | 6, 1024, 1024] map gradient each that autograd
                # then added up: ~80 us per extra view); the views' gradients now meet on the [N, .] attribute tensors
                base = self.core.assemble(position_map, other_map, color_map)
                offset = base['positions'] - self.core.xyz
                base['positions'], base['rotations'] = ops.lbs_transform(base['positions'], base['rotations'], self.core.lbs,
                                                                        items['cano2live_jnt_mats'], self.core.lbs_sparse)
                g = base
            else:
                g = dict(base)
                g['colors'] = ops.gather_colors(color_map, self.core.pix)
            r = render3(g, bg, v['extr'], v['intr'], v['img_w'], v['img_h'])
            ret = {'rgb_map': r['render'].permute(1, 2, 0), 'mask_map': r['mask'].permute(1, 2, 0), 'offset': offset,
                   'pos_map': self._canvas(position_map)}
            if not self.training:
                ret.update({'cano_tex_map': self._canvas(color_map), 'posed_gaussians': g})
            rets.append(ret)
        return rets
