"""SMPL-X body model forward on the MI355X kernels -- SURVEY.md §8(f)-2, the producer of ``cano2live_jnt_mats``.

The reference evaluates its vendored ``smplx.SMPLX`` three times per data item on the CPU inside the data loader (live pose,
canonical pose, live pose without root: ``dataset/dataset_mv_rgb.py:118-143``) and multiplies ``live.A`` with the inverse of
``cano.A`` (``:170-171``); ``AvatarNet.transform_cano2live`` then skins 250 k Gaussians with the result.  ``SMPLX`` below has
the constructor and ``forward`` signature of ``smplx/body_models.py:886-1290`` restricted to what the reference instantiates
(``use_pca=False``, no face contour, no joint mapper) and the same buffer names, reads the same ``SMPLX_{GENDER}.npz``, and
evaluates any batch of poses in ONE pass of ``ag_smplx_forward`` (include/ag_smplx.h): the 61-MB pose-corrective basis is read
once for the whole batch.  ``data_item`` is the three-call pattern of the dataset as one batch of three + ``ag_mat4_mul_inverse``.

Forward only (the reference calls it under ``torch.no_grad()``, dataset_mv_rgb.py:118); float32; there is no CPU path.
The vertex ids of the 21 extra joints are SMPL-X model-topology constants (smplx/vertex_ids.py:49-72).
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib

# nose, reye, leye, rear, lear | LBigToe, LSmallToe, LHeel, RBigToe, RSmallToe, RHeel | left finger tips | right finger tips
# (order of vertex_joint_selector.py:35-66 with use_hands and use_feet_keypoints)
_EXTRA_JOINT_VERTS = (9120, 9929, 9448, 616, 6, 5770, 5780, 8846, 8463, 8474, 8635,
                      5361, 4933, 5058, 5169, 5286, 8079, 7669, 7794, 7905, 8022)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


@dataclass
class SMPLXOutput:
    """Fields of smplx/utils.py:30-75 that SMPLX.forward fills."""
    vertices: Optional[torch.Tensor] = None
    joints: Optional[torch.Tensor] = None
    full_pose: Optional[torch.Tensor] = None
    global_orient: Optional[torch.Tensor] = None
    transl: Optional[torch.Tensor] = None
    v_shaped: Optional[torch.Tensor] = None
    betas: Optional[torch.Tensor] = None
    body_pose: Optional[torch.Tensor] = None
    left_hand_pose: Optional[torch.Tensor] = None
    right_hand_pose: Optional[torch.Tensor] = None
    expression: Optional[torch.Tensor] = None
    jaw_pose: Optional[torch.Tensor] = None
    A: Optional[torch.Tensor] = None

    def __getitem__(self, key):
        return getattr(self, key)

    def get(self, key, default=None):
        return getattr(self, key, default)

    def keys(self):
        return [k for k in self.__dataclass_fields__]

    def items(self):
        return [(k, getattr(self, k)) for k in self.keys()]


def mat4_mul_inverse(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``a @ inverse(b)`` for 4x4 matrices, ``b`` broadcast over the leading dimension of ``a`` when it has fewer matrices
    (dataset_mv_rgb.py:170-171)."""
    if a.shape[-2:] != (4, 4) or b.shape[-2:] != (4, 4) or not a.is_cuda or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise RuntimeError("mat4_mul_inverse: float32 GPU tensors [..., 4, 4]")
    a, b = a.contiguous(), b.contiguous()
    n, nb = a.numel() // 16, b.numel() // 16
    if nb == 0 or n % nb:
        raise RuntimeError(f"mat4_mul_inverse: {n} matrices against {nb}")
    out = torch.empty_like(a)
    with torch.cuda.device(a.device):
        _lib.check(_lib.lib().ag_mat4_mul_inverse(_p(out), _p(a), _p(b), n, nb, _stream(a.device)), "ag_mat4_mul_inverse")
    return out


class SMPLX(nn.Module):
    NUM_BODY_JOINTS = 21
    NUM_HAND_JOINTS = 15
    NUM_FACE_JOINTS = 3
    NUM_JOINTS = NUM_BODY_JOINTS + 2 * NUM_HAND_JOINTS + NUM_FACE_JOINTS
    SHAPE_SPACE_DIM = 300
    EXPRESSION_SPACE_DIM = 100

    def __init__(self, model_path, gender: str = 'neutral', use_pca: bool = True, num_pca_comps: int = 6, flat_hand_mean: bool = False,
                 batch_size: int = 1, num_betas: int = 10, num_expression_coeffs: int = 10, ext: str = 'npz', device='cuda',
                 use_face_contour: bool = False, dtype=torch.float32, **kwargs):
        """``model_path``: the directory holding ``SMPLX_{GENDER}.npz`` or the file itself (body_models.py:967-979), or a dict
        of its arrays.  use_pca / face contour are not built (the reference passes use_pca=False everywhere)."""
        super().__init__()
        if use_pca:
            raise NotImplementedError("SMPLX(use_pca=True): the reference constructs every model with use_pca=False")
        if use_face_contour:
            raise NotImplementedError("SMPLX(use_face_contour=True) is not used by the reference")
        if dtype != torch.float32:
            raise NotImplementedError("the device path is float32")
        if isinstance(model_path, dict):
            data = model_path
        else:
            path = os.path.join(model_path, f"SMPLX_{gender.upper()}.{ext}") if os.path.isdir(model_path) else model_path
            if not os.path.exists(path):
                raise FileNotFoundError(f"Path {path} does not exist!")
            if ext != 'npz' and not path.endswith('.npz'):
                raise ValueError("only the .npz model files are read here")
            data = np.load(path, allow_pickle=True)
        self.batch_size = batch_size
        self.gender = gender
        self.use_pca = False
        self.flat_hand_mean = flat_hand_mean
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=np.float32))

        sd = np.asarray(data['shapedirs'])
        if sd.ndim < 3:
            sd = sd[:, :, None]
        if sd.shape[-1] < self.SHAPE_SPACE_DIM + self.EXPRESSION_SPACE_DIM:        # body_models.py:1052-1062
            print(f'WARNING: You are using a {self.name()} model, with only 10 shape and 10 expression coefficients.')
            e0, e1 = 10, 20
            num_expression_coeffs = min(num_expression_coeffs, 10)
            num_betas = min(num_betas, 10)
        else:
            e0, e1 = self.SHAPE_SPACE_DIM, self.SHAPE_SPACE_DIM + num_expression_coeffs
            num_betas = min(num_betas, self.SHAPE_SPACE_DIM)
            num_expression_coeffs = min(num_expression_coeffs, self.EXPRESSION_SPACE_DIM)
        self._num_betas, self._num_expression_coeffs = num_betas, num_expression_coeffs
        V = sd.shape[0]
        pd = np.asarray(data['posedirs'])
        parents = np.asarray(data['kintree_table'])[0].astype(np.int64).copy()
        parents[0] = -1
        if np.any(parents[1:] >= np.arange(1, len(parents))) or np.any(parents[1:] < 0):
            raise ValueError("kintree_table: every joint's parent must precede it")
        self.faces = np.asarray(data['f'])
        reg = lambda n, t: self.register_buffer(n, t.to(device), persistent=not n.startswith('_'))
        reg('v_template', f32(data['v_template']))
        reg('shapedirs', f32(sd[:, :, :num_betas]))
        reg('expr_dirs', f32(sd[:, :, e0:e1]))
        reg('posedirs', f32(pd.reshape(-1, pd.shape[-1]).T))                       # [P, 3V]  (body_models.py:248-252)
        Jr = data['J_regressor']
        reg('J_regressor', f32(Jr.toarray() if hasattr(Jr, 'toarray') else Jr))
        reg('parents', torch.from_numpy(parents))
        reg('lbs_weights', f32(data['weights']))
        reg('faces_tensor', torch.from_numpy(self.faces.astype(np.int64)))
        reg('lmk_faces_idx', torch.from_numpy(np.asarray(data['lmk_faces_idx']).astype(np.int64)))
        reg('lmk_bary_coords', f32(data['lmk_bary_coords']))
        zeros45 = np.zeros(45, np.float32)
        reg('left_hand_mean', f32(zeros45 if flat_hand_mean else data['hands_meanl']))
        reg('right_hand_mean', f32(zeros45 if flat_hand_mean else data['hands_meanr']))
        reg('pose_mean', torch.cat([torch.zeros(3 + 3 * self.NUM_BODY_JOINTS + 9), self.left_hand_mean.cpu(), self.right_hand_mean.cpu()]))
        self._pose_mean_host = self.pose_mean.cpu().clone()
        # device-side forms the kernels read
        reg('_dirs', torch.cat([self.shapedirs, self.expr_dirs], -1).contiguous())     # [V, 3, NB]  (body_models.py:1233)
        reg('_parents32', self.parents.to(torch.int32))
        extra = np.asarray(_EXTRA_JOINT_VERTS, np.int64)
        if V <= extra.max():
            raise ValueError(f"model has {V} vertices; the SMPL-X extra-joint vertex ids need {extra.max() + 1}")
        tri = self.faces.astype(np.int64)[np.asarray(data['lmk_faces_idx']).astype(np.int64)]
        kp_idx = np.concatenate([np.repeat(extra[:, None], 3, 1), tri], 0)
        kp_w = np.concatenate([np.tile(np.array([[1., 0., 0.]], np.float32), (len(extra), 1)),
                               np.asarray(data['lmk_bary_coords'], np.float32)], 0)
        reg('_kp_idx', torch.from_numpy(kp_idx.astype(np.int32)))
        reg('_kp_w', f32(kp_w))
        self._desc = None

    def name(self) -> str:
        return 'SMPL-X'

    @property
    def num_betas(self):
        return self._num_betas

    @property
    def num_expression_coeffs(self):
        return self._num_expression_coeffs

    def _model(self):
        if self._desc is None or self._desc[0] != self.posedirs.data_ptr():
            m = _lib.AgSmplxModel()
            m.V, m.J, m.NB = self.v_template.shape[0], self.parents.shape[0], self._dirs.shape[-1]
            for name, t in (('v_template', self.v_template), ('shapedirs', self._dirs), ('posedirs', self.posedirs),
                            ('J_regressor', self.J_regressor), ('parents', self._parents32), ('lbs_weights', self.lbs_weights)):
                if not t.is_cuda or not t.is_contiguous():
                    raise _lib.AgNativeError(f"SMPLX.{name} must be a contiguous GPU tensor (there is no CPU path)")
                setattr(m, name, t.data_ptr())
            dev = self.v_template.device
            jt = torch.empty((m.J, 3), dtype=torch.float32, device=dev)
            jd = torch.empty((m.J, 3, max(m.NB, 1)), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):          # once per model: the joint regressor folded through the shape basis
                _lib.check(_lib.lib().ag_smplx_prepare(ctypes.byref(m), _p(jt), _p(jd), _stream(dev)), "ag_smplx_prepare")
            m.joint_template, m.joint_dirs = jt.data_ptr(), jd.data_ptr()
            self._desc = (self.posedirs.data_ptr(), m, jt, jd)
        return self._desc[1]

    def lbs(self, shape_components: torch.Tensor, full_pose: torch.Tensor, transl: Optional[torch.Tensor]):
        """(vertices [B,V,3], posed joints [B,J,3], A [B,J,4,4]) of smplx/lbs.py:152-246 (+ body_models.py:1272-1275)."""
        dev = self.v_template.device
        m = self._model()
        B = full_pose.shape[0]
        comps = shape_components.to(dev, torch.float32).contiguous()
        pose = full_pose.to(dev, torch.float32).reshape(B, m.J, 3).contiguous()
        tr = None if transl is None else transl.to(dev, torch.float32).reshape(B, 3).contiguous()
        if comps.shape != (B, m.NB):
            raise RuntimeError(f"shape components {tuple(comps.shape)} != ({B}, {m.NB})")
        verts = torch.empty((B, m.V, 3), dtype=torch.float32, device=dev)
        joints = torch.empty((B, m.J, 3), dtype=torch.float32, device=dev)
        A = torch.empty((B, m.J, 4, 4), dtype=torch.float32, device=dev)
        L = _lib.lib()
        nws = L.ag_smplx_workspace_floats(ctypes.byref(m), B)
        ws = torch.empty((max(nws, 1),), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.ag_smplx_forward(ctypes.byref(m), B, _p(comps), _p(pose), _p(tr), _p(verts), _p(joints), _p(A), _p(ws), nws,
                                          _stream(dev)), "ag_smplx_forward")
        return verts, joints, A

    def forward(self, betas=None, global_orient=None, body_pose=None, left_hand_pose=None, right_hand_pose=None, transl=None,
                expression=None, jaw_pose=None, leye_pose=None, reye_pose=None, return_verts: bool = True,
                return_full_pose: bool = False, pose2rot: bool = True, return_shaped: bool = False, **kwargs) -> SMPLXOutput:
        """body_models.py:1114-1290.  Arguments left None take the reference's defaults (zero parameters of the module).
        `return_shaped` defaults to False here (the reference's extra `v_shaped` output is unused by its callers)."""
        if not pose2rot:
            raise NotImplementedError("pose2rot=False (rotation-matrix input) is not used by the reference")
        dev = self.v_template.device
        given = [x for x in (betas, global_orient, body_pose) if x is not None]
        B = max([x.shape[0] for x in given], default=self.batch_size)
        z = lambda n: torch.zeros((B, n), dtype=torch.float32, device=dev)
        g = lambda x, n: z(n) if x is None else x.to(dev, torch.float32).reshape(-1, n)
        global_orient, body_pose = g(global_orient, 3), g(body_pose, 3 * self.NUM_BODY_JOINTS)
        jaw_pose, leye_pose, reye_pose = g(jaw_pose, 3), g(leye_pose, 3), g(reye_pose, 3)
        left_hand_pose, right_hand_pose = g(left_hand_pose, 45), g(right_hand_pose, 45)
        betas, expression = g(betas, self._num_betas), g(expression, self._num_expression_coeffs)
        full_pose = torch.cat([global_orient, body_pose, jaw_pose, leye_pose, reye_pose, left_hand_pose, right_hand_pose], 1)
        if full_pose.shape[0] != B:
            full_pose = full_pose.expand(B, -1)
        full_pose = full_pose + self.pose_mean
        if betas.shape[0] != B:                                                       # body_models.py:1218-1221
            betas = betas.expand(B, -1)
        if expression.shape[0] != B:
            expression = expression.expand(B, -1)
        comps = torch.cat([betas, expression], -1)
        if transl is not None and transl.shape[0] != B:
            transl = transl.expand(B, -1)
        verts, joints, A = self.lbs(comps, full_pose, transl)
        K = self._kp_idx.shape[0]
        extra = torch.empty((B, K, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ag_smplx_keypoints(_p(extra), _p(verts), _p(self._kp_idx), _p(self._kp_w), B, verts.shape[1], K,
                                                     _stream(dev)), "ag_smplx_keypoints")
        joints = torch.cat([joints, extra], 1)
        v_shaped = None
        if return_shaped:                                                              # body_models.py:1277-1279 (betas only)
            only_betas = torch.cat([betas, torch.zeros_like(expression)], -1).contiguous()
            v_shaped = torch.empty_like(verts)
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().ag_smplx_shape(ctypes.byref(self._model()), B, _p(only_betas), _p(v_shaped), _stream(dev)),
                           "ag_smplx_shape")
        return SMPLXOutput(vertices=verts if return_verts else None, joints=joints, betas=betas, expression=expression,
                           global_orient=global_orient, body_pose=body_pose, left_hand_pose=left_hand_pose,
                           right_hand_pose=right_hand_pose, jaw_pose=jaw_pose, v_shaped=v_shaped, transl=transl,
                           full_pose=full_pose if return_full_pose else None, A=A)

    def data_item(self, smpl_data, pose_idx: int, cano_global_orient, cano_transl, cano_body_pose) -> dict:
        """dataset/dataset_mv_rgb.py:118-143,155-171 for one frame: the live, canonical and live-without-root evaluations as ONE
        batch of three, then both `cano2live` matrix sets.  `smpl_data`: the tensors / arrays of smpl_params.npz (host memory,
        as the reference's dataset holds them): the three parameter rows are assembled on the host and uploaded in one copy."""
        dev = self.v_template.device
        h = lambda x: torch.as_tensor(x).detach().to('cpu', torch.float32).reshape(-1)
        row = lambda k: h(smpl_data[k][pose_idx])
        nb, ne = self._num_betas, self._num_expression_coeffs
        n_pose = 3 * (self.NUM_JOINTS + 1)
        o1, o2 = 3 * (nb + ne), 3 * (nb + ne + n_pose)
        buf = torch.zeros((o2 + 9,), dtype=torch.float32)             # three contiguous blocks: components | poses | transl
        comps, pose, tr = buf[:o1].view(3, nb + ne), buf[o1:o2].view(3, n_pose), buf[o2:].view(3, 3)
        comps[:, :nb] = h(smpl_data['betas'][0])[:nb]
        comps[:, nb:] = row('expression')[:ne]
        # full_pose = global_orient | body 63 | jaw | leye | reye | left hand 45 | right hand 45  (body_models.py:1203-1210)
        pose[0, 0:3], pose[1, 0:3] = row('global_orient'), h(cano_global_orient)
        pose[0, 3:66] = pose[2, 3:66] = row('body_pose')
        pose[1, 3:66] = h(cano_body_pose)
        pose[:, 66:69] = row('jaw_pose')
        pose[0, 75:120], pose[0, 120:165] = row('left_hand_pose'), row('right_hand_pose')
        pose += self._pose_mean_host
        tr[0], tr[1] = row('transl'), h(cano_transl)
        d = buf.to(dev, non_blocking=True)
        verts, joints, A = self.lbs(d[:o1].view(3, nb + ne), d[o1:o2].view(3, n_pose), d[o2:].view(3, 3))
        K = self._kp_idx.shape[0]
        extra = torch.empty((2, K, 3), dtype=torch.float32, device=dev)      # the dataset reads joints of live and canonical only
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ag_smplx_keypoints(_p(extra), _p(verts), _p(self._kp_idx), _p(self._kp_w), 2, verts.shape[1], K,
                                                     _stream(dev)), "ag_smplx_keypoints")
        c2l = mat4_mul_inverse(torch.stack([A[0], A[2]]), A[1])
        v = verts
        return {
            'joints': joints[0, :22], 'kin_parent': self.parents[:22].to(torch.long),
            'live_smpl_v': v[0], 'cano_smpl_v': v[1], 'live_smpl_v_woRoot': v[2], 'cano_jnts': torch.cat([joints[1], extra[1]], 0),
            'cano2live_jnt_mats': c2l[0], 'cano2live_jnt_mats_woRoot': c2l[1],
            'live_bounds': torch.stack([v[0].min(0)[0] - 0.15, v[0].max(0)[0] + 0.15], 0),
            'global_orient': d[o1:o1 + 3], 'transl': d[o2:o2 + 3],
        }
