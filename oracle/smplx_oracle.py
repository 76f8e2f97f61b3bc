"""CPU oracle (torch, dtype-generic) of the SMPL-X forward the data item needs.  TEST INFRASTRUCTURE ONLY.

Restates smplx/body_models.py:1114-1290 (SMPLX.forward: parameter assembly, `pose_mean`, shape ++ expression, extra joints,
landmarks, `transl`) and smplx/lbs.py:152-246 (lbs), :299-330 (batch_rodrigues), :347-405 (batch_rigid_transform),
:108-149 (vertices2landmarks), smplx/vertex_joint_selector.py:25-76, and the three-call pattern of
dataset/dataset_mv_rgb.py:118-143,164-171 on plain torch ops.  Pinned by tests/golden/smplx_body.npz, which
tests/golden/make_golden_smplx.py produced by running the REFERENCE'S OWN `smplx.SMPLX` class (imported from
/root/reference) on the synthetic model file of `synth.smplx_model_arrays` -- the licensed SMPL-X model files are not in this
image, so the arithmetic is pinned, the real model's numbers are not.
"""
from __future__ import annotations

import numpy as np
import torch

# smplx/vertex_ids.py:49-72 ('smplx' table) in the order vertex_joint_selector.py:35-66 concatenates them:
# nose, reye, leye, rear, lear | LBigToe, LSmallToe, LHeel, RBigToe, RSmallToe, RHeel | l{thumb..pinky} | r{thumb..pinky}
EXTRA_JOINT_VERTS = (9120, 9929, 9448, 616, 6, 5770, 5780, 8846, 8463, 8474, 8635,
                     5361, 4933, 5058, 5169, 5286, 8079, 7669, 7794, 7905, 8022)


def model_tensors(arrays, dtype=torch.float32, num_betas=10, num_expression_coeffs=10, flat_hand_mean=True):
    """body_models.py:142-260 (SMPL), :604-660 (SMPLH hands), :995-1073 (SMPLX): what the constructor keeps."""
    t = lambda a: torch.as_tensor(np.asarray(a, np.float64)).to(dtype)
    sd = np.asarray(arrays['shapedirs'])
    if sd.shape[-1] < 300 + 100:                       # body_models.py:1052-1062
        e0, e1 = 10, 20
        num_expression_coeffs = min(num_expression_coeffs, 10)
    else:
        e0, e1 = 300, 300 + num_expression_coeffs
    pd = np.asarray(arrays['posedirs'])
    parents = np.asarray(arrays['kintree_table'])[0].astype(np.int64).copy()
    parents[0] = -1
    zeros45 = np.zeros(45)
    return {
        'v_template': t(arrays['v_template']),
        'shapedirs': t(sd[:, :, :num_betas]),
        'expr_dirs': t(sd[:, :, e0:e1]),
        'posedirs': t(pd.reshape(-1, pd.shape[-1]).T),                # [P, 3V]
        'J_regressor': t(arrays['J_regressor']),
        'parents': torch.as_tensor(parents),
        'lbs_weights': t(arrays['weights']),
        'faces': torch.as_tensor(np.asarray(arrays['f']).astype(np.int64)),
        'lmk_faces_idx': torch.as_tensor(np.asarray(arrays['lmk_faces_idx']).astype(np.int64)),
        'lmk_bary_coords': t(arrays['lmk_bary_coords']),
        'left_hand_mean': t(zeros45 if flat_hand_mean else arrays['hands_meanl']),
        'right_hand_mean': t(zeros45 if flat_hand_mean else arrays['hands_meanr']),
    }


def rodrigues(rv):
    """lbs.py:316-330: angle = |rv + 1e-8|, axis = rv / angle, R = I + sin K + (1 - cos) K K."""
    angle = torch.sqrt(((rv + 1e-8) ** 2).sum(-1, keepdim=True))
    ax = rv / angle
    K = torch.zeros(rv.shape[:-1] + (3, 3), dtype=rv.dtype)
    K[..., 0, 1], K[..., 0, 2] = -ax[..., 2], ax[..., 1]
    K[..., 1, 0], K[..., 1, 2] = ax[..., 2], -ax[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -ax[..., 1], ax[..., 0]
    s, c = torch.sin(angle)[..., None], torch.cos(angle)[..., None]
    return torch.eye(3, dtype=rv.dtype) + s * K + (1 - c) * (K @ K)


def rigid_chain(R, Jrest, parents):
    """lbs.py:373-405: per-joint [R | J - J_parent], accumulated down the tree; A removes the rest-pose joint."""
    B, Jn = R.shape[:2]
    G = torch.zeros(B, Jn, 4, 4, dtype=R.dtype)
    for j in range(Jn):
        M = torch.zeros(B, 4, 4, dtype=R.dtype)
        M[:, :3, :3] = R[:, j]
        M[:, 3, 3] = 1
        p = int(parents[j])
        M[:, :3, 3] = Jrest[:, j] - (Jrest[:, p] if p >= 0 else 0)
        G[:, j] = M if p < 0 else G[:, p] @ M
    A = G.clone()
    A[:, :, :3, 3] -= (G[:, :, :3, :3] @ Jrest[..., None])[..., 0]
    return G[:, :, :3, 3], A


def forward(m, betas, global_orient=None, body_pose=None, left_hand_pose=None, right_hand_pose=None, transl=None,
            expression=None, jaw_pose=None, leye_pose=None, reye_pose=None):
    """body_models.py:1185-1290 with use_pca = False; returns dict(vertices, joints [B,127,3], A, full_pose)."""
    dt = m['v_template'].dtype
    B = max(betas.shape[0], 1 if global_orient is None else global_orient.shape[0], 1 if body_pose is None else body_pose.shape[0])
    z = lambda n: torch.zeros(B, n, dtype=dt)
    g = lambda x, n: z(n) if x is None else x.to(dt).reshape(-1, n)
    full_pose = torch.cat([g(global_orient, 3), g(body_pose, 63), g(jaw_pose, 3), g(leye_pose, 3), g(reye_pose, 3),
                           g(left_hand_pose, 45), g(right_hand_pose, 45)], 1)
    pose_mean = torch.cat([torch.zeros(3 + 63 + 9, dtype=dt), m['left_hand_mean'], m['right_hand_mean']])
    full_pose = full_pose + pose_mean
    betas = betas.to(dt)
    if betas.shape[0] != B:
        betas = betas.expand(B, -1)
    comps = torch.cat([betas, g(expression, m['expr_dirs'].shape[-1])], 1)
    dirs = torch.cat([m['shapedirs'], m['expr_dirs']], -1)                       # [V, 3, NB]
    V = dirs.shape[0]
    v_shaped = m['v_template'] + (dirs.reshape(V * 3, -1) @ comps.T).T.reshape(B, V, 3)
    Jrest = torch.einsum('jv,bvc->bjc', m['J_regressor'], v_shaped)
    R = rodrigues(full_pose.reshape(B, -1, 3))
    feat = (R[:, 1:] - torch.eye(3, dtype=dt)).reshape(B, -1)
    v_posed = v_shaped + (feat @ m['posedirs']).reshape(B, V, 3)
    Jposed, A = rigid_chain(R, Jrest, m['parents'])
    T = (m['lbs_weights'] @ A.reshape(B, -1, 16)).reshape(B, V, 4, 4)
    verts = (T[..., :3, :3] @ v_posed[..., None])[..., 0] + T[..., :3, 3]
    tri = m['faces'][m['lmk_faces_idx']]                                          # [L, 3]
    lmk = (verts[:, tri] * m['lmk_bary_coords'][None, :, :, None]).sum(2)
    joints = torch.cat([Jposed, verts[:, list(EXTRA_JOINT_VERTS)], lmk], 1)
    if transl is not None:
        tr = transl.to(dt).reshape(-1, 1, 3)
        joints, verts = joints + tr, verts + tr
        A = A.clone()
        A[:, :, :3, 3] += tr
    return {'vertices': verts, 'joints': joints, 'A': A, 'full_pose': full_pose}


def data_item(m, p, pose_idx, cano_global_orient, cano_transl, cano_body_pose):
    """dataset/dataset_mv_rgb.py:118-143,164-171: live / canonical / live-without-root and the two cano2live matrix sets."""
    s = lambda k: torch.as_tensor(p[k][pose_idx][None])
    betas = torch.as_tensor(p['betas'][0][None])
    live = forward(m, betas, s('global_orient'), s('body_pose'), s('left_hand_pose'), s('right_hand_pose'), s('transl'),
                   s('expression'), s('jaw_pose'))
    cano = forward(m, betas, cano_global_orient[None], cano_body_pose[None], None, None, cano_transl[None], s('expression'), s('jaw_pose'))
    woroot = forward(m, betas, None, s('body_pose'), None, None, None, s('expression'), s('jaw_pose'))
    inv = torch.linalg.inv(cano['A'][0])
    return {'live_smpl_v': live['vertices'][0], 'cano_smpl_v': cano['vertices'][0], 'live_smpl_v_woRoot': woroot['vertices'][0],
            'joints': live['joints'][0], 'cano_jnts': cano['joints'][0],
            'cano2live_jnt_mats': live['A'][0] @ inv, 'cano2live_jnt_mats_woRoot': woroot['A'][0] @ inv}
