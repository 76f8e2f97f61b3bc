#!/usr/bin/env python
"""Golden fixture of the SMPL-X forward FROM THE REFERENCE'S OWN `smplx.SMPLX` CLASS (build container only).

Writes `synth.smplx_model_arrays()` (a synthetic model with the array names / shapes of SMPLX_NEUTRAL.npz -- the licensed
model files are not in this image) to a temporary .npz, constructs the reference class from /root/reference/smplx exactly as
dataset/dataset_mv_rgb.py:42 does, and evaluates the three calls of a data item (live, canonical, live without root:
dataset_mv_rgb.py:118-143) plus the `cano2live` products (:170-171) for two frames of `synth.smplx_pose_params`, in float32
(what the reference runs) and float64 (truth).  Only outputs are stored; model and parameters regenerate from the seed.

    python tests/golden/make_golden_smplx.py
"""
import math
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, "/root/reference")
import smplx  # noqa: E402  (reference code)

from animatablegaussians_amd import synth  # noqa: E402

assert smplx.__file__.startswith("/root/reference/"), smplx.__file__
arrays = synth.smplx_model_arrays()
params = synth.smplx_pose_params(n=2)
# config.py:9-15
cano_pose = np.zeros(75, np.float32)
cano_pose[3 + 3 * 1 + 2] = math.radians(25)
cano_pose[3 + 3 * 2 + 2] = math.radians(-25)
cano_pose = torch.from_numpy(cano_pose)
cano_transl, cano_go, cano_bp = cano_pose[:3], cano_pose[3:6], cano_pose[6:69]

out = {}
with tempfile.TemporaryDirectory() as d:
    np.savez(os.path.join(d, "SMPLX_NEUTRAL.npz"), **arrays)
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        model = smplx.SMPLX(model_path=d, gender='neutral', use_pca=False, num_pca_comps=45, flat_hand_mean=True, batch_size=1,
                            dtype=dt)
        p = {k: torch.from_numpy(v).to(dt) for k, v in params.items()}
        for pose_idx in range(2):
            with torch.no_grad():
                live = model.forward(betas=p['betas'][0][None], global_orient=p['global_orient'][pose_idx][None],
                                     transl=p['transl'][pose_idx][None], body_pose=p['body_pose'][pose_idx][None],
                                     jaw_pose=p['jaw_pose'][pose_idx][None], expression=p['expression'][pose_idx][None],
                                     left_hand_pose=p['left_hand_pose'][pose_idx][None],
                                     right_hand_pose=p['right_hand_pose'][pose_idx][None])
                cano = model.forward(betas=p['betas'][0][None], global_orient=cano_go.to(dt)[None], transl=cano_transl.to(dt)[None],
                                     body_pose=cano_bp.to(dt)[None], jaw_pose=p['jaw_pose'][pose_idx][None],
                                     expression=p['expression'][pose_idx][None])
                woroot = model.forward(betas=p['betas'][0][None], body_pose=p['body_pose'][pose_idx][None],
                                       jaw_pose=p['jaw_pose'][pose_idx][None], expression=p['expression'][pose_idx][None])
            item = {
                'live_smpl_v': live.vertices[0], 'cano_smpl_v': cano.vertices[0], 'live_smpl_v_woRoot': woroot.vertices[0],
                'joints': live.joints[0], 'cano_jnts': cano.joints[0], 'live_A': live.A[0], 'cano_A': cano.A[0],
                'cano2live_jnt_mats': torch.matmul(live.A[0], torch.linalg.inv(cano.A[0])),
                'cano2live_jnt_mats_woRoot': torch.matmul(woroot.A[0], torch.linalg.inv(cano.A[0])),
            }
            for k, v in item.items():
                out[f"{tag}_{pose_idx}_{k}"] = v.numpy().astype(np.float32 if tag == "f32" else np.float64)

# keep the fixture small: float64 truth only for the small tensors and a 1/16 vertex sample
keep = {}
for k, v in out.items():
    if k.startswith("f64") and v.shape[0] == 10475:
        keep[k] = v[::16].astype(np.float64)
    else:
        keep[k] = v
path = os.path.join(HERE, "smplx_body.npz")
np.savez_compressed(path, **keep)
print("wrote", path, os.path.getsize(path) >> 10, "KiB")
err = max(float(np.abs(out[f"f32_{i}_live_smpl_v"] - out[f"f64_{i}_live_smpl_v"]).max()) for i in range(2))
print("reference fp32 vs fp64, live vertices: max abs", err)
