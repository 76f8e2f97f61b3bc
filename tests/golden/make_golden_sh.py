#!/usr/bin/env python
"""Golden fixture of the spherical-harmonics colour path FROM THE REFERENCE'S OWN CODE (oracle/_ref, build container
only): one small scene per active degree 0..3 with M = 16 coefficients per Gaussian, colours that go negative for a
share of the Gaussians (so the clamp and its zeroed gradient are exercised), forward state + every gradient.

    OMP_NUM_THREADS=1 python tests/golden/make_golden_sh.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from animatablegaussians_amd import camera, synth  # noqa: E402
from oracle import ref_raster  # noqa: E402

P, W, H, FOCAL, M = 600, 96, 80, 100.0, 16
sc = synth.random_gaussians(P=P, seed=synth.SEED + 99, img=max(W, H), focal=FOCAL)
sc["intr"] = np.array([[FOCAL, 0, W / 2], [0, FOCAL, H / 2], [0, 0, 1]], np.float32)
sc.update(synth.upstream_grads(W, H, 4321))
rng = np.random.default_rng(2024)
shs = (rng.normal(0, 0.6, (P, M, 3))).astype(np.float32)
shs[:, 0, :] = rng.normal(-0.8, 1.5, (P, 3)).astype(np.float32)
cam = camera.camera_from_intr_extr(sc["extr"], sc["intr"], W, H)
out = {"in_" + k: sc[k] for k in ("means3D", "scales", "rotations", "opacities", "bg", "extr", "intr", "dL_dcolor",
                                  "dL_ddepth", "dL_dalpha")}
out["in_shs"] = shs
out["in_img_wh"] = np.array([W, H], np.int32)
for k in ("viewmatrix", "projmatrix", "campos"):
    out["cam_" + k] = cam[k]
out["cam_tanfov"] = np.array([cam["tanfovx"], cam["tanfovy"]], np.float64)
for deg in range(4):
    r = ref_raster.RefRasterizer()
    st = r.forward(sc["means3D"], None, sc["opacities"], sc["scales"], sc["rotations"], sc["bg"], cam["viewmatrix"],
                   cam["projmatrix"], cam["campos"], cam["tanfovx"], cam["tanfovy"], W, H, shs=shs, sh_degree=deg)
    g = r.backward(sc["dL_dcolor"], sc["dL_ddepth"], sc["dL_dalpha"])
    vis = st["radii"] > 0
    assert st["clamped"][vis].any() and not st["clamped"][vis].all()
    for k in ("radii", "color", "depth", "alpha", "n_contrib", "point_list"):
        out[f"d{deg}_st_{k}"] = st[k]
    out[f"d{deg}_st_num_rendered"] = np.array([st["num_rendered"]], np.int64)
    out[f"d{deg}_st_rgb"] = np.where(vis[:, None], st["rgb"], 0).astype(np.float32)      # rgb of culled Gaussians is never written
    out[f"d{deg}_st_clamped"] = np.where(vis[:, None], st["clamped"], 0).astype(np.uint8)
    for k, v in g.items():
        out[f"d{deg}_g_{k}"] = v
np.savez_compressed(os.path.join(HERE, "raster_sh_p600_96x80.npz"), **out)
print("wrote raster_sh_p600_96x80.npz:", len(out), "arrays; R =", [int(out[f"d{d}_st_num_rendered"][0]) for d in range(4)])
