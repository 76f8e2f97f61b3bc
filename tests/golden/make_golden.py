#!/usr/bin/env python
"""Generate the golden fixtures of the rasterizer path FROM THE REFERENCE'S OWN CODE.

Runs the reference's cuda_rasterizer sources on the CPU (oracle/_ref, built by oracle/ref_build.py from
/root/reference -- only possible in the build container) on small seeded scenes and stores inputs, every
intermediate state and all outputs/gradients as compressed .npz files next to this script.  The fixtures are what
travels: tests check the oracle restatement (CPU) and the HIP path (GPU) against them without the reference present.

    OMP_NUM_THREADS=1 python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from animatablegaussians_amd import camera, synth  # noqa: E402
from oracle import ref_raster  # noqa: E402

SCENES = {
    # name: (P, W, H, focal, use_cov3d_precomp)
    "raster_random_p1500_128": (1500, 128, 128, 137.5, False),
    "raster_ragged_p800_100x60_cov3d": (800, 100, 60, 110.0, True),
}


def build(name):
    P, W, H, focal, precomp = SCENES[name]
    sc = synth.random_gaussians(P=P, seed=synth.SEED + len(name), img=max(W, H), focal=focal)
    sc["img_w"], sc["img_h"] = W, H
    sc["intr"] = np.array([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1]], np.float32)
    sc.update(synth.upstream_grads(W, H, 777 + P))
    cam = camera.camera_from_intr_extr(sc["extr"], sc["intr"], W, H)
    r = ref_raster.RefRasterizer()
    cov = None
    if precomp:   # take the reference's own cov3D of the scale/rotation path as the precomputed input
        st0 = r.forward(sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"], sc["bg"],
                        cam["viewmatrix"], cam["projmatrix"], cam["campos"], cam["tanfovx"], cam["tanfovy"], W, H)
        cov = st0["cov3D"].copy()
        r = ref_raster.RefRasterizer()
    st = r.forward(sc["means3D"], sc["colors"], sc["opacities"], None if precomp else sc["scales"],
                   None if precomp else sc["rotations"], sc["bg"], cam["viewmatrix"], cam["projmatrix"], cam["campos"],
                   cam["tanfovx"], cam["tanfovy"], W, H, cov3D_precomp=cov)
    g = r.backward(sc["dL_dcolor"], sc["dL_ddepth"], sc["dL_dalpha"])
    out = {"in_" + k: sc[k] for k in ("means3D", "scales", "rotations", "opacities", "colors", "bg", "extr", "intr",
                                      "dL_dcolor", "dL_ddepth", "dL_dalpha")}
    out["in_img_wh"] = np.array([W, H], np.int32)
    if precomp:
        out["in_cov3D_precomp"] = cov
    for k in ("viewmatrix", "projmatrix", "campos"):
        out["cam_" + k] = cam[k]
    out["cam_tanfov"] = np.array([cam["tanfovx"], cam["tanfovy"]], np.float64)
    vis = st["radii"] > 0
    for k, v in st.items():
        if k == "num_rendered":
            out["st_num_rendered"] = np.array([v], np.int64)
        elif k in ("depths", "means2D", "cov3D", "conic_opacity"):
            out["st_" + k] = np.where(vis.reshape((-1,) + (1,) * (v.ndim - 1)), v, 0).astype(v.dtype)
        else:
            out["st_" + k] = v
    for k, v in g.items():
        out["g_" + k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: R={st['num_rendered']} visible={int(vis.sum())} -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    assert ref_raster.available(), "build oracle/_ref first: python oracle/ref_build.py"
    for n in SCENES:
        build(n)
