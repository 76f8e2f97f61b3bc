"""Calibration of the end-to-end gradient bar (tests/test_raster_gpu.py level 4): GPU forward -> GPU blend backward on its own state
vs oracle forward -> oracle blend backward, against  lim = 1e-4*|ref| + 64*eps*abs_sum + K * D  where D = how far the reference
algorithm itself moves the element when every exp() is scaled by 1 + 2^-20 (oracle/raster_oracle.c: ago_set_exp_scale).
    python profiles/parity_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as h  # noqa: E402
from animatablegaussians_amd import synth  # noqa: E402
from oracle import raster_oracle as ro  # noqa: E402


def probe(name, scene, cam):
    ref = h.oracle_forward(scene, cam)
    ro.set_exp_scale(1.0 + 2.0 ** -20)
    try:
        ref_p = h.oracle_forward(scene, cam)
    finally:
        ro.set_exp_scale(1.0)
    frag = ref["fragile"].astype(bool) | ref_p["fragile"].astype(bool) | (ref["n_contrib"] != ref_p["n_contrib"])
    keep = (~frag).astype(np.float32)[None]
    grads = {k: np.ascontiguousarray(scene[k] * keep) for k in ("dL_dcolor", "dL_ddepth", "dL_dalpha")}
    acc = ro.backward_blend(ref, scene["colors"], scene["bg"], grads["dL_dcolor"], grads["dL_ddepth"], grads["dL_dalpha"])
    ro.set_exp_scale(1.0 + 2.0 ** -20)
    try:
        acc_p = ro.backward_blend(ref_p, scene["colors"], scene["bg"], grads["dL_dcolor"], grads["dL_ddepth"], grads["dL_dalpha"])
    finally:
        ro.set_exp_scale(1.0)
    fw = h.gpu_native_forward(scene, cam)
    got = h.gpu_native_backward(fw, grads)
    eps = float(np.finfo(np.float32).eps)
    print(f"{name}: masked pixels {frag.mean():.2e}")
    for nm, slots in h._SLOT_OF.items():
        for col, slot in enumerate(slots):
            if slot is None:
                continue
            g, r, rp = (np.asarray(a[nm], np.float64)[:, col] for a in (got, acc, acc_p))
            d = np.abs(g - r)
            D = np.abs(rp - r)
            base = 1e-4 * np.abs(r) + 64 * eps * acc["abs_sum"][:, slot].astype(np.float64) + 1e-7
            line = f"   {nm}[{col}] base-only worst {float((d / base).max()):8.1f} n>{int((d > base).sum()):6d};"
            for K in (1, 2, 4, 8, 16):
                lim = base + K * D
                line += f"  K={K}: worst {float((d / lim).max()):6.2f} n> {int((d > lim).sum())}"
            # how much of the tolerance is the sensitivity term where it matters
            line += f"  | median K*D/base at K=4: {float(np.median(4 * D / base)):.2e}, frac(4D>base) {float((4 * D > base).mean()):.2e}"
            print(line)


scene = synth.random_gaussians(P=10000, img=512)
probe("config1", scene, h.cam_of(scene))
av = synth.avatar_map_gaussians()
scene = dict(av, **synth.free_view_cameras()[1])
scene.update(synth.upstream_grads(1024, 1024, 11))
probe("config2", scene, h.cam_of(scene))
