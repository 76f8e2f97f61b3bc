"""Worst blend-backward accumulator elements of BASELINE configs[4] (1.07 M Gaussians @2048^2) against the fp64 oracle, per slot, with the
error percentiles: run once per kernel variant (AG_BWD_KERNEL=0|1) to compare their error statistics (r03: identical)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as h
from animatablegaussians_amd import synth
from oracle import raster_oracle as ro
S = W = 2048
av = synth.avatar_map_gaussians(S)
scene = dict(av, **synth.free_view_cameras(8, img=W, focal=2200.0)[3]); scene.update(synth.upstream_grads(W, W, 17))
cam = h.cam_of(scene)
ref = h.oracle_forward(scene, cam)
keep = (~ref["fragile"].astype(bool)).astype(np.float32)[None]
grads = {k: np.ascontiguousarray(scene[k] * keep) for k in ("dL_dcolor", "dL_ddepth", "dL_dalpha")}
acc_ref = ro.backward_blend(ref, scene["colors"], scene["bg"], grads["dL_dcolor"], grads["dL_ddepth"], grads["dL_dalpha"])
gpu = h.gpu_native_forward(scene, cam)
got = h.gpu_native_backward(gpu, grads, alphas=ref["alpha"])
eps = float(np.finfo(np.float32).eps)
out = {}
for name, slots in h._SLOT_OF.items():
    for col, slot in enumerate(slots):
        if slot is None: continue
        g = np.asarray(got[name], np.float64)[:, col]; r = np.asarray(acc_ref[name], np.float64)[:, col]
        lim = 1e-4 * np.abs(r) + 128 * eps * acc_ref["abs_sum"][:, slot].astype(np.float64) + 1e-7
        ratio = np.abs(g - r) / lim
        i = int(np.argmax(ratio))
        q = np.percentile(ratio, [50, 99, 99.99])
        print(f"{name}[{col}]: worst ratio {ratio[i]:.3f} at {i}: got {g[i]:.6e} ref {r[i]:.6e} abs_sum {acc_ref['abs_sum'][i, slot]:.4e}; ratio pct 50/99/99.99 {q[0]:.4f} {q[1]:.4f} {q[2]:.4f}; mean|d|/(eps*abs_sum) {np.mean(np.abs(g-r)/(eps*acc_ref['abs_sum'][:,slot]+1e-30)):.3f}")
        if name == "dL_dmeans2D" and col == 0:
            np.save(os.path.join(ROOT, "gpurun_out", f"m2x_{os.environ.get('AG_BWD_KERNEL','0')}.npy"), g.astype(np.float32))
            print("   tiles_touched", ref["tiles_touched"][i], "radius", ref["radii"][i], "opacity", ref["conic_opacity"][i], "means2D", ref["means2D"][i])
