"""Where the host time of one raster step goes (single stream): C-ABI calls (HIP launches + the wait for the instance count)
vs Python / torch around them.  Wraps the three library entry points with perf_counter."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from animatablegaussians_amd import _lib, camera, synth  # noqa: E402
from animatablegaussians_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402

dev = torch.device("cuda", 0)
L = _lib.lib()
acc = {}


class Timed:
    def __init__(self, fn, name):
        self.fn, self.name = fn, name

    def __call__(self, *a):
        t = time.perf_counter_ns()
        r = self.fn(*a)
        acc[self.name] = acc.get(self.name, 0) + time.perf_counter_ns() - t
        return r


for n in ("ag_raster_forward_optimistic", "ag_raster_backward", "ag_raster_forward_plan", "ag_raster_forward_render"):
    setattr(L, n, Timed(getattr(L, n), n))

W = H = 1024
av = synth.avatar_map_gaussians()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
leaves = [t(av[k]).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "colors")]
means3D, scales, rotations, opacities, colors = leaves
bg = t(av["bg"])
rasts = []
for c in synth.free_view_cameras(8, img=W):
    cm = camera.camera_from_intr_extr(c["extr"], c["intr"], W, H)
    rasts.append(GaussianRasterizer(GaussianRasterizationSettings(H, W, cm["tanfovx"], cm["tanfovy"], bg, 1.0, t(cm["viewmatrix"]),
                                                                  t(cm["projmatrix"]), 0, t(cm["campos"]), False, False)))
up = synth.upstream_grads(W, H, 1)
g = (t(up["dL_dcolor"]), t(up["dL_ddepth"]), t(up["dL_dalpha"]))
from torch.autograd.graph import _engine_run_backward  # noqa: E402
seg = {"zeros": 0, "forward": 0, "backward": 0, "reset": 0}
N = 1000
if len(sys.argv) > 1 and sys.argv[1] == "single":
    torch.autograd.set_multithreading_enabled(False).__enter__()       # backward nodes run on the calling thread
for it in range(N + 100):
    if it == 100:
        torch.cuda.synchronize()
        acc.clear()
        for k in seg:
            seg[k] = 0
        t_all = time.perf_counter_ns()
    t0 = time.perf_counter_ns()
    m2 = torch.zeros_like(means3D, requires_grad=True)
    t1 = time.perf_counter_ns()
    color, radii, depth, alpha = rasts[it % 8](means3D=means3D, means2D=m2, opacities=opacities, shs=None, colors_precomp=colors,
                                               scales=scales, rotations=rotations, cov3D_precomp=None)
    t2 = time.perf_counter_ns()
    _engine_run_backward((color, depth, alpha), g, False, False, (), allow_unreachable=True, accumulate_grad=True)
    t3 = time.perf_counter_ns()
    for leaf in leaves:
        leaf.grad = None
    t4 = time.perf_counter_ns()
    seg["zeros"] += t1 - t0; seg["forward"] += t2 - t1; seg["backward"] += t3 - t2; seg["reset"] += t4 - t3
torch.cuda.synchronize()
total = (time.perf_counter_ns() - t_all) / N / 1e3
print(f"step {total:.1f} us  |  " + "  ".join(f"{k} {v / N / 1e3:.1f}" for k, v in seg.items()))
print("inside the C ABI: " + "  ".join(f"{k} {v / N / 1e3:.1f}" for k, v in acc.items()))
