"""Is the 3-stream raster headline bound by the host?  The same loop (FusedRasterStep, views rotating over 3 streams) on the full scene and on
every 16th Gaussian of it (a sixteenth of the GPU work, identical host work).   python profiles/raster_host_bound_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import camera, synth
from animatablegaussians_amd.rasterizer import FusedRasterStep, GaussianRasterizationSettings
dev = torch.device("cuda:0")
W = H = 1024
av = synth.avatar_map_gaussians()
cams = synth.free_view_cameras(8, img=W)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
up = synth.upstream_grads(W, H, 12345)
g = [t(up[k]) for k in ("dL_dcolor", "dL_ddepth", "dL_dalpha")]
bg = t(av["bg"])
for sub in (1, 16, 256):
    arrs = [t(av[k][::sub]) for k in ("means3D", "colors", "opacities", "scales", "rotations")]
    P = arrs[0].shape[0]
    settings = []
    for c in cams:
        cm = camera.camera_from_intr_extr(c["extr"], c["intr"], W, H)
        settings.append(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cm["tanfovx"], tanfovy=cm["tanfovy"], bg=bg,
                                                      scale_modifier=1.0, viewmatrix=t(cm["viewmatrix"]), projmatrix=t(cm["projmatrix"]),
                                                      sh_degree=0, campos=t(cm["campos"]), prefiltered=False, debug=False))
    for ns in (1, 3):
        fused = FusedRasterStep(P, W, H, dev, n_streams=ns)
        def step(i):
            fused.view(settings[i % 8], *arrs, *g, slot=i % ns)
        for i in range(100): step(i)
        fused.join(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        N = 600
        for i in range(N): step(i)
        fused.join(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"P = {P:7d}  streams {ns}: {N / dt:8.0f} views/s  ({dt / N * 1e6:6.1f} us per view)")
