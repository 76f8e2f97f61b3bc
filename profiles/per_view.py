"""Per-view kernel times of the raster path (HIP-event hooks of the library) and tile-list statistics: which views are
critical-path bound by a few very long tile lists."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as h  # noqa: E402
from animatablegaussians_amd import _lib, synth  # noqa: E402

# python profiles/per_view.py [S]   S = canvas height (1024: configs[1], 268 k Gaussians @1024^2; 2048: configs[4], 1.07 M @2048^2)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
IMG = S
av = synth.avatar_map_gaussians(S)
cams = synth.free_view_cameras(8, img=IMG, focal=1100.0 * S / 1024)
print("view      R  tiles  mean_len  max_len | pre  scan  scat  sort   fwd   bwd  prebwd (us)")
for vi, camd in enumerate(cams):
    scene = dict(av, **camd)
    scene.update(synth.upstream_grads(IMG, IMG, 11))
    cam = h.cam_of(scene)
    for _ in range(2):
        fw = h.gpu_native_forward(scene, cam)
        h.gpu_native_backward(fw, {k: scene[k] for k in ("dL_dcolor", "dL_ddepth", "dL_dalpha")})
    torch.cuda.synchronize()
    _lib.prof_enable(range(_lib.AG_K_COUNT))
    for _ in range(5):
        fw = h.gpu_native_forward(scene, cam)
        h.gpu_native_backward(fw, {k: scene[k] for k in ("dL_dcolor", "dL_ddepth", "dL_dalpha")})
    torch.cuda.synchronize()
    bd = _lib.prof_collect()
    _lib.prof_enable([])
    rng = fw["ranges"].astype(np.int64)
    ln = rng[:, 1] - rng[:, 0]
    ne = ln[ln > 0]
    us = {k: 1e3 * ms / max(n, 1) for k, (n, ms) in bd.items()}
    print(f"{vi:4d} {fw['num_rendered']:7d} {len(ne):6d} {ne.mean():9.0f} {ne.max():8d} | " + " ".join(
        f"{us.get(k, 0):5.0f}" for k in ("preprocess_kernel", "tile_scan_kernel", "scatter_kernel", "tile_sort_kernel",
                                         "blend_forward_kernel", "blend_backward_kernel", "preprocess_backward_kernel")))

# ---- static work assignment vs actual work: how unbalanced is the blend backward? ----------------------------------
# item = (tile rank by list length, region); workgroup b of G takes XCD x = b % 8, items of tile ranks x, x+8, ... dealt
# round-robin inside the XCD (ag_common.h ItemIter).  Actual work of an item ~ walk length = max n_contrib of its 32 pixels.
G = 512
for vi in ((0, 2) if S == 1024 else ()):
    scene = dict(av, **cams[vi])
    fw = h.gpu_native_forward(scene, h.cam_of(scene))
    rng = fw["ranges"].astype(np.int64)
    ln = rng[:, 1] - rng[:, 0]
    order = np.argsort(-ln, kind="stable")
    order = order[ln[order] > 0]
    nc = fw["n_contrib"].astype(np.int64)                       # [H, W]
    Hh, Ww = nc.shape
    gx = (Ww + 15) // 16
    work = np.zeros(G)
    items = 0
    for b in range(G):
        x, i, stride = b % 8, b // 8, (G + 7 - (b % 8)) // 8
        while True:
            rank = (i // 8) * 8 + x
            if rank >= len(order):
                break
            tile, reg = order[rank], i % 8
            ty, tx = tile // gx, tile % gx
            y0, x0 = ty * 16 + (reg >> 1) * 4, tx * 16 + (reg & 1) * 8
            w = nc[y0:y0 + 4, x0:x0 + 8].max() if y0 < Hh and x0 < Ww else 0
            work[b] += w
            items += 1
            i += stride
    print(f"view {vi}: {items} items over {G} workgroups; walk length per workgroup: mean {work.mean():.0f}, max {work.max():.0f}, "
          f"min {work.min():.0f} -> balance {work.mean() / work.max():.2f}; longest single item {ln.max()}")
