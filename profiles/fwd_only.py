"""Forward-only timing of the rasterizer stages (HIP events via ag_prof_*), all 8 views.  Debug aid."""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import helpers as h
from animatablegaussians_amd import synth, _lib
from animatablegaussians_amd.rasterizer import native_rasterize_gaussians
av = synth.avatar_map_gaussians()
empty = torch.Tensor([])
res = {}
for vi in range(8):
    camd = synth.free_view_cameras()[vi]
    scene = dict(av, **camd); cam = h.cam_of(scene)
    rs = h.gpu_settings(scene, cam); inp = h.gpu_inputs(scene)
    def fwd():
        return native_rasterize_gaussians(rs.bg, inp["means3D"], inp["colors"], inp["opacities"], inp["scales"], inp["rotations"], 1.0,
                                          empty, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, 1024, 1024, empty, 0, rs.campos, False, False)
    for _ in range(3): fwd()
    torch.cuda.synchronize()
    _lib.prof_enable(range(_lib.AG_K_COUNT))
    for _ in range(10): fwd()
    torch.cuda.synchronize()
    pr = _lib.prof_collect(); _lib.prof_enable([])
    res[vi] = {k: round(1e3 * ms / max(n, 1), 1) for k, (n, ms) in pr.items() if n}
for vi, r in res.items():
    print(vi, r)
