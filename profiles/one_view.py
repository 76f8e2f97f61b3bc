import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import numpy as np, torch
import helpers as h
from animatablegaussians_amd import synth
vi = int(sys.argv[1]) if len(sys.argv) > 1 else 2
av = synth.avatar_map_gaussians(); camd = synth.free_view_cameras()[vi]
scene = dict(av, **camd); scene.update(synth.upstream_grads(1024, 1024, 11))
cam = h.cam_of(scene)
for it in range(3):
    fw = h.gpu_native_forward(scene, cam)
    g = h.gpu_native_backward(fw, {k: scene[k] for k in ('dL_dcolor','dL_ddepth','dL_dalpha')})
torch.cuda.synchronize()
print('done view', vi, 'R', fw['num_rendered'])
