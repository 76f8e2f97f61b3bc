"""Training step with the three StyleUNets replayed from hipGraphs (torch.cuda.make_graphed_callables per network, the two decoder branches
forked inside each capture, the three replays on the three network streams) against the eager step.  Probe: timing + loss equality."""
import os
import sys
import time

import numpy as np
import torch

os.environ["AG_CAPTURE_BRANCHES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_avatar  # noqa: E402

dev = torch.device("cuda:0")
step = bench_avatar.TrainingStep(dev)
net = step.net


class Wrap(torch.nn.Module):
    def __init__(self, unet, with_view):
        super().__init__()
        self.unet, self.with_view = unet, with_view

    def forward(self, style, pose, *views):
        if self.with_view:
            return self.unet([style], pose, randomize_noise=False, view_feature1=views[0], view_feature2=views[1])[0]
        return self.unet([style], pose, randomize_noise=False)[0]


def timeit(n=8):
    for i in range(3):
        step(i, 1)
    torch.cuda.synchronize()
    hosts = []
    t0 = time.perf_counter()
    for i in range(n):
        h0 = time.perf_counter()
        step(i, 1)
        hosts.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, float(np.median(hosts)) * 1e3


print("eager   step %.2f ms (host issue %.2f ms)" % timeit())
items = dict(step.views[0])
net.get_pose_map(items)
x = items['smpl_pos_map'][:3][None].contiguous()
fv, bv = net.get_viewdir_feat(items)
g_pos = torch.cuda.make_graphed_callables(Wrap(net.position_net, False), (net.position_style.detach().clone().requires_grad_(True), x.clone()))
g_oth = torch.cuda.make_graphed_callables(Wrap(net.other_net, False), (net.other_style.detach().clone().requires_grad_(True), x.clone()))
g_col = torch.cuda.make_graphed_callables(Wrap(net.color_net, True), (net.color_style.detach().clone().requires_grad_(True), x.clone(),
                                                                      fv.detach().clone().requires_grad_(True), bv.detach().clone().requires_grad_(True)))
orig = net.get_maps


def graphed_maps(pose_map, front_viewdirs=None, back_viewdirs=None):
    xx = pose_map[None].contiguous()
    return tuple(net._concurrently([lambda: g_pos(net.position_style, xx), lambda: g_oth(net.other_style, xx),
                                    lambda: g_col(net.color_style, xx, front_viewdirs, back_viewdirs)],
                                   shared=[xx, front_viewdirs, back_viewdirs]))


net.get_maps = graphed_maps
print("graphed step %.2f ms (host issue %.2f ms)" % timeit())
print("memory allocated %.1f GB" % (torch.cuda.memory_allocated() / 2 ** 30))
