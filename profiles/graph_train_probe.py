"""Can a DualStyleUNet forward + backward run from captured hipGraphs (torch.cuda.make_graphed_callables)?  Timing and equality probe."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import synth  # noqa: E402
from animatablegaussians_amd.styleunet import DualStyleUNet  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1)
net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2).to(dev)
pose = synth.pose_map(512).to(dev)
style = (torch.ones(1, 512) / np.sqrt(512)).to(dev)
G = torch.randn(1, 6, 1024, 1024, device=dev)


class Wrap(torch.nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, style, pose):
        return self.net([style], pose, randomize_noise=False)[0]


w = Wrap(net)


def run(mod):
    for p in net.parameters():
        p.grad = None
    out = mod(style, pose)
    (out * G).sum().backward()
    return out


def timeit(mod, n=10):
    for _ in range(3):
        run(mod)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run(mod)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out_e = run(w).detach().clone()
grads_e = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
print("eager  fwd+bwd %.2f ms" % timeit(w))
gw = torch.cuda.make_graphed_callables(w, (style, pose.clone().requires_grad_(False)))
out_g = run(gw).detach().clone()
grads_g = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
print("graphed fwd+bwd %.2f ms" % timeit(gw))
print("output equal:", torch.equal(out_e, out_g), float((out_e - out_g).abs().max()))
worst = max(float((grads_e[k] - grads_g[k]).abs().max() / (grads_e[k].abs().max() + 1e-30)) for k in grads_e)
print("grads: same keys", set(grads_e) == set(grads_g), "worst relative difference", worst)
