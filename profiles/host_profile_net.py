"""cProfile of the host side of DualStyleUNet forward + backward (the backward's Python runs on autograd's thread and is profiled through
its own profiler enabled inside the first backward call).   python profiles/host_profile_net.py"""
import cProfile
import os
import pstats
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import synth  # noqa: E402
from animatablegaussians_amd.styleunet import DualStyleUNet  # noqa: E402

dev = torch.device("cuda:0")
os.environ["AG_SINGLE_STREAM"] = "1"
net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2).to(dev)
pose = synth.pose_map(512).to(dev)
style = (torch.ones(1, 512) / np.sqrt(512)).to(dev)
G = torch.randn(1, 6, 1024, 1024, device=dev)


def fwd():
    net.zero_grad(set_to_none=True)
    images, _ = net([style], pose, randomize_noise=False)
    return (images * G).sum()


for _ in range(3):
    fwd().backward()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
losses = [fwd() for _ in range(5)]
pr.disable()
print("==== forward (5 passes)")
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
threading.setprofile(None)
pr2 = cProfile.Profile()
# autograd executes custom backward functions on its own thread: profile that thread too
threading.setprofile(lambda *a: None)
pr2.enable()
for loss in losses:
    loss.backward()
pr2.disable()
print("==== backward (5 passes, calling thread only; custom Function.backward runs on the autograd thread for CUDA graphs)")
pstats.Stats(pr2).sort_stats("tottime").print_stats(12)
