"""Kernel durations of the 1x1 ToRGB / FromRGB convolutions (run under rocprofv3 --kernel-trace, summarise with summarize_rocprof.py
or the per-grid query below):  python profiles/pointwise_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import conv as agc  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(64, 12, 512), (128, 12, 256), (256, 12, 128), (512, 12, 64), (512, 12, 32), (512, 12, 16), (64, 32, 512), (512, 32, 16),
          (3, 128, 256), (3, 256, 128), (3, 512, 64), (3, 512, 16)]
for cin, cout, hw in shapes:
    x = torch.randn(1, cin, hw, hw, device=dev, requires_grad=True)
    w = torch.randn(cout, cin, 1, 1, device=dev, requires_grad=True)
    for _ in range(4):
        y = agc.conv2d(x, w)
        gx, gw = torch.autograd.grad(y, (x, w), torch.ones_like(y))
torch.cuda.synchronize()
print("done")
