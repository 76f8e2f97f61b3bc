"""DualStyleUNet 512 -> 1024 (network/avatar.py:34) forward and forward+backward wall time on the MFMA path.
    python profiles/styleunet_bench.py [steps]"""
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import synth  # noqa: E402
from animatablegaussians_amd.styleunet import DualStyleUNet  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2)
net.load_reference_state_dict(synth.named_fill(net.reference_state_dict()))
net = net.to(dev)
pose = synth.pose_map(512).to(dev).requires_grad_(True)
style = (torch.ones(1, 512) / np.sqrt(512)).to(dev)
G = torch.randn(1, 6, 1024, 1024, device=dev)


def run(backward):
    if backward:
        net.zero_grad(set_to_none=True)       # as the training step does (no AccumulateGrad add kernels in the profile)
        pose.grad = None
        images, _ = net([style], pose, randomize_noise=False)
        (images * G).sum().backward()
    else:
        with torch.no_grad():
            net([style], pose, randomize_noise=False)


for backward in (False, True):
    for _ in range(3):          # allocator pools of the side streams and clocks settle over the first iterations
        run(backward)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run(backward)
    torch.cuda.synchronize()
    print(("fwd+bwd" if backward else "fwd    "), f"{(time.perf_counter() - t0) / steps * 1e3:.2f} ms")
