"""One warm-up + N timed DualStyleUNet forwards (no grad) -- for rocprofv3 kernel-time vs wall-time accounting."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import synth  # noqa: E402
from animatablegaussians_amd.styleunet import DualStyleUNet  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
mode = sys.argv[2] if len(sys.argv) > 2 else "fwd"
dev = torch.device("cuda:0")
net = DualStyleUNet().to(dev)
pose = synth.pose_map(512).to(dev).requires_grad_(mode != "fwd")
style = (torch.ones(1, 512) / np.sqrt(512)).to(dev)
G = torch.randn(1, 6, 1024, 1024, device=dev)


def run():
    if mode == "fwd":
        with torch.no_grad():
            net([style], pose, randomize_noise=False)
    else:
        images, _ = net([style], pose, randomize_noise=False)
        (images * G).sum().backward()


run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    run()
torch.cuda.synchronize()
print(f"{mode}: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/iter over {steps} iters")
