"""Host issue time against GPU time of one DualStyleUNet forward + backward and of the whole training step: is the Python thread or the
GPU the bottleneck?   python profiles/host_vs_gpu.py
host = wall time of the Python calls with the GPU queue drained before and NOT waited for after; total = the same + the final wait."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_avatar  # noqa: E402
from animatablegaussians_amd import synth  # noqa: E402
from animatablegaussians_amd.styleunet import DualStyleUNet  # noqa: E402

dev = torch.device("cuda:0")
net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2).to(dev)
pose = synth.pose_map(512).to(dev)
style = (torch.ones(1, 512) / np.sqrt(512)).to(dev)
G = torch.randn(1, 6, 1024, 1024, device=dev)


def net_pass(_i):
    net.zero_grad(set_to_none=True)
    images, _ = net([style], pose, randomize_noise=False)
    (images * G).sum().backward()


step = bench_avatar.TrainingStep(dev)
for name, fn in (("DualStyleUNet fwd+bwd", net_pass), ("training step, 1 view", lambda i: step(i, 1)), ("training step, 4 views", lambda i: step(i, 4))):
    for i in range(3):
        fn(i)
    hosts, totals = [], []
    for i in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hosts.append((t1 - t0) * 1e3)
        totals.append((t2 - t0) * 1e3)
    print(f"{name:24s} host issue {np.median(hosts):7.2f} ms   until the GPU is done {np.median(totals):7.2f} ms   (GPU tail after the last call {np.median(totals) - np.median(hosts):6.2f} ms)")
