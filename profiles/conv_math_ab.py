"""Interleaved A/B of the convolution arithmetic modes on (a) one DualStyleUNet forward + backward and (b) the whole training step, in ONE
process: blocks of steps alternate between the modes so that clock / temperature history is shared.   python profiles/conv_math_ab.py [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_avatar  # noqa: E402
from animatablegaussians_amd import conv as agc, synth  # noqa: E402
from animatablegaussians_amd.styleunet import DualStyleUNet  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
modes = ("fp32", "split_bf16", "split_bf16x3")
net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2).to(dev)
pose = synth.pose_map(512).to(dev)
style = (torch.ones(1, 512) / np.sqrt(512)).to(dev)
G = torch.randn(1, 6, 1024, 1024, device=dev)


def net_pass(_i):
    net.zero_grad(set_to_none=True)
    images, _ = net([style], pose, randomize_noise=False)
    (images * G).sum().backward()


step = bench_avatar.TrainingStep(dev)
legs = (("DualStyleUNet fwd+bwd", net_pass, 8), ("training step, 1 view", lambda i: step(i, 1), 6), ("training step, 4 views", lambda i: step(i, 4), 4))
res = {(leg[0], m): [] for leg in legs for m in modes}
for rep in range(reps):
    for name, fn, n in legs:
        for m in modes:
            agc.set_math(m)
            res[(name, m)].append(bench_avatar.timed(fn, n, 2, dev))
for name, _, _ in legs:
    print(name + ": " + "   ".join(f"{m} {np.median(res[(name, m)]):7.2f} ms (min {min(res[(name, m)]):6.2f})" for m in modes))
