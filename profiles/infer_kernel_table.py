"""Kernel table of one inference frame (AvatarNet.render in eval mode under no_grad, bench_avatar.TrainingStep.infer):  python profiles/infer_kernel_table.py"""
import collections
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_avatar  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

dev = torch.device("cuda:0")
step = bench_avatar.TrainingStep(dev)
step.net.eval()
for i in range(4):
    step.infer(i, 1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step.infer(4, 1)
    torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    for k in getattr(e, "kernels", None) or []:
        name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", k.name))[:80]
        acc[name][0] += 1
        acc[name][1] += float(k.duration)
print(f"inference frame: {sum(v[1] for v in acc.values()) / 1e3:.2f} ms of kernels in {sum(v[0] for v in acc.values())} launches")
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t:9.1f} us {n:4d}  {k}")
