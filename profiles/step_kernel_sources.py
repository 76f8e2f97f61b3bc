"""Which aten operators launch the small device kernels of the training step (fills, copies)?   python profiles/step_kernel_sources.py"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_avatar  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

dev = torch.device("cuda:0")
step = bench_avatar.TrainingStep(dev)
for i in range(3):
    step(i, 1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(3, 1)
    torch.cuda.synchronize()
by = collections.defaultdict(collections.Counter)
tot = collections.Counter()
for e in prof.events():
    for k in getattr(e, "kernels", []) or []:
        name = k.name
        key = "fill" if "FillFunctor" in name else "memcpy" if ("copyBuffer" in name or "Memcpy" in name) else "memset" if ("fillBuffer" in name or "Memset" in name) else None
        if key:
            tot[key] += 1
            by[key][(e.name, str(e.input_shapes)[:70])] += 1
for key, n in tot.items():
    print(key, n)
    for (op, shp), c in by[key].most_common(25):
        print(f"   {c:4d}  {op:40s} {shp}")
