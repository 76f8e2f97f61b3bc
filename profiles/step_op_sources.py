"""Which Python lines of the training step issue the small torch kernels (fills, copies, adds)?   python profiles/step_op_sources.py"""
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
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    step(3, 1)
    torch.cuda.synchronize()
WATCH = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::clone", "aten::zeros", "aten::zeros_like", "aten::cat", "aten::sum",
         "aten::mul", "aten::contiguous")
by = collections.defaultdict(collections.Counter)
cnt = collections.Counter()
for e in prof.events():
    if e.name in WATCH:
        cnt[e.name] += 1
        where = "?"
        for fr in (e.stack or []):
            if "/root/repo" in fr or "animatablegaussians_amd" in fr or "bench_avatar" in fr:
                where = fr.split("/")[-1][:70]
                break
        by[e.name][(where, str(e.input_shapes)[:60])] += 1
for k, v in cnt.most_common():
    print(k, v)
    for (w, s), c in by[k].most_common(10):
        print(f"      {c:4d}  {w:72s} {s}")
