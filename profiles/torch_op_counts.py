"""Counts of aten operators (and their input shapes) in one DualStyleUNet forward + backward: which small torch kernels are left?"""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import synth
from animatablegaussians_amd.styleunet import DualStyleUNet
dev = torch.device("cuda:0")
net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2).to(dev)
pose = synth.pose_map(512).to(dev); style = (torch.ones(1, 512) / np.sqrt(512)).to(dev); G = torch.randn(1, 6, 1024, 1024, device=dev)
def one():
    for p in net.parameters(): p.grad = None
    images, _ = net([style], pose, randomize_noise=False); (images * G).sum().backward()
one(); one()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    one()
cnt = collections.Counter(); shapes = collections.defaultdict(collections.Counter)
for e in prof.events():
    if e.name.startswith("aten::") and e.name in ("aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::cat", "aten::copy_", "aten::zeros", "aten::zero_", "aten::fill_", "aten::sum", "aten::clone", "aten::empty", "aten::contiguous", "aten::addmm", "aten::mm", "aten::flip", "aten::zeros_like", "aten::empty_like", "aten::neg", "aten::slice", "aten::narrow", "aten::split_with_sizes", "aten::reshape", "aten::view"):
        cnt[e.name] += 1
        shapes[e.name][str(e.input_shapes)[:90]] += 1
for k, v in cnt.most_common():
    print(k, v)
    if k in ("aten::add", "aten::add_", "aten::mul", "aten::copy_", "aten::zero_", "aten::fill_", "aten::sum", "aten::clone", "aten::flip", "aten::cat"):
        for s, c in shapes[k].most_common(8): print("      ", c, s)
