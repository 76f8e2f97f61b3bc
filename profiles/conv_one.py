"""One convolution shape, repeated -- target for rocprofv3 PMC passes.
    python profiles/conv_one.py Cin Cout H W [k stride pad reps mode]   (mode: fwd | dgrad | wgrad)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import conv as agc  # noqa: E402

a = sys.argv[1:]
cin, cout, h, w = (int(v) for v in a[:4])
k, s, p, reps = (int(v) for v in (a[4:8] + ["3", "1", "1", "10"][len(a[4:8]):]))
mode = a[8] if len(a) > 8 else "fwd"
dev = torch.device("cuda:0")
x = torch.randn(1, cin, h, w, device=dev, requires_grad=(mode == "dgrad"))
wt = torch.randn(cout, cin, k, k, device=dev, requires_grad=(mode == "wgrad"))
y = agc.conv2d(x, wt, stride=s, padding=p)
gy = torch.randn_like(y)
for _ in range(reps):
    if mode == "fwd":
        with torch.no_grad():
            agc.conv2d(x, wt, stride=s, padding=p)
    else:
        torch.autograd.grad(y, x if mode == "dgrad" else wt, gy, retain_graph=True)
torch.cuda.synchronize()
