"""Time one forward convolution shape with events: python profiles/conv_time.py Cin Cout H W [k s p reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import conv as agc  # noqa: E402

a = sys.argv[1:]
cin, cout, h, w = (int(v) for v in a[:4])
k, s, p, reps = (int(v) for v in (a[4:8] + ["3", "1", "1", "20"][len(a[4:8]):]))
dev = torch.device("cuda:0")
x = torch.randn(1, cin, h, w, device=dev)
wt = torch.randn(cout, cin, k, k, device=dev)
with torch.no_grad():
    y = agc.conv2d(x, wt, stride=s, padding=p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        agc.conv2d(x, wt, stride=s, padding=p)
    e1.record()
    torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
fl = 2.0 * cin * cout * k * k * y.shape[2] * y.shape[3]
print(f"AG_CONV_DBG={os.environ.get('AG_CONV_DBG', '0'):>3}  {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s (nominal)")
