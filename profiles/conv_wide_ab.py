"""A/B of the 128 x 256 gather tile (AG_CONV_WIDE=1: 4 accumulators per wave, one workgroup per CU) against the 128 x 128 tile
(two workgroups per CU) on the large M = 128 .. 512 layers:  python profiles/conv_wide_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import conv as agc  # noqa: E402

dev = torch.device("cuda:0")
orig = agc._Conv.apply


def timeit(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


shapes = [(256, 128, 256), (128, 128, 256), (512, 256, 128), (256, 256, 128), (1024, 512, 64), (512, 512, 64), (1024, 512, 32)]
for cin, cout, hw in shapes:
    x = torch.randn(1, cin, hw, hw, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev)
    xi = x.clone().requires_grad_(True)
    row = []
    for wide in ("0", "1", "0", "1"):
        os.environ["AG_CONV_WIDE"] = wide
        tf = timeit(lambda: orig(x, w, None, None, 0, 1, 1, 1.0))
        yi = orig(xi, w, None, None, 0, 1, 1, 1.0)
        gy = torch.ones_like(yi)
        td = timeit(lambda: torch.autograd.grad(yi, xi, gy, retain_graph=True))
        gf = 2.0 * cin * cout * 9 * hw * hw / 1e9
        row.append(f"wide={wide}: fwd {tf:.0f} us ({gf / tf * 1e3:.0f} TF) dgrad {td:.0f} us ({gf / td * 1e3:.0f} TF)")
    print(f"{cin}->{cout} @{hw}:  " + " | ".join(row))
os.environ.pop("AG_CONV_WIDE", None)
