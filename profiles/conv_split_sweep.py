"""Split-K sweep of the convolution kernels on the layer shapes whose grids do not fill the chip:  python profiles/conv_split_sweep.py
Forces the split count through AG_CONV_SPLITS / AG_WGRAD_SPLITS (read per call by the library) and times forward, input gradient and
weight gradient with HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import conv as agc  # noqa: E402

dev = torch.device("cuda:0")
orig = agc._Conv.apply


def timeit(fn, reps=20):
    fn()
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


shapes_all = [(0, 1024, 512, 64, 3, 1, 1), (0, 512, 512, 64, 3, 1, 1), (0, 512, 256, 128, 3, 1, 1), (0, 256, 256, 128, 3, 1, 1),
          (0, 1024, 512, 32, 3, 1, 1), (0, 512, 512, 32, 3, 1, 1), (1, 512, 512, 32, 3, 2, 0), (0, 512, 512, 16, 3, 1, 1),
          (0, 128, 128, 256, 3, 1, 1), (0, 256, 128, 256, 3, 1, 1)]
shapes = shapes_all if len(sys.argv) < 2 else [shapes_all[int(i)] for i in sys.argv[1].split(",")]
for kind, cin, cout, hw, k, s, p in shapes:
    x = torch.randn(1, cin, hw, hw, device=dev)
    wt = torch.randn((cout, cin, k, k) if kind == 0 else (cin, cout, k, k), device=dev)
    xi = x.clone().requires_grad_(True)
    wi = wt.clone().requires_grad_(True)
    print(f"{'conv' if kind == 0 else 'convT'} {cin}->{cout} @{hw}")
    for var, label in (("AG_CONV_SPLITS", "fwd/dgrad"), ("AG_WGRAD_SPLITS", "wgrad")):
        row = []
        sweep = (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 0) if var == "AG_CONV_SPLITS" else (3, 4, 6, 7, 8, 14, 16, 21, 28, 32, 56, 64, 96, 128, 0)
        for sp in sweep:
            if sp:
                os.environ[var] = str(sp)
            else:
                os.environ.pop(var, None)
            if var == "AG_CONV_SPLITS":
                tf = timeit(lambda: orig(x, wt, None, None, kind, s, p, 1.0))
                yi = orig(xi, wt, None, None, kind, s, p, 1.0)
                gy = torch.ones_like(yi)
                td = timeit(lambda: torch.autograd.grad(yi, xi, gy, retain_graph=True))
                row.append(f"{sp or 'auto'}:{tf:.0f}/{td:.0f}")
            else:
                yw = orig(x, wi, None, None, kind, s, p, 1.0)
                gy = torch.ones_like(yw)
                tw = timeit(lambda: torch.autograd.grad(yw, wi, gy, retain_graph=True))
                row.append(f"{sp or 'auto'}:{tw:.0f}")
            os.environ.pop(var, None)
        print(f"   {label:9s} " + "  ".join(row))
