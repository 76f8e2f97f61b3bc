"""Host time inside every custom autograd Function (forward and backward bodies, wall clock on whichever thread runs them) during training
steps, against the step's total host issue time.   python profiles/host_op_times.py"""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_avatar  # noqa: E402
import animatablegaussians_amd as pkg  # noqa: E402
from animatablegaussians_amd import avatar_ops, conv, rasterizer, styleunet_ops  # noqa: E402

acc = collections.defaultdict(lambda: [0, 0.0])


def wrap(cls, name):
    for meth in ("forward", "backward"):
        fn = getattr(cls, meth)

        def timed(*a, _fn=fn, _key=f"{name}.{meth}", **k):
            t0 = time.perf_counter()
            try:
                return _fn(*a, **k)
            finally:
                e = acc[_key]
                e[0] += 1
                e[1] += time.perf_counter() - t0
        setattr(cls, meth, staticmethod(timed))


for mod in (conv, styleunet_ops, avatar_ops, rasterizer):
    for k, v in vars(mod).items():
        if isinstance(v, type) and issubclass(v, torch.autograd.Function) and v is not torch.autograd.Function:
            wrap(v, f"{mod.__name__.split('.')[-1]}.{k}")

dev = torch.device("cuda:0")
step = bench_avatar.TrainingStep(dev)
for i in range(3):
    step(i, 1)
torch.cuda.synchronize()
acc.clear()
N = 5
t0 = time.perf_counter()
for i in range(N):
    step(i, 1)
host = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
print(f"per step: host issue {host / N * 1e3:.1f} ms, until done {total / N * 1e3:.1f} ms")
s = 0.0
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{t / N * 1e3:7.2f} ms  {n // N:5d} calls  {t / n * 1e6:6.1f} us each  {k}")
    s += t
print(f"{s / N * 1e3:7.2f} ms inside custom Function bodies per step")
