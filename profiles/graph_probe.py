"""Does capturing DualStyleUNet in a hipGraph pay?  Eager vs graph replay, forward (no grad) and forward+backward
(torch.cuda.make_graphed_callables)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import synth  # noqa: E402
from animatablegaussians_amd.styleunet import DualStyleUNet  # noqa: E402

dev = torch.device("cuda:0")
net = DualStyleUNet().to(dev)
pose = synth.pose_map(512).to(dev)
style = (torch.ones(1, 512) / np.sqrt(512)).to(dev)


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    eager = timeit(lambda: net([style], pose, randomize_noise=False))
    static_in = pose.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            net([style], static_in, randomize_noise=False)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_out, _ = net([style], static_in, randomize_noise=False)
    ref, _ = net([style], pose, randomize_noise=False)
    g.replay()
    torch.cuda.synchronize()
    print("graph output equals eager:", bool(torch.equal(ref, static_out)))
    graphed = timeit(lambda: g.replay())
print(f"forward: eager {eager:.2f} ms, hipGraph replay {graphed:.2f} ms")


class Wrap(torch.nn.Module):
    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, p):
        return self.inner([style], p, randomize_noise=False)[0]


G = torch.randn(1, 6, 1024, 1024, device=dev)
w = Wrap(net)
pin = pose.clone().requires_grad_(True)


def train_eager():
    (w(pin) * G).sum().backward()


e2 = timeit(train_eager, 5)
try:
    gw = torch.cuda.make_graphed_callables(w, (pose.clone().requires_grad_(True),))

    def train_graph():
        (gw(pin) * G).sum().backward()

    g2 = timeit(train_graph, 5)
    print(f"forward+backward: eager {e2:.2f} ms, graphed {g2:.2f} ms")
except Exception as ex:  # noqa: BLE001
    print(f"forward+backward: eager {e2:.2f} ms, graph capture failed: {type(ex).__name__}: {str(ex)[:300]}")
