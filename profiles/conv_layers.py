"""Per-layer view of the DualStyleUNet convolutions on the MFMA path: every distinct (kind, Cin, Cout, H, W, k, stride,
pad) the network issues, how often, and the isolated time / TFLOP/s of its forward, input-gradient and weight-gradient.
    python profiles/conv_layers.py [out.csv] [fp32|split_bf16]"""
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AG_UNFUSED_LAYERS"] = "1"       # the layer list is collected by spying on the per-kernel convolution node
from animatablegaussians_amd import conv as agc, synth  # noqa: E402
from animatablegaussians_amd.styleunet import DualStyleUNet  # noqa: E402

dev = torch.device("cuda:0")
if len(sys.argv) > 2:
    agc.set_math(sys.argv[2])
seen = collections.Counter()
orig = agc._Conv.apply


def spy(x, w, bias, out_scale, kind, stride, padding, weight_scale=1.0):
    cout = w.shape[0] if kind == agc.AG_CONV else w.shape[1]
    seen[(kind, x.shape[1], cout, x.shape[2], x.shape[3], w.shape[-1], stride, padding)] += 1
    return orig(x, w, bias, out_scale, kind, stride, padding, weight_scale)


agc._Conv.apply = spy
net = DualStyleUNet().to(dev)
with torch.no_grad():
    net([torch.ones(1, 512, device=dev) / np.sqrt(512)], synth.pose_map(512).to(dev), randomize_noise=False)
agc._Conv.apply = orig


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3  # us


rows = []
for (kind, cin, cout, h, w, k, s, p), cnt in sorted(seen.items(), key=lambda kv: -kv[1]):
    x = torch.randn(1, cin, h, w, device=dev, requires_grad=True)
    wt = torch.randn((cout, cin, k, k) if kind == agc.AG_CONV else (cin, cout, k, k), device=dev, requires_grad=True)
    y = orig(x, wt, None, None, kind, s, p)
    gy = torch.randn_like(y)
    flops = 2.0 * cin * cout * k * k * (y.shape[2] * y.shape[3] if kind == agc.AG_CONV else h * w)
    tf = timeit(lambda: orig(x.detach(), wt.detach(), None, None, kind, s, p))
    x.grad = wt.grad = None
    xi = x.detach().requires_grad_(True)
    yi = orig(xi, wt.detach(), None, None, kind, s, p)
    ti = timeit(lambda: torch.autograd.grad(yi, xi, gy, retain_graph=True))
    wi = wt.detach().requires_grad_(True)
    yw = orig(x.detach(), wi, None, None, kind, s, p)
    tw = timeit(lambda: torch.autograd.grad(yw, wi, gy, retain_graph=True))
    rows.append((("conv" if kind == agc.AG_CONV else "convT"), cin, cout, h, w, k, s, p, cnt, flops / 1e9, tf, ti, tw))

tot = [sum(r[8] * r[10 + i] for r in rows) for i in range(3)]
totf = sum(r[8] * r[9] for r in rows)
lines = [f"# conv math: {agc.get_math()}", "kind,Cin,Cout,H,W,k,stride,pad,calls,GFLOP,fwd_us,dgrad_us,wgrad_us,fwd_TF,dgrad_TF,wgrad_TF,fwd_share"]
for r in sorted(rows, key=lambda r: -r[8] * r[10]):
    lines.append(",".join(str(v) for v in r[:9]) + f",{r[9]:.2f},{r[10]:.1f},{r[11]:.1f},{r[12]:.1f},"
                 f"{r[9] / r[10] * 1e3:.1f},{r[9] / r[11] * 1e3:.1f},{r[9] / r[12] * 1e3:.1f},{r[8] * r[10] / tot[0]:.3f}")
lines.append(f"# network: {totf:.1f} GFLOP forward; conv time fwd {tot[0] / 1e3:.2f} ms, dgrad {tot[1] / 1e3:.2f} ms, wgrad {tot[2] / 1e3:.2f} ms"
             f" -> {totf / tot[0] * 1e3:.1f} / {totf / tot[1] * 1e3:.1f} / {totf / tot[2] * 1e3:.1f} TFLOP/s")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(out + "\n")
