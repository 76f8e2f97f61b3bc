"""Poison the conv workspace (NaN bytes) before every call and look for NaNs / mode differences per layer (debug probe)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import conv as agc, synth
from animatablegaussians_amd.styleunet import DualStyleUNet
dev = torch.device("cuda:0")
orig_ws = agc._workspace
def poisoned(d, dev_):
    t, n = orig_ws(d, dev_)
    t.fill_(0xFF)
    return t, n
agc._workspace = poisoned
orig = agc._Conv.apply
bad = []
def spy(x, w, bias, out_scale, kind, stride, padding, weight_scale=1.0):
    agc.set_math("fp32")
    y0 = orig(x, w, bias, out_scale, kind, stride, padding, weight_scale)
    agc.set_math("split_bf16")
    y1 = orig(x, w, bias, out_scale, kind, stride, padding, weight_scale)
    d = float((y0 - y1).abs().max()); sc = float(y0.abs().max())
    if not (d <= 1e-4 * sc):
        bad.append((kind, tuple(x.shape), tuple(w.shape), stride, padding, d, sc, bool(torch.isnan(y1).any()), bool(torch.isnan(y0).any())))
    return y1
agc._Conv.apply = spy
net = DualStyleUNet(out_channels=3).to(dev) if False else DualStyleUNet().to(dev)
with torch.no_grad():
    net([torch.ones(1, 512, device=dev) / np.sqrt(512)], synth.pose_map(512).to(dev), randomize_noise=False)
print(len(bad), "layers differ between the modes beyond 1e-4 of scale")
for b in bad: print(b)
