import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from animatablegaussians_amd import conv as agc, synth
from animatablegaussians_amd.styleunet import DualStyleUNet
dev = torch.device("cuda:0")
orig = agc._Conv.apply
bad = []
def spy(x, w, bias, out_scale, kind, stride, padding, weight_scale=1.0):
    y1 = orig(x, w, bias, out_scale, kind, stride, padding, weight_scale)
    y2 = orig(x, w, bias, out_scale, kind, stride, padding, weight_scale)
    if not torch.equal(y1, y2):
        d = (y1 - y2).abs()
        bad.append((kind, tuple(x.shape), tuple(w.shape), stride, padding, float(d.max()), int((d > 0).sum()), torch.isnan(y1).any().item()))
    return y1
agc._Conv.apply = spy
net = DualStyleUNet().to(dev)
with torch.no_grad():
    net([torch.ones(1, 512, device=dev) / np.sqrt(512)], synth.pose_map(512).to(dev), randomize_noise=False)
print(len(bad), "non-repeatable convs")
for b in bad: print(b)
