"""Which convolution shapes change their result when another stream keeps the GPU busy?  (debug probe)
    python profiles/conv_concurrency_probe.py [mode]"""
import collections, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AG_UNFUSED_LAYERS"] = "1"       # the spy below hooks the per-kernel convolution node
from animatablegaussians_amd import conv as agc, synth
from animatablegaussians_amd.styleunet import DualStyleUNet
dev = torch.device("cuda:0")
if len(sys.argv) > 1:
    agc.set_math(sys.argv[1])
seen = collections.OrderedDict()
orig = agc._Conv.apply
def spy(x, w, bias, out_scale, kind, stride, padding, weight_scale=1.0):
    cout = w.shape[0] if kind == agc.AG_CONV else w.shape[1]
    seen[(kind, x.shape[1], cout, x.shape[2], x.shape[3], w.shape[-1], stride, padding)] = 1
    return orig(x, w, bias, out_scale, kind, stride, padding, weight_scale)
agc._Conv.apply = spy
net = DualStyleUNet().to(dev)
with torch.no_grad():
    net([torch.ones(1, 512, device=dev) / np.sqrt(512)], synth.pose_map(512).to(dev), randomize_noise=False)
agc._Conv.apply = orig
side = torch.cuda.Stream()
other = torch.cuda.Stream()
bx = torch.randn(1, 256, 128, 128, device=dev); bw = torch.randn(256, 256, 3, 3, device=dev)
big = torch.randn(64 << 20, device=dev)
nbad = 0
for (kind, cin, cout, h, w, k, s, p) in seen:
    x = torch.randn(1, cin, h, w, device=dev)
    wt = torch.randn((cout, cin, k, k) if kind == agc.AG_CONV else (cin, cout, k, k), device=dev)
    with torch.no_grad():
        y0 = orig(x, wt, None, None, kind, s, p, 1.0).clone()
        torch.cuda.synchronize()
        diffs = []
        for rep in range(6):
            with torch.cuda.stream(side):
                for _ in range(3):
                    orig(bx, bw, None, None, agc.AG_CONV, 1, 1, 1.0)       # another convolution on a second stream
            with torch.cuda.stream(other):
                big.mul_(1.0001)
            y = orig(x, wt, None, None, kind, s, p, 1.0)
            diffs.append(float((y - y0).abs().max()))
        torch.cuda.synchronize()
    if max(diffs) > 0:
        nbad += 1
        print("CHANGES", ("conv" if kind == agc.AG_CONV else "convT"), cin, cout, h, w, k, s, p, ["%.2e" % d for d in diffs], "scale %.2e" % float(y0.abs().max()))
print(nbad, "of", len(seen), "shapes change under concurrency, mode", agc.get_math())
