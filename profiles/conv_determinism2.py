import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from animatablegaussians_amd import conv as agc
dev = torch.device("cuda:0")
for (cin, cout, hw, k, s, p) in [(1, 64, 512, 4, 2, 1), (64, 128, 256, 4, 2, 1), (3, 128, 513, 3, 2, 0), (12, 64, 64, 3, 1, 1)]:
    x = torch.randn(1, cin, hw, hw, device=dev); w = torch.randn(cout, cin, k, k, device=dev)
    ys = [agc.conv2d(x, w, None, stride=s, padding=p) for _ in range(4)]
    # dirty the workspace in between
    junk = agc.conv2d(torch.randn(1, 512, 32, 32, device=dev), torch.randn(512, 512, 3, 3, device=dev), None, padding=1)
    ys.append(agc.conv2d(x, w, None, stride=s, padding=p))
    ref = torch.nn.functional.conv2d(x.cpu().double(), w.cpu().double(), None, stride=s, padding=p)
    print(cin, cout, hw, k, [bool(torch.equal(ys[0], y)) for y in ys[1:]], float((ys[0].cpu().double() - ref).abs().max()), float((ys[-1].cpu().double() - ref).abs().max()), torch.isnan(ys[-1]).any().item())
