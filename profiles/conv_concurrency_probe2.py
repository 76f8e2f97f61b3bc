"""Victim: the 3 -> 512 1x1 convolution (pointwise kernel, 256 bytes of LDS) and a plain torch op; aggressors: one kind of work on a second
stream.  Which aggressor changes the victim's result?  (debug probe)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import conv as agc
dev = torch.device("cuda:0")
orig = agc._Conv.apply
side = torch.cuda.Stream()
x = torch.randn(1, 3, 32, 32, device=dev); wt = torch.randn(512, 3, 1, 1, device=dev)
xv = torch.randn(1, 64, 64, 64, device=dev); wv = torch.randn(64, 64, 3, 3, device=dev)     # second victim: a small MFMA conv
b128 = (torch.randn(1, 256, 128, 128, device=dev), torch.randn(256, 256, 3, 3, device=dev))
b64 = (torch.randn(1, 64, 256, 256, device=dev), torch.randn(64, 64, 3, 3, device=dev))
big = torch.randn(64 << 20, device=dev)

def aggress(kind):
    if kind == "none":
        return
    if kind == "elementwise":
        big.mul_(1.0001); return
    mode, which = kind.split(":")
    agc.set_math(mode)
    a, b = b128 if which == "bm128" else b64
    if which == "wgrad":
        a = b128[0]; bb = b128[1].clone().requires_grad_(True)
        with torch.enable_grad():
            y = orig(a, bb, None, None, agc.AG_CONV, 1, 1, 1.0)
            torch.autograd.grad(y, bb, torch.ones_like(y))
        return
    for _ in range(3):
        orig(a, b, None, None, agc.AG_CONV, 1, 1, 1.0)

with torch.no_grad():
    agc.set_math("split_bf16")
    y0 = orig(x, wt, None, None, agc.AG_CONV, 1, 0, 1.0).clone()
    red_in = torch.randn(8192, 4096, device=dev)
    v0 = (torch.softmax(red_in, dim=1).sum(dim=1) + torch.logsumexp(red_in, dim=1)).clone()
    torch.cuda.synchronize()
    for kind in ("none", "elementwise", "fp32:bm128", "split_bf16:bm128", "split_bf16:bm64", "split_bf16x3:bm128", "split_bf16:wgrad", "fp32:wgrad"):
        d1, d2 = [], []
        for rep in range(8):
            with torch.cuda.stream(side):
                aggress(kind)
            agc.set_math("split_bf16")
            y = orig(x, wt, None, None, agc.AG_CONV, 1, 0, 1.0)
            v = torch.softmax(red_in, dim=1).sum(dim=1) + torch.logsumexp(red_in, dim=1)      # LDS-using torch reductions
            d1.append(float((y - y0).abs().max())); d2.append(float((v - v0).abs().max()))
            torch.cuda.synchronize()
        print(f"{kind:22s} pointwise victim: {sum(d > 0 for d in d1)}/8 changed (max {max(d1):.2e})   torch-reduction victim: {sum(d > 0 for d in d2)}/8 changed (max {max(d2):.2e})")
