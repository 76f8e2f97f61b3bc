"""Pattern of the corruption of the 3 -> 512 pointwise convolution under a concurrent split convolution (debug probe)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import conv as agc
dev = torch.device("cuda:0")
orig = agc._Conv.apply
side = torch.cuda.Stream()
x = torch.randn(1, 3, 32, 32, device=dev); wt = torch.randn(512, 3, 1, 1, device=dev)
b128 = (torch.randn(1, 256, 128, 128, device=dev), torch.randn(256, 256, 3, 3, device=dev))
with torch.no_grad():
    y0 = orig(x, wt, None, None, agc.AG_CONV, 1, 0, 1.0).clone()
    ref = (wt.view(512, 3) @ x.view(3, -1)).view_as(y0)
    print("serial result vs matmul:", float((y0 - ref).abs().max()))
    torch.cuda.synchronize()
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(3):
                orig(*b128, None, None, agc.AG_CONV, 1, 1, 1.0)
        y = orig(x, wt, None, None, agc.AG_CONV, 1, 0, 1.0)
        torch.cuda.synchronize()
        bad = ((y - y0).abs() > 0)[0].view(512, -1)
        rows = bad.any(dim=1).nonzero().flatten().tolist()
        cols = bad.any(dim=0).nonzero().flatten().tolist()
        print(f"rep {rep}: {int(bad.sum())} bad elements; rows {rows[:12]}{'...' if len(rows) > 12 else ''} ({len(rows)}); cols {cols[:8]}... ({len(cols)})",
              "nan" if torch.isnan(y).any() else "", "x intact", bool(torch.equal((wt.view(512, 3) @ x.view(3, -1)).view_as(y0), ref)))
        for m in rows[:3]:
            n = int(bad[m].nonzero().flatten()[0])
            yy, y00 = y[0].view(512, -1), y0[0].view(512, -1)
            print(f"     y[{m}][{n}] = {float(yy[m, n]):.6f}; serial result {float(y00[m, n]):.6f}; serial row m+1: {float(y00[min(m + 1, 511), n]):.6f}, row m-1: {float(y00[m - 1, n]):.6f}, row m+2: {float(y00[min(m + 2, 511), n]):.6f}")
