"""Throughput of the MFMA convolutions at DualStyleUNet layer sizes vs torch (MIOpen) on the same GPU.  Debug aid."""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from animatablegaussians_amd import conv as agc
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
cases = [("conv", 512, 512, 64, 3, 1, 1), ("conv", 1024, 512, 64, 3, 1, 1), ("conv", 256, 256, 128, 3, 1, 1),
         ("conv", 128, 128, 256, 3, 1, 1), ("conv", 64, 64, 512, 3, 1, 1), ("conv", 128, 256, 257, 3, 2, 0),
         ("convT", 512, 512, 32, 3, 2, 0), ("convT", 128, 64, 256, 3, 2, 0), ("conv", 64, 12, 512, 1, 1, 0)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for kind, Cin, Cout, S, k, s, p in cases:
    x = torch.randn(1, Cin, S, S, device='cuda', requires_grad=True)
    w = (torch.randn(*((Cout, Cin, k, k) if kind == 'conv' else (Cin, Cout, k, k)), device='cuda') * 0.05).requires_grad_(True)
    ours = (lambda: agc.conv2d(x, w, None, stride=s, padding=p)) if kind == 'conv' else (lambda: agc.conv_transpose2d(x, w, None, stride=2))
    ref = (lambda: F.conv2d(x, w, None, stride=s, padding=p)) if kind == 'conv' else (lambda: F.conv_transpose2d(x, w, None, stride=2))
    y = ours(); gy = torch.randn_like(y)
    flop = 2.0 * y.numel() * Cin * k * k if kind == 'conv' else 2.0 * x.numel() * Cout * k * k
    def fb(f):
        def run():
            x.grad = None; w.grad = None
            f().backward(gy)
        return run
    with torch.no_grad():
        to, tr = timeit(ours), timeit(ref)
    tob, trb = timeit(fb(ours)), timeit(fb(ref))
    print(f"{kind:5s} {Cin:4d}->{Cout:4d} @{S:3d} k{k} s{s}: fwd ours {to*1e3:7.3f} ms ({flop/to/1e12:5.1f} TF)  torch {tr*1e3:7.3f} ms ({flop/tr/1e12:5.1f} TF) | "
          f"fwd+bwd ours {tob*1e3:7.3f} ms ({3*flop/tob/1e12:5.1f} TF)  torch {trb*1e3:7.3f} ms ({3*flop/trb/1e12:5.1f} TF)")
