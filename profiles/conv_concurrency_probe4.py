"""Is a bare v_mfma_f32_32x32x16_bf16 stream on a second HIP stream enough to disturb the 3 -> 512 pointwise convolution?  (debug probe)"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import _lib, conv as agc
dev = torch.device("cuda:0")
orig = agc._Conv.apply
side = torch.cuda.Stream()
x = torch.randn(1, 3, 32, 32, device=dev); wt = torch.randn(512, 3, 1, 1, device=dev)
out = torch.zeros(64, device=dev)
L = _lib.lib()
with torch.no_grad():
    y0 = orig(x, wt, None, None, agc.AG_CONV, 1, 0, 1.0).clone()
    torch.cuda.synchronize()
    for name, fn in (("bf16 MFMA stream", L.ag_debug_mfma_rate_bf16), ("fp32 MFMA stream", L.ag_debug_mfma_rate)):
        changed = 0
        for rep in range(10):
            fn(2048, 20000, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(side.cuda_stream))
            ys = [orig(x, wt, None, None, agc.AG_CONV, 1, 0, 1.0) for _ in range(20)]
            torch.cuda.synchronize()
            changed += sum(int(not torch.equal(y, y0)) for y in ys)
        print(f"{name}: {changed} of 200 pointwise results changed")
