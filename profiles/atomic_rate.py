"""Line-coalesced float-atomic rate of the memory side (ag_debug_atomic_rate): the ceiling of flushing the blend backward's sums with
less pre-reduction.  One wave instruction = 4 accumulator lines x `comps` floats; lines drawn pseudo-randomly from P = 268 348."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import _lib  # noqa: E402

L = _lib.lib()
P = 268348
acc = torch.zeros(P * 16, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for blocks in (512, 768, 2048):
    for comps in (10, 16):
        for iters in (64, 256):
            L.ag_debug_atomic_rate(ctypes.c_void_p(acc.data_ptr()), P, blocks, iters, comps, st)
            torch.cuda.synchronize()
            e0.record()
            L.ag_debug_atomic_rate(ctypes.c_void_p(acc.data_ptr()), P, blocks, iters, comps, st)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3
            lines = blocks * 8 * iters * 4
            print(f"blocks {blocks:5d} comps {comps:2d} iters {iters:4d}: {us:8.1f} us for {lines / 1e6:6.2f} M line requests -> {lines / us:7.1f} lines/us"
                  f" ({lines * comps * 4 / us / 1e3:6.1f} GB/s of operands)")
