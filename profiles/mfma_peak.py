"""Attainable fp32 MFMA rate of this box: waves issuing nothing but v_mfma_f32_32x32x2_f32 (the instruction the
convolutions run on), at 1..8 waves per SIMD, short and long runs (clock behaviour under sustained matrix load)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
out = torch.zeros(16, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for waves_per_simd in (1, 2, 4):
    blocks = 256 * waves_per_simd
    for iters in (2000, 20000, 200000):
        L.ag_debug_mfma_rate(blocks, 100, ctypes.c_void_p(out.data_ptr()), stream)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        L.ag_debug_mfma_rate(blocks, iters, ctypes.c_void_p(out.data_ptr()), stream)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        flops = blocks * 4 * iters * 4 * (32 * 32 * 2 * 2)
        print(f"{waves_per_simd} waves/SIMD, {iters:6d} iters: {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s")

print("v_mfma_f32_32x32x16_bf16 (the instruction a bf16 split of the fp32 operands would run on):")
for waves_per_simd in (1, 2, 4):
    blocks = 256 * waves_per_simd
    for iters in (20000, 200000):
        L.ag_debug_mfma_rate_bf16(blocks, 100, ctypes.c_void_p(out.data_ptr()), stream)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        L.ag_debug_mfma_rate_bf16(blocks, iters, ctypes.c_void_p(out.data_ptr()), stream)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        flops = blocks * 4 * iters * 4 * (32 * 32 * 16 * 2)
        print(f"{waves_per_simd} waves/SIMD, {iters:6d} iters: {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s  (/3 = {flops / ms / 3e9:6.1f} fp32-equivalent)")
