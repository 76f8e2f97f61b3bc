"""Replicates the order of bench_avatar.full_step_probe's legs and prints every block, to find which leg disturbs the next."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_avatar  # noqa: E402
from animatablegaussians_amd import conv as agc  # noqa: E402

dev = torch.device("cuda:0")
step = bench_avatar.TrainingStep(dev)
for mode, V in [("split_bf16", 1), ("split_bf16", 4), ("split_bf16x3", 1), ("split_bf16x3", 4), ("split_bf16", 4), ("split_bf16", 16), ("split_bf16", 4)]:
    agc.set_math(mode)
    torch.cuda.empty_cache()
    bench_avatar.timed(lambda i: step(i, V), 1, 2, dev)
    blocks = [bench_avatar.timed(lambda i: step(i, V), 2, 0, dev) for _ in range(4)]
    st = torch.cuda.memory_stats(dev)
    print(f"{mode:14s} V = {V:2d}: " + " ".join(f"{b:7.1f}" for b in blocks) + f" ms; reserved {torch.cuda.memory_reserved(dev) / 2**30:.1f} GiB, "
          f"allocated {torch.cuda.memory_allocated(dev) / 2**30:.1f} GiB, mallocs {st['num_device_alloc']}, frees {st['num_device_free']}", flush=True)
