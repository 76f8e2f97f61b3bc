"""ms per training step at V views of one pose per step (render_views):  python profiles/views_scaling.py [V ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_avatar  # noqa: E402

dev = torch.device("cuda:0")
step = bench_avatar.TrainingStep(dev)
for V in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16]:
    torch.cuda.empty_cache()
    ms = bench_avatar.timed(lambda i: step(i, V), 3, 2, dev)
    print(f"V = {V:2d}: {ms:8.2f} ms per step, {1e3 * V / ms:6.2f} views/s, peak memory {torch.cuda.max_memory_allocated(dev) / 2**30:.1f} GiB", flush=True)
