"""A few training steps (1 view per step) for rocprofv3 --kernel-trace --stats:  python profiles/fullstep_prof.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_avatar  # noqa: E402

dev = torch.device("cuda:0")
step = bench_avatar.TrainingStep(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for i in range(3 + n):
    step(i, 1)
torch.cuda.synchronize()
