#!/usr/bin/env python
"""Time of one data item's SMPL-X work (three model evaluations + cano2live products, dataset_mv_rgb.py:118-171):
MI355X kernels (SMPLX.data_item, HIP-event timed, and the skinning launch alone) against the CPU oracle restatement on the
host cores (what the reference's data loader does per item).  `python profiles/smplx_time.py`"""
import json
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animatablegaussians_amd import synth  # noqa: E402
from animatablegaussians_amd.smplx import SMPLX  # noqa: E402
from oracle import smplx_oracle as so  # noqa: E402

arrays, params = synth.smplx_model_arrays(), synth.smplx_pose_params(n=4)
cp = np.zeros(75, np.float32)
cp[5], cp[8] = math.radians(25), math.radians(-25)
cp = torch.from_numpy(cp)
dev = torch.device("cuda", 0)
model = SMPLX(arrays, use_pca=False, flat_hand_mean=True, device=dev)
dp = {k: torch.from_numpy(v) for k, v in params.items()}     # host tensors, as the reference's dataset holds them
for i in range(5):
    model.data_item(dp, i % 4, cp[3:6], cp[:3], cp[6:69])
torch.cuda.synchronize()
go, tr, bp = cp[3:6], cp[:3], cp[6:69]
n = 200
t0 = time.perf_counter()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(n):
    model.data_item(dp, i % 4, go, tr, bp)
e1.record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n
gpu = e0.elapsed_time(e1) / n

# the batched model evaluation alone (4 kernels + key points), back to back
comps = torch.randn(3, 20, device=dev)
pose = torch.randn(3, 165, device=dev) * 0.3
trl = torch.randn(3, 3, device=dev)
for _ in range(5):
    model.lbs(comps, pose, trl)
e0.record()
for _ in range(n):
    model.lbs(comps, pose, trl)
e1.record()
torch.cuda.synchronize()
lbs_ms = e0.elapsed_time(e1) / n

m = so.model_tensors(arrays, torch.float32)
cpu = {}
for threads in (1, 8, 32):                     # a data-loader worker is single-threaded; more threads only help the 61-MB GEMV
    torch.set_num_threads(threads)
    so.data_item(m, params, 0, cp[3:6], cp[:3], cp[6:69])
    t0 = time.perf_counter()
    k = 10
    for i in range(k):
        so.data_item(m, params, i % 4, cp[3:6], cp[:3], cp[6:69])
    cpu[threads] = round(1e3 * (time.perf_counter() - t0) / k, 2)
bytes_once = arrays['posedirs'].size * 4 + 3 * (10475 * 3 * 20 * 4 + 2 * 55 * 10475 * 4)
print(json.dumps({
    "data_item_ms_wall": round(1e3 * wall, 4), "data_item_ms_gpu": round(gpu, 4), "batched_lbs_ms_gpu": round(lbs_ms, 4),
    "posedirs_stream_GBps_lower_bound": round(bytes_once / (lbs_ms * 1e-3) / 1e9, 1),
    "cpu_oracle_ms_by_threads": cpu, "host_cores": os.cpu_count()}))
