"""Workload for the HBM-traffic PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one counter per pass):
a calibration copy of known size (torch clone of 256 MiB: 256 MiB read + 256 MiB written) followed by a few steps of the
headline bench workload.  profiles/traffic_summarize.py turns the two counter CSVs into per-kernel bytes per launch,
applying the gfx950 FETCH_SIZE correction the calibration copy measures (MI355X_MICROARCH.md, HBM section)."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
x = torch.empty(64 << 20, dtype=torch.float32, device="cuda:0").normal_()
torch.cuda.synchronize()
for _ in range(4):
    y = x.clone()
torch.cuda.synchronize()
del x, y
# one stream: with views overlapped the counters of a kernel include whatever ran beside it; no full-step legs (their large device copies
# would be mistaken for the calibration copy)
sys.argv = [os.path.join(ROOT, "bench.py"), "--steps", "16", "--warmup", "4", "--no-cpu-baseline", "--no-full-step", "--no-stress", "--streams", "1"]
sys.path.insert(0, ROOT)
import bench  # noqa: E402

bench.main()

# the per-Gaussian assembly and skinning kernels (gather_* / lbs_* of ag_avatar.hip), a few eager launches each
import bench_avatar  # noqa: E402
from animatablegaussians_amd import avatar_ops as ops  # noqa: E402
from animatablegaussians_amd.avatar import AvatarRenderCore  # noqa: E402

dev = torch.device("cuda:0")
core = AvatarRenderCore.synthetic(device=dev)
S = int(core.map_side)
maps = [torch.randn(1, c, S, S, device=dev).requires_grad_(True) for c in (6, 16, 6)]
A = bench_avatar.joint_transforms(core.lbs.shape[1], dev)
for _ in range(6):
    pos, opa, sca, rot, col = ops.gather_activate(*maps, core.pix, core.xyz, core.opacity_raw, core.scaling_raw, core.rotation_raw)
    p2, r2 = ops.lbs_transform(pos, rot, core.lbs, A, core.lbs_sparse)
    (p2.sum() + r2.sum() + opa.sum() + sca.sum() + col.sum()).backward()
torch.cuda.synchronize()
