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
