"""Per-launch table of the MFMA convolutions of ONE training step (1 view): shape, tile, split-K count, HIP-event duration, TFLOP/s, and
the launch's distance from the fp16 matrix pipe's power floor on real operands (profiles/r05_mfma_floor_f16.txt: 20.7 ns per MFMA per SIMD).
    python profiles/conv_launch_table.py [out.csv]
Every gather-conv / wgrad launch is bracketed by HIP events on its stream (ag_prof_*, tagged with the shape by the launcher)."""
import collections
import csv
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_avatar  # noqa: E402
from animatablegaussians_amd import _lib, conv as agc  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/conv_launch_table.csv"
FLOOR_NS = float(os.environ.get("AG_MFMA_FLOOR_NS", "20.7"))
dev = torch.device("cuda:0")
step = bench_avatar.TrainingStep(dev)
for i in range(6):
    step(i, 1)
torch.cuda.synchronize()
_lib.prof_enable([_lib.AG_K_GATHER_CONV, _lib.AG_K_WGRAD])
step(6, 1)
torch.cuda.synchronize()
n, ms, work = _lib.prof_collect_work(records_path=out)
_lib.prof_enable([])
terms = {"split_f16": 3, "split_bf16": 6, "split_bf16x3": 3, "f16": 1}.get(agc.get_math(), 0)
rows = list(csv.DictReader(open(out)))
groups = collections.OrderedDict()
for r in rows:
    k = (r["kernel"], r["tag"])
    g = groups.setdefault(k, [0, 0.0, 0.0])
    g[0] += 1
    g[1] += float(r["ms"])
    g[2] += float(r["work"])
tot_ms = sum(g[1] for g in groups.values())
tot_w = sum(g[2] for g in groups.values())
print(f"math {agc.get_math()}: {len(rows)} launches, {tot_ms:.2f} ms, {tot_w / 1e9:.0f} GFLOP, {tot_w / tot_ms / 1e9:.1f} TFLOP/s; floor {FLOOR_NS} ns per MFMA per SIMD")
print(f"{'ms':>8s} {'n':>3s} {'GFLOP':>8s} {'TF/s':>7s} {'floor ms':>8s} {'x floor':>7s}  tag")
lost = 0.0
for (kern, tag), (cnt, t, w) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    mfma_per_simd = w * terms / 32768.0 / 1024.0 if terms else 0.0
    floor_ms = mfma_per_simd * FLOOR_NS * 1e-6
    lost += max(0.0, t - floor_ms)
    print(f"{t:8.3f} {cnt:3d} {w / 1e9:8.1f} {w / t / 1e9 if t else 0:7.1f} {floor_ms:8.3f} {t / floor_ms if floor_ms else 0:7.2f}  {tag}")
print(f"time above the matrix pipe's power floor: {lost:.2f} of {tot_ms:.2f} ms")
