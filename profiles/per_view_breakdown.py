"""What does one more view of the same pose cost in a training step, kernel by kernel?   python profiles/per_view_breakdown.py [Va Vb]
Profiles one step at Va (default 1) and one at Vb (default 3) views of one pose with the torch profiler (device kernel times) and prints
(time at Vb - time at Va) / (Vb - Va) per kernel family: the per-extra-view bill of the view-dependent tail (colour decoder's last stage, gather,
LBS, rasterizer, loss) that decides BASELINE configs[2] / [3]."""
import collections
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("AG_PKG_ROOT"):      # same-box A/B against another copy of the Python package (same native library)
    sys.path.insert(0, os.path.abspath(os.environ["AG_PKG_ROOT"]))
    os.environ.setdefault("AG_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "animatablegaussians_amd", "lib", "libag_hip.so"))
import bench_avatar  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

Va, Vb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 3)
dev = torch.device("cuda:0")
step = bench_avatar.TrainingStep(dev)


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"at::native::\(anonymous namespace\)::|at::native::", "", name)
    m = re.match(r"(vectorized_elementwise_kernel|elementwise_kernel_manual_unroll|reduce_kernel)<.*?(\w+Functor\w*|\w+Ops|func_wrapper_t|direct_copy\w*|compare_scalar\w*)", name)
    if m:
        return m.group(1) + ":" + m.group(2)
    return name[:70]


def one(V):
    for i in range(4):
        step(i, V)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step(4, V)
        torch.cuda.synchronize()
    acc = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        for k in getattr(e, "kernels", None) or []:
            a = acc[short(k.name)]
            a[0] += 1
            a[1] += float(k.duration)
    return acc


a, b = one(Va), one(Vb)
if os.environ.get("AG_DUMP") == "1":      # the whole kernel table of the step at Va views
    print(f"--- step at {Va} view(s), every kernel family ---")
    for k, (n, t) in sorted(a.items(), key=lambda kv: -kv[1][1]):
        print(f"{t:10.1f} us {n:5d}  {k}")
    print("--- end ---")
keys = sorted(set(a) | set(b), key=lambda k: -(b.get(k, [0, 0.0])[1] - a.get(k, [0, 0.0])[1]))
dv = Vb - Va
tot_t = sum(b[k][1] for k in b) - sum(a[k][1] for k in a)
tot_n = sum(b[k][0] for k in b) - sum(a[k][0] for k in a)
print(f"step at {Va} view(s): {sum(v[1] for v in a.values()) / 1e3:.2f} ms of kernels in {sum(v[0] for v in a.values())} launches; at {Vb}: "
      f"{sum(v[1] for v in b.values()) / 1e3:.2f} ms in {sum(v[0] for v in b.values())}")
print(f"per extra view: {tot_t / dv / 1e3:.3f} ms of kernel time, {tot_n / dv:.0f} launches")
print(f"{'us / view':>10s} {'launches':>9s}  kernel")
cum = 0.0
for k in keys:
    d = (b.get(k, [0, 0.0])[1] - a.get(k, [0, 0.0])[1]) / dv
    n = (b.get(k, [0, 0.0])[0] - a.get(k, [0, 0.0])[0]) / dv
    if abs(d) < 3.0:
        continue
    cum += d
    print(f"{d:10.1f} {n:9.1f}  {k}")
print(f"(rows shown: {cum / 1e3:.3f} ms)")
