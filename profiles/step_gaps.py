"""Where does the GPU sit idle inside a training step?   python profiles/step_gaps.py [V]
Profiles one pipelined step (the next step already queued behind it) with the torch profiler, sorts the device kernels by start time and adds up the
idle time between the end of everything before a kernel and its start, by the family of the kernel that ENDED last and of the one that starts."""
import collections
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_avatar  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
step = bench_avatar.TrainingStep(dev)


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"<.*$", "", name)
    name = re.sub(r"at::native::\(anonymous namespace\)::|at::native::", "", name)
    return name[:44]


for i in range(4):
    step(i, V)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(4, 8):
        step(i, V)
    torch.cuda.synchronize()
ks = []
for e in prof.events():
    for k in getattr(e, "kernels", None) or []:
        pass
for e in prof.events():
    if e.device_type is not None and "cuda" in str(e.device_type).lower() and e.time_range is not None:
        ks.append((e.time_range.start, e.time_range.end, short(e.name)))
ks.sort()
if not ks:
    print("no device events")
    sys.exit(0)
# the middle two steps of the four: away from the fill and the drain of the queue
t0, t1 = ks[0][0], max(k[1] for k in ks)
lo, hi = t0 + (t1 - t0) * 0.25, t0 + (t1 - t0) * 0.75
busy_end, last = None, None
idle = collections.defaultdict(lambda: [0, 0.0])
hist = collections.Counter()
tot_idle = tot_busy = 0.0
n = 0
for s, e, name in ks:
    if busy_end is not None and s >= lo and s <= hi:
        gap = s - busy_end
        if gap > 0:
            tot_idle += gap
            a = idle[(last, name)]
            a[0] += 1
            a[1] += gap
            hist[min(int(gap), 20)] += 1
        else:
            hist[0] += 1
        n += 1
        tot_busy += max(0.0, e - max(s, busy_end))
    if busy_end is None or e > busy_end:
        busy_end, last = e, name
span = hi - lo
print(f"V = {V}: window {span / 1e3:.2f} ms (two of four pipelined steps), {n} launches, device busy {tot_busy / 1e3:.2f} ms, idle {tot_idle / 1e3:.2f} ms = "
      f"{100 * tot_idle / span:.1f} % of the window, {tot_idle / max(n, 1):.2f} us per launch")
print("gap histogram (us -> launches):", dict(sorted(hist.items())))
print("idle time by (kernel that ended last -> kernel that starts), top 25:")
for (a, b), (c, t) in sorted(idle.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t:8.1f} us {c:5d} x {t / c:6.2f}  {a} -> {b}")
by_next = collections.defaultdict(lambda: [0, 0.0])
for (a, b), (c, t) in idle.items():
    by_next[b][0] += c
    by_next[b][1] += t
print("idle time in front of a kernel family, top 15:")
for b, (c, t) in sorted(by_next.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"  {t:8.1f} us {c:5d} x {t / c:6.2f}  {b}")
