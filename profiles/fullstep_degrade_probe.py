"""Why do multi-view steps get slower over time in one process?  Host issue vs GPU time per step, a fixed reference kernel, allocator state."""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_avatar  # noqa: E402

dev = torch.device("cuda:0")
step = bench_avatar.TrainingStep(dev)
ref = torch.empty(64 << 20, device=dev)


def ref_us():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        ref.mul_(1.0001)
    e1.record()
    e1.synchronize()
    return 100 * e0.elapsed_time(e1)


V = int(os.environ.get("V", "4"))
for i in range(int(os.environ.get("N", "40"))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(i, V)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    st = torch.cuda.memory_stats(dev)
    if i % 4 == 0 or i > 34:
        print(f"step {i:3d}: host {1e3 * (t1 - t0):7.1f} ms, total {1e3 * (t2 - t0):7.1f} ms, ref kernel {ref_us():6.1f} us, reserved "
              f"{torch.cuda.memory_reserved(dev) / 2**30:5.1f} GiB, peak alloc {torch.cuda.max_memory_allocated(dev) / 2**30:5.1f} GiB, "
              f"mallocs {st['num_device_alloc']}, gc objects {len(gc.get_objects())}, segments {st['segment.all.current']}, "
              f"inactive split {st['inactive_split.all.current']}", flush=True)
