"""Which Python lines of the training step launch the device kernels that are not ours (torch glue)?
    python profiles/step_glue_sources.py [views]
Groups every non-`ag::` kernel of ONE training step by (kernel family, aten operator, innermost frame inside this repository), with the
summed device time, so that each fill / copy / add / cat / reduction can be traced to the line that asked for it."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("AG_PKG_ROOT"):      # same-box A/B against another copy of the Python package (same native library)
    sys.path.insert(0, os.path.abspath(os.environ["AG_PKG_ROOT"]))
    os.environ.setdefault("AG_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "animatablegaussians_amd", "lib", "libag_hip.so"))
import bench_avatar  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

views = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
step = bench_avatar.TrainingStep(dev)
for i in range(4):
    step(i, views)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(4, views)
    torch.cuda.synchronize()


def family(name):
    for key, tag in (("FillFunctor", "fill"), ("fillBuffer", "memset"), ("copyBuffer", "memcpy"), ("Memcpy", "memcpy"), ("Memset", "memset"),
                     ("CatArray", "cat"), ("CUDAFunctor_add", "add"), ("CUDAFunctorOnSelf_add", "add_"), ("AUnaryFunctor", "mul_scalar"),
                     ("BUnaryFunctor", "b_unary"), ("BinaryFunctor", "binary"), ("reduce_kernel", "reduce"), ("multi_tensor_apply", "multi_tensor"),
                     ("Cijk_", "hipblaslt"), ("upsample", "upsample"), ("direct_copy", "copy_kernel"), ("index", "index"), ("radixSort", "sort")):
        if key in name:
            return tag
    return name[:48]


def frame_of(e):
    st = getattr(e, "stack", None) or []
    for f in st:
        if ROOT in f and "profiles/" not in f:
            return f.replace(ROOT + "/", "")[:90]
    return (st[0][:90] if st else "?")


groups = collections.defaultdict(lambda: [0, 0.0])
ours = [0, 0.0]
for e in prof.events():
    ks = getattr(e, "kernels", None) or []
    for k in ks:
        name = k.name
        dur = float(getattr(k, "duration", 0.0) or 0.0)
        if name.startswith("ag::") or " ag::" in name or "void ag::" in name:
            ours[0] += 1
            ours[1] += dur
            continue
        g = groups[(family(name), e.name, frame_of(e), str(e.input_shapes)[:60])]
        g[0] += 1
        g[1] += dur
tot_n = sum(v[0] for v in groups.values())
tot_t = sum(v[1] for v in groups.values())
print(f"views {views}: our kernels {ours[0]} launches {ours[1] / 1e3:.2f} ms; other kernels {tot_n} launches {tot_t / 1e3:.2f} ms")
byfam = collections.defaultdict(lambda: [0, 0.0])
for (fam, _, _, _), v in groups.items():
    byfam[fam][0] += v[0]
    byfam[fam][1] += v[1]
for fam, v in sorted(byfam.items(), key=lambda kv: -kv[1][1]):
    print(f"  {fam:24s} {v[0]:4d} launches {v[1]:9.1f} us")
print()
for (fam, op, fr, shp), v in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1]:8.1f} us {v[0]:4d}x  {fam:14s} {op:34s} {fr:90s} {shp}")
