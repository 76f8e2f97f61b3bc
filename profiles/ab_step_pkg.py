"""Same-box A/B of the training step between two copies of the Python package (same native library):
    python profiles/ab_step_pkg.py <root that holds the other animatablegaussians_amd/ | -> [views ...]
'-' = the package of this checkout.  Prints ms per step (bench_avatar.TrainingStep, 3 timed blocks)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
other = sys.argv[1]
os.environ.setdefault("AG_LIB_PATH", os.path.join(ROOT, "animatablegaussians_amd", "lib", "libag_hip.so"))
sys.path.insert(0, ROOT)
if other != "-":
    sys.path.insert(0, os.path.abspath(other))
import torch  # noqa: E402
import bench_avatar  # noqa: E402
import animatablegaussians_amd  # noqa: E402

dev = torch.device("cuda:0")
step = bench_avatar.TrainingStep(dev)
for V in [int(a) for a in sys.argv[2:]] or [1]:
    torch.cuda.empty_cache()
    ms = [bench_avatar.timed(lambda i: step(i, V), 8, 3, dev) for _ in range(3)]
    print(f"{os.path.dirname(animatablegaussians_amd.__file__)}: V = {V}: " + " ".join(f"{m:.2f}" for m in ms) + " ms per step", flush=True)
