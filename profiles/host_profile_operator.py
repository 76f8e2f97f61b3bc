"""cProfile of the host side of the OPERATOR path (GaussianRasterizer + torch.autograd.backward, bench.py --operator-path): where the Python time per
view goes.  python profiles/host_profile_operator.py"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [os.path.join(ROOT, "bench.py"), "--steps", "600", "--warmup", "50", "--prewarm", "50", "--no-cpu-baseline", "--no-full-step", "--no-stress", "--operator-path"]
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats("bench.py|rasterizer.py|_lib.py|autograd|function.py", 25)
st.sort_stats("tottime").print_stats(30)
