"""cProfile of the host side of the raster step (bench.py's loop): where the Python time per view goes."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [os.path.join(ROOT, "bench.py"), "--steps", "400", "--warmup", "50", "--no-cpu-baseline", "--streams", "2"]
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats("bench.py|rasterizer.py|_lib.py|autograd|function.py", 30)
st.sort_stats("tottime").print_stats(28)
