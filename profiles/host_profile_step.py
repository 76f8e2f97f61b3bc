"""cProfile of the calling thread during training steps (forward glue, loss, optimizer; the backward's Python runs on autograd's thread)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_avatar
dev = torch.device("cuda:0")
step = bench_avatar.TrainingStep(dev)
for i in range(3):
    step(i, 1)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(5):
    step(i, 1)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
