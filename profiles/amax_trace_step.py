import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench_avatar
dev = torch.device("cuda:0")
step = bench_avatar.TrainingStep(dev)
for i in range(3): step(i, 1)
torch.cuda.synchronize()
sys.stderr.write("=====STEP=====\n"); sys.stderr.flush()
step(3, 1); torch.cuda.synchronize()
sys.stderr.write("=====END=====\n")
