"""How many grouped layer / comb calls of one AvatarNet.get_maps get their input maxima from the producer (21 of 34 at HEAD):  python profiles/handover_count.py"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from animatablegaussians_amd import grouped as gr
from animatablegaussians_amd.avatar import AvatarNet
n = {"hit": 0, "miss": 0}
orig = gr._handed_maxima
def counted(x):
    m = orig(x)
    n["hit" if m is not None else "miss"] += 1
    return m
gr._handed_maxima = counted
dev = torch.device("cuda:0")
net = AvatarNet.synthetic({'with_viewdirs': True}, device=dev)
pose = torch.randn(3, 512, 512, device=dev); vf = torch.randn(1, 128, 128, 128, device=dev)
with torch.no_grad():
    net.get_maps(pose, vf, vf)
print("handed-over inputs:", n)
