"""Two evaluations of AvatarNet.get_maps on the same inputs, per conv mode: bit-equal?  (debug probe)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from animatablegaussians_amd import conv as agc
from animatablegaussians_amd.avatar import AvatarNet
import test_avatar_net_gpu as T
torch.manual_seed(31359)
net = AvatarNet.synthetic({'with_viewdirs': True})
items = T._items(net)
net.get_pose_map(items)
net.eval()
res = {}
with torch.no_grad():
    for mode in ("fp32", "split_bf16", "fp32", "split_bf16"):
        agc.set_math(mode)
        fv, bv = net.get_viewdir_feat(items)
        a = [t.clone() for t in net.get_maps(items['smpl_pos_map'][:3], fv, bv)]
        fv2, bv2 = net.get_viewdir_feat(items)
        b = [t.clone() for t in net.get_maps(items['smpl_pos_map'][:3], fv2, bv2)]
        print(mode, "viewdir equal", torch.equal(fv, fv2), torch.equal(bv, bv2), "maps equal", [torch.equal(x, y) for x, y in zip(a, b)],
              "max diff", [float((x - y).abs().max()) for x, y in zip(a, b)])
        if mode in res:
            print("   vs earlier run of this mode:", [float((x - y).abs().max()) for x, y in zip(a, res[mode])])
        res[mode] = a
    print("fp32 vs split:", [float((x - y).abs().max()) for x, y in zip(res["fp32"], res["split_bf16"])], "scale", [float(x.abs().max()) for x in res["fp32"]])
