#!/usr/bin/env python
"""Golden fixture of the reference modules' ``state_dict`` LAYOUT (build container only).

Imports /root/reference/network/styleunet/dual_styleunet.py (stub modules stand in for its two compiled extensions) and
records, for the two DualStyleUNet configurations ``network/avatar.py:34-36`` instantiates (out_ch 3 and 8): every
``state_dict`` key in order with its shape and kind (parameter / buffer), the ``named_parameters`` order (what
``torch.optim.Adam`` indexes its state by, main_avatar.py:55-58,787-813), and the values of the constant buffers (FIR and
Haar kernels) the reference stores in ``net.pt``.  Plus the ``AvatarNet`` level layout: the three networks in
registration order followed by ``viewdir_net.{0,2}.{weight,bias}`` (network/avatar.py:34-50).

    python tests/golden/make_golden_state_layout.py
"""
import json
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.modules.setdefault("fused", types.ModuleType("fused"))
sys.modules.setdefault("upfirdn2d", types.ModuleType("upfirdn2d"))
sys.path.insert(0, "/root/reference")
from network.styleunet.dual_styleunet import DualStyleUNet  # noqa: E402  (reference code)

out = {}
for out_ch in (3, 8):
    torch.manual_seed(0)
    net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=out_ch, out_size=1024, style_dim=512, n_mlp=2)
    params = dict(net.named_parameters())
    sd = net.state_dict()
    consts = {k: v.tolist() for k, v in sd.items() if k.endswith((".kernel", ".ll", ".lh", ".hl", ".hh"))}
    out[f"out_ch_{out_ch}"] = {
        "keys": [[k, list(v.shape), "param" if k in params else "buffer"] for k, v in sd.items()],
        "param_order": list(params.keys()),
        "constants": consts,
        "n_params": sum(p.numel() for p in params.values()),
    }
# AvatarNet registers color_net, position_net, other_net, then (with_viewdirs) viewdir_net = Sequential(Conv2d, LeakyReLU, Conv2d)
vd = torch.nn.Sequential(torch.nn.Conv2d(1, 64, 4, 2, 1), torch.nn.LeakyReLU(0.2, inplace=True), torch.nn.Conv2d(64, 128, 4, 2, 1))
out["avatar_net"] = {"children": ["color_net", "position_net", "other_net", "viewdir_net"],
                     "viewdir_net": [[k, list(v.shape)] for k, v in vd.state_dict().items()]}
with open(os.path.join(HERE, "state_layout.json"), "w") as f:
    json.dump(out, f)
print({k: len(v.get("keys", [])) for k, v in out.items()})
