#!/usr/bin/env python
"""Golden fixture of LPIPS(net='vgg') FROM THE REFERENCE'S OWN CLASS (build container only).

Imports /root/reference/network/lpips (lpips.py, __init__.py, pretrained_networks.py) with a stand-in `torchvision` module whose
`models.vgg16(pretrained=...)` returns the published VGG16 `features` stack (configuration 'D') -- torchvision itself is not
installed here and its ImageNet weights cannot be fetched -- fills every parameter with `synth.named_fill`-style name-seeded values
(trunk: He-scaled normal, small biases; lin: non-negative), and records value, per-level values and the gradient w.r.t. the first
image for two random 64x64 images, with and without `normalize`.

    python tests/golden/make_golden_lpips.py
"""
import os
import sys
import types
import zlib

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def vgg16_features():
    cfg, layers, cin = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M'], [], 3
    for v in cfg:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


tv = types.ModuleType("torchvision")
tv.models = types.ModuleType("torchvision.models")
tv.models.vgg16 = lambda pretrained=False: types.SimpleNamespace(features=vgg16_features())
sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, tv.models
sys.path.insert(0, "/root/reference")
from network.lpips import LPIPS  # noqa: E402  (reference code)

from animatablegaussians_amd.lpips import lpips_named_fill  # noqa: E402

ref = LPIPS(net='vgg', pretrained=False, pnet_rand=True, verbose=False)
# the reference registers its lin layers twice (lin0.. and the ModuleList lins.0..: the same tensors under two names)
sd = lpips_named_fill({k: v for k, v in ref.state_dict().items() if not k.startswith(("scaling_layer", "lins."))})
missing, unexpected = ref.load_state_dict(sd, strict=False)
assert not unexpected and all(m.startswith(("scaling_layer", "lins.")) for m in missing), (missing, unexpected)
assert not ref.training

g = torch.Generator().manual_seed(99)
out = {}
for tag, normalize in (("n", True), ("r", False)):
    a = torch.rand(1, 3, 64, 64, generator=g) if normalize else torch.rand(1, 3, 64, 64, generator=g) * 2 - 1
    b = (a + 0.3 * torch.randn(1, 3, 64, 64, generator=g)).clamp(0 if normalize else -1, 1)
    a = a.requires_grad_(True)
    val, res = ref.forward(a, b, retPerLayer=True, normalize=normalize)
    val.sum().backward()
    out.update({f"{tag}_in0": a.detach().numpy(), f"{tag}_in1": b.numpy(), f"{tag}_val": val.detach().numpy(),
                f"{tag}_res": np.array([float(r.detach()) for r in res], np.float32), f"{tag}_grad": a.grad.numpy()})
np.savez_compressed(os.path.join(HERE, "lpips_vgg_64.npz"), **out)
print("wrote lpips_vgg_64.npz; values", float(out["n_val"].item()), float(out["r_val"].item()), "per level", out["n_res"])
