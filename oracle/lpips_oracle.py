"""CPU oracle (torch fp32 + autograd) of LPIPS(net='vgg').  TEST INFRASTRUCTURE ONLY.

Restates network/lpips/lpips.py:84-127 (forward, spatial = False, lpips = True), :129-137 (ScalingLayer),
network/lpips/__init__.py:40-42 (normalize_tensor), network/lpips/pretrained_networks.py:97-134 (the five slices of
torchvision's vgg16().features) on plain torch ops.  torchvision is third-party and absent from /root/reference and this image
(requirements.txt pins torchvision==0.15.2): the `features` stack is its published configuration 'D'
[64, 64, M, 128, 128, M, 256, 256, 256, M, 512, 512, 512, M, 512, 512, 512, M], Conv2d(3x3, padding 1) + ReLU(inplace), MaxPool2d(2, 2).
Pinned by tests/golden/lpips_vgg_64.npz, which tests/golden/make_golden_lpips.py produced by running the REFERENCE'S OWN LPIPS class
on CPU with a stand-in `torchvision.models.vgg16` of that configuration and name-seeded weights; trunk weights: parity unpinned.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

SLICES = ((0, 2), (5, 7), (10, 12, 14), (17, 19, 21), (24, 26, 28))
SHIFT = torch.tensor([-.030, -.088, -.188])[None, :, None, None]
SCALE = torch.tensor([.458, .448, .450])[None, :, None, None]


def vgg_taps(x, sd):
    taps, h = [], x
    for si, idxs in enumerate(SLICES):
        if si:
            h = F.max_pool2d(h, 2, 2)
        for i in idxs:
            h = F.relu(F.conv2d(h, sd[f"net.slice{si + 1}.{i}.weight"], sd[f"net.slice{si + 1}.{i}.bias"], padding=1))
        taps.append(h)
    return taps


def normalize_tensor(f, eps=1e-10):
    return f / (torch.sqrt(torch.sum(f ** 2, dim=1, keepdim=True) + eps) + eps)


def lpips(in0, in1, sd, normalize=False):
    if normalize:
        in0, in1 = 2 * in0 - 1, 2 * in1 - 1
    o0, o1 = vgg_taps((in0 - SHIFT) / SCALE, sd), vgg_taps((in1 - SHIFT) / SCALE, sd)
    res = []
    for k in range(5):
        d = (normalize_tensor(o0[k]) - normalize_tensor(o1[k])) ** 2
        res.append(F.conv2d(d, sd[f"lin{k}.model.1.weight"]).mean([2, 3], keepdim=True))
    val = res[0]
    for r in res[1:]:
        val = val + r
    return val, res
