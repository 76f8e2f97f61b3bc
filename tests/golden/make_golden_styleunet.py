#!/usr/bin/env python
"""Golden vectors of fused_bias_act / upfirdn2d, produced by the REFERENCE'S OWN PYTHON (build container only).

Imports /root/reference/network/styleunet/{fused_act,upfirdn2d}.py with empty stub modules standing in for the two
compiled extensions (the CPU branches fused_act.py:118-129 and upfirdn2d.py:186-227 never touch them) and records
inputs, outputs and autograd gradients for a set of small cases that covers every (up, down, pad, kernel)
combination DualStyleUNet uses plus cropping (negative pad) and ragged sizes.

    python tests/golden/make_golden_styleunet.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.modules.setdefault("fused", types.ModuleType("fused"))
sys.modules.setdefault("upfirdn2d", types.ModuleType("upfirdn2d"))
sys.path.insert(0, "/root/reference")
from network.styleunet import fused_act as ref_fused  # noqa: E402  (reference code)
from network.styleunet import upfirdn2d as ref_up  # noqa: E402  (reference code)

torch.manual_seed(31359)
out = {}

# ---- fused leaky relu (act=3): forward + grads wrt input and bias -------------------------------------------------
for i, shape in enumerate([(2, 5, 6, 7), (1, 8, 16, 16), (3, 4)]):
    x = torch.randn(*shape, requires_grad=True)
    b = torch.randn(shape[1], requires_grad=True)
    y = ref_fused.fused_leaky_relu(x, b, 0.2, 2 ** 0.5)
    g = torch.randn_like(y)
    y.backward(g)
    out.update({f"lrelu{i}_x": x.detach().numpy(), f"lrelu{i}_b": b.detach().numpy(), f"lrelu{i}_y": y.detach().numpy(),
                f"lrelu{i}_g": g.numpy(), f"lrelu{i}_gx": x.grad.numpy(), f"lrelu{i}_gb": b.grad.numpy()})

# ---- upfirdn2d ---------------------------------------------------------------------------------------------------
k4 = torch.tensor([1.0, 3.0, 3.0, 1.0])
k4 = k4[None, :] * k4[:, None]
k4 = k4 / k4.sum()
haar = {"ll": [[0.5, 0.5], [0.5, 0.5]], "lh": [[-0.5, -0.5], [0.5, 0.5]], "hl": [[-0.5, 0.5], [-0.5, 0.5]], "hh": [[0.5, -0.5], [-0.5, 0.5]]}
cases = [
    # name, (N, C, H, W), kernel, up, down, pad
    ("blur_p21", (1, 4, 17, 17), k4, 1, 1, (2, 1)),             # Blur before a stride-2 conv (pad 2,2 / 2,1 variants)
    ("blur_p22", (1, 3, 16, 16), k4, 1, 1, (2, 2)),
    ("blur_up_p11", (1, 4, 17, 17), k4 * 4, 1, 1, (1, 1)),      # Blur after conv_transpose (kernel * up^2)
    ("upsample2", (2, 3, 8, 9), k4 * 4, 2, 1, (2, 1)),           # Upsample(factor 2)
    ("downsample2", (1, 3, 16, 18), k4, 1, 2, (1, 1)),           # Downsample(factor 2)
    ("haar_hl", (1, 6, 12, 10), torch.tensor(haar["hl"]), 1, 2, (0, 0)),
    ("ihaar_lh", (1, 6, 6, 5), torch.tensor(haar["lh"]), 2, 1, (1, 0)),
    ("crop_negpad", (1, 2, 11, 13), k4, 1, 1, (-1, 2)),
    ("asym_3x2", (2, 2, 7, 5), torch.randn(3, 2), 2, 3, (2, 1)),
]
for name, shape, k, up, down, pad in cases:
    x = torch.randn(*shape, requires_grad=True)
    y = ref_up.upfirdn2d(x, k, up=up, down=down, pad=pad)       # CPU tensor -> upfirdn2d_native
    g = torch.randn_like(y)
    y.backward(g)
    out.update({f"up_{name}_x": x.detach().numpy(), f"up_{name}_k": k.numpy(), f"up_{name}_y": y.detach().numpy(),
                f"up_{name}_g": g.numpy(), f"up_{name}_gx": x.grad.numpy(),
                f"up_{name}_cfg": np.array([up, down, pad[0], pad[1]], np.int32)})
np.savez_compressed(os.path.join(HERE, "styleunet_ops.npz"), **out)
print("wrote styleunet_ops.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "styleunet_ops.npz")) // 1024, "KiB")
