"""MFMA convolutions (forward, input gradient, weight gradient) vs torch fp32 CPU convolution + autograd.

Tolerance: exact-fp32 MFMA with a different summation order than the CPU reference over up to K = 4608 products:
|diff| <= 1e-5 * (sum-magnitude scale) -- expressed as rtol 1e-4 on values with an absolute floor of 1e-5 * max|ref|."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    # name, kind, Cin, Cout, H, W, k, stride, padding
    ("c3x3_s1_p1", "conv", 64, 64, 40, 36, 3, 1, 1),
    ("c3x3_s1_p1_wide", "conv", 192, 130, 17, 23, 3, 1, 1),
    ("c3x3_s2_p0", "conv", 32, 48, 35, 33, 3, 2, 0),
    ("c1x1_rgb_in", "conv", 3, 128, 31, 29, 1, 1, 0),
    ("c1x1_torgb", "conv", 64, 12, 24, 24, 1, 1, 0),
    ("c4x4_s2_p1", "conv", 1, 64, 32, 32, 4, 2, 1),
    ("c3x3_s2_p1", "conv", 16, 16, 20, 21, 3, 2, 1),
    ("ct3x3_s2", "convT", 16, 32, 9, 7, 3, 2, 0),
    ("ct3x3_s2_big", "convT", 128, 64, 16, 16, 3, 2, 0),
    ("c3x3_512", "conv", 512, 512, 16, 16, 3, 1, 1),
    # 1 x 1 convolutions that run as streaming VALU kernels (ag_conv_pointwise.hip): ToRGB heads (12 / 32 output rows) and FromRGB
    # (3 input channels), pixel counts that are multiples of 4 (the MFMA path keeps the rest: c1x1_rgb_in above)
    ("pw_torgb12_small", "conv", 512, 12, 16, 16, 1, 1, 0),
    ("pw_torgb12_slices", "conv", 64, 12, 160, 132, 1, 1, 0),
    ("pw_torgb32", "conv", 256, 32, 128, 130, 1, 1, 0),
    ("pw_torgb16_odd_channels", "conv", 70, 16, 144, 116, 1, 1, 0),
    ("pw_fromrgb", "conv", 3, 128, 32, 32, 1, 1, 0),
    ("pw_fromrgb_512", "conv", 3, 512, 16, 16, 1, 1, 0),
]


def _close(a, b, name):
    a, b = a.detach().cpu().double().numpy(), b.detach().double().numpy()
    lim = 1e-4 * np.abs(b) + 1e-5 * np.abs(b).max() + 1e-7
    d = np.abs(a - b)
    assert np.isfinite(a).all(), name
    assert (d <= lim).all(), f"{name}: max diff {d.max():.3e} (ref max {np.abs(b).max():.3e}), worst ratio {(d / lim).max():.2f}"


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_forward_backward(case):
    import torch
    import torch.nn.functional as F
    from animatablegaussians_amd import conv as agc
    name, kind, Cin, Cout, H, W, k, stride, padding = case
    g = torch.Generator().manual_seed(hash(name) % 1000)
    x = torch.randn(1, Cin, H, W, generator=g, requires_grad=True)
    wshape = (Cout, Cin, k, k) if kind == "conv" else (Cin, Cout, k, k)
    w = (torch.randn(*wshape, generator=g) / np.sqrt(Cin * k * k)).requires_grad_(True)
    b = torch.randn(Cout, generator=g, requires_grad=True)
    ref = F.conv2d(x, w, b, stride=stride, padding=padding) if kind == "conv" else F.conv_transpose2d(x, w, b, stride=2)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    xg, wg, bg = (t.detach().cuda().requires_grad_(True) for t in (x, w, b))
    fn = agc.conv2d if kind == "conv" else agc.conv_transpose2d
    out = fn(xg, wg, bg, stride=stride, padding=padding)
    assert out.shape == ref.shape
    _close(out, ref, name + " forward")
    out.backward(gy.cuda())
    _close(xg.grad, x.grad, name + " dL/dx")
    _close(wg.grad, w.grad, name + " dL/dw")
    _close(bg.grad, b.grad, name + " dL/db")


@pytest.mark.parametrize("shape", [(64, 64, 20, 3, 1, 1), (3, 128, 16, 1, 1, 0), (128, 12, 16, 1, 1, 0), (32, 48, 19, 3, 2, 0)])
def test_weight_scale_is_the_prescaled_convolution(shape):
    """conv2d(x, w, weight_scale=s) == conv2d(x, w * s) (EqualConv2d, dual_styleunet.py:100-117), with the weight gradient taken
    w.r.t. the UN-scaled w and an out_scale / bias epilogue on top -- MFMA path and the pointwise kernels."""
    import torch
    import torch.nn.functional as F
    from animatablegaussians_amd import conv as agc
    Cin, Cout, hw, k, stride, padding = shape
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, Cin, hw, hw, generator=g, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, generator=g, requires_grad=True)
    b = torch.randn(Cout, generator=g)
    sc = torch.rand(Cout, generator=g) + 0.5
    s = 1.0 / np.sqrt(Cin * k * k)
    ref = F.conv2d(x, w * s, None, stride=stride, padding=padding) * sc.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    xg, wg = (t.detach().cuda().requires_grad_(True) for t in (x, w))
    out = agc.conv2d(xg, wg, b.cuda(), stride=stride, padding=padding, out_scale=sc.cuda(), weight_scale=s)
    _close(out, ref, "forward")
    out.backward(gy.cuda())
    _close(xg.grad, x.grad, "dL/dx")
    _close(wg.grad, w.grad, "dL/dw")


def test_conv_out_scale_epilogue_and_errors():
    import torch
    import torch.nn.functional as F
    from animatablegaussians_amd import conv as agc
    g = torch.Generator().manual_seed(1)
    x, w = torch.randn(1, 32, 20, 20, generator=g), torch.randn(48, 32, 3, 3, generator=g) * 0.1
    s, b = torch.rand(48, generator=g) + 0.5, torch.randn(48, generator=g)
    ref = F.conv2d(x, w, None, padding=1) * s.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    out = agc.conv2d(x.cuda(), w.cuda(), b.cuda(), padding=1, out_scale=s.cuda())
    _close(out, ref, "scaled epilogue")
    with pytest.raises(RuntimeError):
        agc.conv2d(x.cuda(), w.cuda(), None, stride=3)
    with pytest.raises(RuntimeError):
        agc.conv2d(torch.cat([x, x]).cuda(), w.cuda())
