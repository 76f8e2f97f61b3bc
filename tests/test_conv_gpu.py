"""MFMA convolutions (forward, input gradient, weight gradient) vs torch fp32 CPU convolution + autograd, in both arithmetic
modes of include/ag_conv.h (three-way bf16 split on the bf16 matrix pipe = the default, and exact-product fp32 MFMA).

Tolerance, the same for both modes: fp32 accumulation in a different order than the CPU reference over up to K = 4608 products:
|diff| <= 1e-5 * (sum-magnitude scale) -- expressed as rtol 1e-4 on values with an absolute floor of 1e-5 * max|ref|.
test_split_products_are_fp32_grade measures the two modes against an fp64 convolution."""
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
    # tile-shape edge cases of the per-tap gather kernels: several image rows per 256-pixel N tile, one row per tile, a ragged last tile
    # (odd height), rows wider than a tile, the 65 / 64-wide parity classes of a transposed convolution and of a stride-2 input gradient,
    # one tap per K tile group (1 x 1), more than 128 output rows (two M tiles), and a K long enough for split-K
    ("tiles_c3x3_64", "conv", 32, 48, 64, 64, 3, 1, 1),
    ("tiles_c3x3_wide", "conv", 16, 130, 24, 256, 3, 1, 1),
    ("tiles_c3x3_odd_h", "conv", 48, 40, 37, 128, 3, 1, 1),
    ("tiles_c3x3_512wide", "conv", 16, 24, 9, 512, 3, 1, 1),
    ("tiles_ct3x3", "convT", 32, 16, 64, 64, 3, 2, 0),
    ("tiles_c3x3_s2_dgrad", "conv", 16, 32, 129, 129, 3, 2, 0),
    ("tiles_c1x1", "conv", 64, 160, 64, 64, 1, 1, 0),
    ("tiles_c3x3_k_split", "conv", 256, 64, 32, 32, 3, 1, 1),
    # the K-vectorised weight-gradient loader (round 4: stride-1 "same" convolutions with rows of a multiple of 16 pixels): a row length that
    # is not a power of two, the narrowest row (every quad of a row is an edge quad or next to one), more than one column per thread
    ("bvec_c3x3_w48", "conv", 24, 40, 20, 48, 3, 1, 1),
    ("bvec_c3x3_w16", "conv", 40, 72, 23, 16, 3, 1, 1),
    ("bvec_c3x3_m64", "conv", 48, 64, 32, 64, 3, 1, 1),
]


@pytest.fixture(params=["split_bf16", "fp32", "split_f16"])
def math_mode(request):
    from animatablegaussians_amd import conv as agc
    prev = agc.set_math(request.param)
    yield request.param
    agc.set_math(prev)


def _close(a, b, name):
    a, b = a.detach().cpu().double().numpy(), b.detach().double().numpy()
    lim = 1e-4 * np.abs(b) + 1e-5 * np.abs(b).max() + 1e-7
    d = np.abs(a - b)
    assert np.isfinite(a).all(), name
    assert (d <= lim).all(), f"{name}: max diff {d.max():.3e} (ref max {np.abs(b).max():.3e}), worst ratio {(d / lim).max():.2f}"


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_forward_backward(case, math_mode):
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
def test_weight_scale_is_the_prescaled_convolution(shape, math_mode):
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


@pytest.mark.parametrize("kind,Cin,Cout,hw,k,stride,padding", [("conv", 512, 512, 16, 3, 1, 1), ("conv", 128, 256, 33, 3, 2, 1),
                                                                ("convT", 256, 128, 12, 3, 2, 0)])
def test_split_products_are_fp32_grade(kind, Cin, Cout, hw, k, stride, padding):
    """The contracts of the split modes (include/ag_conv.h), measured against an fp64 convolution of the same fp32 inputs, forward /
    dL/dx / dL/dw, on inputs with a wide dynamic range (lognormal magnitudes, so the low parts of the split matter):
      split_bf16    every product within 2^-23 |a| |b| of the exact one, fp32 accumulation: (a) the rms deviation is within 1.5x of the
                    fp32-MFMA mode's, whose products are exact -- the accumulation order, not the split, sets the error; (b) no element
                    deviates by more than 2^-19 sum |a| |b| (the bound the fp32-MFMA mode is held to as well: accumulation rounding).
      split_bf16x3  every product within 3 * 2^-16 |a| |b|: no element deviates by more than 3 * 2^-16 sum |a| |b| (+ the accumulation
                    bound); and it is measurably coarser than split_bf16 (the mode switch really changes the arithmetic)."""
    import torch
    import torch.nn.functional as F
    from animatablegaussians_amd import conv as agc
    g = torch.Generator().manual_seed(11)

    def wide(*shape):
        return torch.randn(*shape, generator=g) * torch.exp(1.5 * torch.randn(*shape, generator=g))

    x = wide(1, Cin, hw, hw)
    w = wide(*((Cout, Cin, k, k) if kind == "conv" else (Cin, Cout, k, k))) / np.sqrt(Cin * k * k)
    f = (lambda a, b: F.conv2d(a, b, None, stride=stride, padding=padding)) if kind == "conv" else (lambda a, b: F.conv_transpose2d(a, b, None, stride=2))
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = f(xd, wd)
    gy = wide(*ref.shape)
    ref.backward(gy.double())
    xa, wa = x.abs().double().requires_grad_(True), w.abs().double().requires_grad_(True)      # sum |a| |b| of every output
    mag = f(xa, wa)
    mag.backward(gy.abs().double())
    mags = (mag.detach(), xa.grad, wa.grad)
    refs = (ref.detach(), xd.grad, wd.grad)
    names = ("forward", "dL/dx", "dL/dw")
    rms, worst = {}, {}
    for mode in ("fp32", "split_bf16", "split_f16", "split_bf16x3"):
        prev = agc.set_math(mode)
        try:
            xg, wg = (t.clone().cuda().requires_grad_(True) for t in (x, w))
            fn = agc.conv2d if kind == "conv" else agc.conv_transpose2d
            out = fn(xg, wg, None, stride=stride, padding=padding)
            out.backward(gy.cuda())
            got = (out.detach(), xg.grad, wg.grad)
        finally:
            agc.set_math(prev)
        for name, a, r, m in zip(names, got, refs, mags):
            d = (a.cpu().double() - r).abs()
            worst[(mode, name)] = float((d / (m + 1e-300)).max())
            rms[(mode, name)] = float(torch.sqrt((d * d).mean()))
            print(f"{kind} {Cin}->{Cout} {mode:13s} {name:8s} max dev / sum|a||b| = 2^{np.log2(worst[(mode, name)] + 1e-300):6.2f}   rms dev = {rms[(mode, name)]:.3e}")
    for name in names:
        assert worst[("fp32", name)] <= 2.0 ** -19 and worst[("split_bf16", name)] <= 2.0 ** -19, (name, worst)
        assert rms[("split_bf16", name)] <= 1.5 * rms[("fp32", name)] + 1e-12, (name, rms)
        assert worst[("split_f16", name)] <= 2.0 ** -19 and rms[("split_f16", name)] <= 1.5 * rms[("fp32", name)] + 1e-12, (name, worst, rms)
        assert worst[("split_bf16x3", name)] <= 3 * 2.0 ** -16 + 2.0 ** -19, (name, worst)
        assert rms[("split_bf16x3", name)] > 4.0 * rms[("split_bf16", name)], (name, rms)


def test_split_f16_bound_beyond_the_full_precision_range():
    """include/ag_conv.h, AG_CONV_MATH_SPLIT_F16: |error| <= 3 * 2^-24 sum |a| |b| + 2^-39 (M_b sum |a| + M_a sum |b|) per output (plus the
    fp32 accumulation bound 2^-19 sum |a| |b| every mode is held to).  Operands spread over 40 binades (far more than the 17 below a
    tensor's maximum in which the low parts are normal fp16 numbers), and one output channel whose weights are 2^-24 of the tensor's
    largest: the bound holds everywhere, the outputs fed by in-range operands keep the fp32 grade, and the tiny channel degrades in
    RELATIVE terms only (its absolute error stays 2^-15 of the rounding fp32 commits on the large outputs)."""
    import torch
    import torch.nn.functional as F
    from animatablegaussians_amd import conv as agc
    g = torch.Generator().manual_seed(23)
    Cin, Cout, hw = 64, 128, 24
    x = torch.randn(1, Cin, hw, hw, generator=g) * torch.exp2(torch.randint(-40, 1, (1, Cin, hw, hw), generator=g).float())
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * torch.exp2(torch.randint(-12, 1, (Cout, Cin, 3, 3), generator=g).float())
    w[5] *= 2.0 ** -24
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    mag = F.conv2d(x.abs().double(), w.abs().double(), None, padding=1)                              # sum |a| |b|
    sum_x = F.conv2d(x.abs().double(), torch.ones(1, Cin, 3, 3, dtype=torch.float64), None, padding=1)  # sum |x| over the receptive field
    sum_w = w.abs().double().sum(dim=(1, 2, 3)).view(1, Cout, 1, 1)
    Mx, Mw = float(x.abs().max()), float(w.abs().max())
    prev = agc.set_math("split_f16")
    try:
        out = agc.conv2d(x.cuda(), w.cuda(), None, padding=1).cpu().double()
    finally:
        agc.set_math(prev)
    dev = (out - ref).abs()
    bound = (2.0 ** -19 + 3 * 2.0 ** -24) * mag + 2.0 ** -39 * (Mw * sum_x + Mx * sum_w)
    assert bool((dev <= bound).all()), float((dev / bound).max())
    big = mag >= 2.0 ** -10 * float(mag.max())                 # outputs fed by operands near their tensors' maxima: the fp32 grade
    assert float((dev[big] / mag[big]).max()) <= 2.0 ** -19
    tiny = dev[:, 5] / (mag[:, 5] + 1e-300)                    # the 2^-24 channel: relative error degraded, as documented ...
    assert float(dev[:, 5].max()) <= 2.0 ** -38 * Mw * float(sum_x.max())      # ... absolute error far below fp32's rounding of the large outputs
    print(f"split_f16 wide range: worst dev / bound {float((dev / bound).max()):.3f}; tiny channel worst relative error 2^{np.log2(float(tiny.max()) + 1e-300):.1f}")


def test_three_product_mode_vs_torch():
    """split_bf16x3 (opt-in) on the reference comparison of test_conv_forward_backward, at ITS contract: deviation from the fp32 CPU
    convolution <= 1e-4 |ref| + 1e-4 max|ref| (products within 3 * 2^-16)."""
    import torch
    import torch.nn.functional as F
    from animatablegaussians_amd import conv as agc
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 64, 40, 36, generator=g, requires_grad=True)
    w = (torch.randn(96, 64, 3, 3, generator=g) / 24.0).requires_grad_(True)
    ref = F.conv2d(x, w, None, padding=1)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    prev = agc.set_math("split_bf16x3")
    try:
        assert agc.get_math() == "split_bf16x3"
        xg, wg = (t.detach().cuda().requires_grad_(True) for t in (x, w))
        out = agc.conv2d(xg, wg, None, padding=1)
        out.backward(gy.cuda())
    finally:
        agc.set_math(prev)
    for name, a, b in (("forward", out, ref), ("dL/dx", xg.grad, x.grad), ("dL/dw", wg.grad, w.grad)):
        a, b = a.detach().cpu().double().numpy(), b.detach().double().numpy()
        d = np.abs(a - b)
        assert (d <= 1e-4 * np.abs(b) + 1e-4 * np.abs(b).max()).all(), (name, d.max(), np.abs(b).max())
    with pytest.raises(ValueError):
        agc.set_math("tf32")      # the opt-in 11-bit mode is spelled "f16"


def _round11(t, mult=1.0):
    """Oracle of AG_CONV_MATH_F16's operand rounding (include/ag_conv.h): the tensor scaled by the power of two that puts its largest
    magnitude (times ``mult``) into [2^14, 2^15), rounded to fp16 (nearest even), scaled back -- 11 significant bits wherever the scaled
    value is a normal fp16 number.  Returns float64."""
    import torch
    m = float(t.abs().max()) * mult
    e = int(np.floor(np.log2(m))) if m > 0 else 0
    s = 2.0 ** (14 - e)
    return (t.double() * s).to(torch.float16).double() / s


def _tf32_rne(t):
    """A float32 tensor rounded to TF32's 10 explicit mantissa bits, nearest even (cuDNN converts with ties away: differs on exact ties only)."""
    import torch
    b = t.contiguous().view(torch.int32)
    b = (b + 0xFFF + ((b >> 13) & 1)) & ~0x1FFF
    return b.view(torch.float32)


@pytest.mark.parametrize("kind,Cin,Cout,H,W,k,stride,padding", [("conv", 64, 96, 40, 36, 3, 1, 1), ("conv", 256, 64, 32, 32, 3, 1, 1), ("conv", 48, 130, 33, 35, 3, 2, 0),
                                                               ("convT", 128, 64, 16, 16, 3, 2, 0), ("conv", 64, 160, 64, 64, 1, 1, 0), ("conv", 512, 512, 16, 16, 3, 1, 1)])
def test_f16_mode_is_the_convolution_of_operands_rounded_to_11_bits(kind, Cin, Cout, H, W, k, stride, padding):
    """AG_CONV_MATH_F16 (opt-in, round 5) against ITS contract: forward, input gradient and weight gradient equal the fp64 convolution of the
    operands ROUNDED to 11 significant bits (exact products), up to the fp32 accumulation bound 2^-19 sum |a| |b| every mode is held to --
    i.e. the only error of the mode is the operand rounding, which is TF32's (asserted bit for bit on the in-range values).  Against the
    UNROUNDED fp64 convolution the deviation is then the rounding's: <= (2^-11 + 2^-19) sum |a| |b|."""
    import torch
    import torch.nn.functional as F
    from animatablegaussians_amd import conv as agc
    g = torch.Generator().manual_seed(77)
    x = torch.randn(1, Cin, H, W, generator=g) * torch.exp2(torch.randint(-6, 1, (1, Cin, H, W), generator=g).float())
    w = torch.randn(*((Cout, Cin, k, k) if kind == "conv" else (Cin, Cout, k, k)), generator=g) / np.sqrt(Cin * k * k)
    f = (lambda a, b: F.conv2d(a, b, None, stride=stride, padding=padding)) if kind == "conv" else (lambda a, b: F.conv_transpose2d(a, b, None, stride=2))
    gy = torch.randn(f(x, w).shape, generator=g) * torch.exp2(torch.randint(-6, 1, f(x, w).shape, generator=g).float())
    xr, wr, gr = _round11(x), _round11(w), _round11(gy)
    # the rounding IS TF32's on every value whose scaled magnitude is a normal fp16 number
    for t, r in ((x, xr), (w, wr), (gy, gr)):
        inr = t.abs() >= 2.0 ** -28 * float(t.abs().max())
        assert torch.equal(r.float()[inr], _tf32_rne(t)[inr])

    def grads(xo, wo, go):
        xo, wo = xo.clone().requires_grad_(True), wo.clone().requires_grad_(True)
        out = f(xo, wo)
        out.backward(go)
        return out.detach(), xo.grad, wo.grad
    # forward: round(x) * round(w); dL/dx: round(gy) * round(w); dL/dw: round(gy) * round(x)
    ref_f = f(xr, wr)
    ref_dx = grads(x.double(), wr, gr)[1]
    ref_dw = grads(xr, w.double(), gr)[2]
    exact = grads(x.double(), w.double(), gy.double())
    mags = grads(x.abs().double(), w.abs().double(), gy.abs().double())
    prev = agc.set_math("f16")
    try:
        assert agc.get_math() == "f16" and agc.needs_maxima()
        xg, wg = (t.clone().cuda().requires_grad_(True) for t in (x, w))
        fn = agc.conv2d if kind == "conv" else agc.conv_transpose2d
        out = fn(xg, wg, None, stride=stride, padding=padding)
        out.backward(gy.cuda())
        got = (out.detach(), xg.grad, wg.grad)
    finally:
        agc.set_math(prev)
    for name, a, r, ex, m in zip(("forward", "dL/dx", "dL/dw"), got, (ref_f, ref_dx, ref_dw), exact, mags):
        a = a.cpu().double()
        assert bool(torch.isfinite(a).all()), name
        d = (a - r).abs()
        worst = float((d / (m + 1e-300)).max())
        dev_exact = float(((a - ex).abs() / (m + 1e-300)).max())
        print(f"{kind} {Cin}->{Cout} f16 {name:8s}: vs rounded-operand oracle 2^{np.log2(worst + 1e-300):6.2f} of sum|a||b|; vs exact fp64 2^{np.log2(dev_exact + 1e-300):6.2f}")
        assert worst <= 2.0 ** -19, (name, worst)
        assert dev_exact <= 2.0 ** -11 + 2.0 ** -19, (name, dev_exact)


@pytest.mark.parametrize("kind,Cin,Cout,hw,k,stride,padding", [("conv", 64, 64, 128, 3, 1, 1), ("conv", 32, 48, 97, 3, 2, 0), ("convT", 64, 32, 48, 3, 2, 0)])
def test_weight_gradient_is_deterministic_and_needs_no_zeroed_target(kind, Cin, Cout, hw, k, stride, padding, math_mode):
    """Round 5: many pixel slices (the 64-channel layers at 512^2 use 28), summed in slice order by wgrad_reduce_kernel: the same bits on every
    call, and the target is overwritten, not accumulated into (the call used to zero it and add with float atomics)."""
    import torch
    from animatablegaussians_amd import conv as agc
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, Cin, hw, hw, generator=g).cuda()
    w = (torch.randn(*((Cout, Cin, k, k) if kind == "conv" else (Cin, Cout, k, k)), generator=g) / np.sqrt(Cin * k * k)).cuda()
    fn = agc.conv2d if kind == "conv" else agc.conv_transpose2d
    grads = []
    for rep in range(3):
        xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        out = fn(xg, wg, None, stride=stride, padding=padding)
        if rep == 0:
            gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(10)).cuda()
        out.backward(gy)
        grads.append(wg.grad.clone())
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])


@pytest.mark.parametrize("dma", ["1", "2"])
def test_the_opt_in_dma_loader_gives_the_register_staged_loaders_results(dma):
    """Round 6: the LDS-DMA loader over pre-split fp16 planes (`AG_CONV_DMA`, csrc/ag_conv.hip gather_conv_dma_kernel; opt-in, profiles/r06_conv_dma.md) is
    a process-wide switch read once, so it runs in a child process: the convolutions of the fp16 forms -- forward and input gradient, plain 3 x 3, stride
    2, the transposed convolution's parity classes, grouped instances, 128- and 256-row tiles -- must equal the default loader's results computed HERE
    to the last bit of the split (same products, same K order inside a tile; the tile shape changes the order of the fp32 accumulation: 2e-6 of scale)."""
    import json
    import os
    import subprocess
    import sys
    import torch
    from animatablegaussians_amd import conv as agc
    code = r'''
import sys, json, torch
sys.path.insert(0, %r)
from animatablegaussians_amd import conv as agc
agc.set_math(sys.argv[1])
out = {}
for name, (cin, cout, h, w, k, s, p, tr) in json.loads(sys.argv[2]).items():
    g = torch.Generator().manual_seed(sum(map(ord, name)))          # (hash() of a str differs between processes)
    x = torch.randn(1, cin, h, w, generator=g).cuda().requires_grad_(True)
    wt = (torch.randn(cin, cout, k, k, generator=g) if tr else torch.randn(cout, cin, k, k, generator=g)).cuda()
    y = agc.conv_transpose2d(x, wt, stride=s, padding=p) if tr else agc.conv2d(x, wt, stride=s, padding=p)
    gy = torch.randn(y.shape, generator=g).cuda()
    gx, = torch.autograd.grad(y, x, gy)
    out[name] = [float(y.double().sum()), float(y.double().abs().sum()), float(gx.double().sum()), float(gx.double().abs().sum()),
                 y.flatten()[::max(1, y.numel() // 64)][:64].tolist(), gx.flatten()[::max(1, gx.numel() // 64)][:64].tolist()]
agc.check_status()
print("RESULT" + json.dumps(out))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = {"c256_64": (256, 256, 64, 64, 3, 1, 1, False), "c128_96": (128, 128, 96, 96, 3, 1, 1, False), "c64_s2": (64, 128, 64, 64, 3, 2, 1, False),
             "c512_32": (512, 256, 32, 32, 3, 1, 1, False), "t64_up": (64, 32, 24, 24, 3, 2, 0, True), "c48": (48, 80, 40, 36, 3, 1, 1, False),
             # large enough for pick_tile to choose the 256 x 256 and the 128 x 256 tile (AG_CONV_DMA=1 uses the DMA loader only there)
             "c256_256big": (256, 256, 256, 256, 3, 1, 1, False), "c128_384big": (128, 128, 384, 384, 3, 1, 1, False)}
    res = {}
    for label, env in (("base", {"AG_CONV_DMA": "0"}), ("dma", {"AG_CONV_DMA": dma})):
        p = subprocess.run([sys.executable, "-c", code, "split_f16", json.dumps(cases)], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr[-2000:]
        res[label] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][-1][6:])
    for name in cases:
        a, b = res["base"][name], res["dma"][name]
        for i in (0, 2):
            assert abs(a[i] - b[i]) <= 2e-6 * max(a[i + 1], 1e-30), (name, i, a[i], b[i])
        for i in (4, 5):
            sc = max(abs(v) for v in a[i]) + 1e-30
            assert max(abs(u - v) for u, v in zip(a[i], b[i])) <= 2e-6 * sc, (name, i)
