"""fused_bias_act / upfirdn2d: oracle vs reference-generated goldens (CPU), HIP vs goldens and oracle (GPU).

Tolerance: these are <= 16-tap fp32 FIR sums and single fp32 multiplies: 1e-5 relative + 1e-6 absolute."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "styleunet_ops.npz")
UP_CASES = ["blur_p21", "blur_p22", "blur_up_p11", "upsample2", "downsample2", "haar_hl", "ihaar_lh", "crop_negpad", "asym_3x2"]


def _up_args(cfg, k):
    up, down, p0, p1 = (int(v) for v in cfg)
    return dict(up_x=up, up_y=up, down_x=down, down_y=down, pad_x0=p0, pad_x1=p1, pad_y0=p0, pad_y1=p1)


def _grad_args(cfg, k, in_hw, out_hw):
    """Backward of upfirdn2d = upfirdn2d with swapped up/down, flipped kernel and g_pad (upfirdn2d.py:117-135)."""
    up, down, p0, p1 = (int(v) for v in cfg)
    kh, kw = k.shape
    (in_h, in_w), (out_h, out_w) = in_hw, out_hw
    return dict(up_x=down, up_y=down, down_x=up, down_y=up, pad_x0=kw - p0 - 1, pad_x1=in_w * up - out_w * down + p0 - up + 1,
                pad_y0=kh - p0 - 1, pad_y1=in_h * up - out_h * down + p0 - up + 1)


def test_oracle_matches_reference_goldens():
    import torch
    from oracle import styleunet_oracle as so
    z = np.load(GOLD)
    for i in range(3):
        x, b, g = (torch.from_numpy(z[f"lrelu{i}_{k}"]) for k in ("x", "b", "g"))
        y = so.fused_bias_act(x, b, None, 3, 0, 0.2, 2 ** 0.5)
        np.testing.assert_allclose(y.numpy(), z[f"lrelu{i}_y"], rtol=1e-6, atol=1e-7)
        gx = so.fused_bias_act(g, None, y, 3, 1, 0.2, 2 ** 0.5)       # fused_act.py:41-43
        np.testing.assert_allclose(gx.numpy(), z[f"lrelu{i}_gx"], rtol=1e-6, atol=1e-7)
        dims = [0] + list(range(2, gx.dim()))
        np.testing.assert_allclose(gx.sum(dims).numpy(), z[f"lrelu{i}_gb"], rtol=1e-5, atol=1e-5)
    for name in UP_CASES:
        x, k, g = (torch.from_numpy(z[f"up_{name}_{s}"]) for s in ("x", "k", "g"))
        cfg = z[f"up_{name}_cfg"]
        N, C, H, W = x.shape
        y = so.upfirdn2d(x.reshape(-1, H, W), k, **_up_args(cfg, k))
        ref = z[f"up_{name}_y"]
        assert tuple(y.shape) == (N * C,) + ref.shape[2:], name
        np.testing.assert_allclose(y.numpy().reshape(ref.shape), ref, rtol=1e-5, atol=1e-6, err_msg=name)
        gx = so.upfirdn2d(g.reshape(-1, *g.shape[2:]), torch.flip(k, [0, 1]), **_grad_args(cfg, k, (H, W), ref.shape[2:]))
        np.testing.assert_allclose(gx.numpy().reshape(x.shape), z[f"up_{name}_gx"], rtol=1e-5, atol=1e-6, err_msg=name + " grad")


@pytest.mark.gpu
def test_hip_ops_match_reference_goldens():
    import torch
    from animatablegaussians_amd import styleunet_ops as ops
    z = np.load(GOLD)
    empty = torch.empty(0, device="cuda")
    for i in range(3):
        x, b, g = (torch.from_numpy(z[f"lrelu{i}_{k}"]).cuda() for k in ("x", "b", "g"))
        y = ops.fused_bias_act(x, b, empty, 3, 0, 0.2, 2 ** 0.5)
        np.testing.assert_allclose(y.cpu().numpy(), z[f"lrelu{i}_y"], rtol=1e-6, atol=1e-7)
        gx = ops.fused_bias_act(g, empty, y, 3, 1, 0.2, 2 ** 0.5)
        np.testing.assert_allclose(gx.cpu().numpy(), z[f"lrelu{i}_gx"], rtol=1e-6, atol=1e-7)
    for name in UP_CASES:
        x, k, g = (torch.from_numpy(z[f"up_{name}_{s}"]).cuda() for s in ("x", "k", "g"))
        cfg = z[f"up_{name}_cfg"]
        N, C, H, W = x.shape
        a = _up_args(cfg, k)
        y = ops.upfirdn2d(x.reshape(-1, H, W, 1), k, a["up_x"], a["up_y"], a["down_x"], a["down_y"], a["pad_x0"], a["pad_x1"], a["pad_y0"], a["pad_y1"])
        ref = z[f"up_{name}_y"]
        np.testing.assert_allclose(y.cpu().numpy().reshape(ref.shape), ref, rtol=1e-5, atol=1e-6, err_msg=name)
        b = _grad_args(cfg, k, (H, W), ref.shape[2:])
        gx = ops.upfirdn2d(g.reshape(-1, *g.shape[2:], 1), torch.flip(k, [0, 1]).contiguous(), b["up_x"], b["up_y"], b["down_x"], b["down_y"],
                           b["pad_x0"], b["pad_x1"], b["pad_y0"], b["pad_y1"])
        np.testing.assert_allclose(gx.cpu().numpy().reshape(x.shape), z[f"up_{name}_gx"], rtol=1e-5, atol=1e-6, err_msg=name + " grad")


@pytest.mark.gpu
def test_hip_ops_at_styleunet_sizes_vs_oracle():
    """The largest activation of a DualStyleUNet ([1,64,512,512]) and all (act, grad) switch cases, odd sizes/tails."""
    import torch
    from animatablegaussians_amd import styleunet_ops as ops
    from oracle import styleunet_oracle as so
    g = torch.Generator().manual_seed(5)
    for shape in [(1, 64, 512, 512), (2, 7, 33, 31), (5, 3)]:
        x = torch.randn(*shape, generator=g)
        b = torch.randn(shape[1], generator=g)
        r = torch.randn(*shape, generator=g)
        for act, grad in [(3, 0), (3, 1), (3, 2), (1, 0), (1, 1), (1, 2)]:
            for use_b in (True, False):
                got = ops.fused_bias_act(x.cuda(), b.cuda() if use_b else torch.empty(0, device="cuda"), r.cuda(), act, grad, 0.2, 1.25)
                want = so.fused_bias_act(x, b if use_b else None, r, act, grad, 0.2, 1.25)
                np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-6, atol=1e-7)
    k4 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    k4 = (k4[None] * k4[:, None]) / 64 * 4
    x = torch.randn(128, 257, 257, generator=g)     # Blur after the 128->256 conv_transpose
    got = ops.upfirdn2d(x.cuda()[..., None].contiguous(), k4.cuda(), 1, 1, 1, 1, 1, 1, 1, 1)
    want = so.upfirdn2d(x, k4, 1, 1, 1, 1, 1, 1, 1, 1)
    np.testing.assert_allclose(got.cpu().numpy()[..., 0], want.numpy(), rtol=1e-5, atol=1e-6)
    x = torch.randn(12, 512, 512, generator=g)      # wavelet-skip upsample to 1024^2
    got = ops.upfirdn2d(x.cuda()[..., None].contiguous(), k4.cuda(), 2, 2, 1, 1, 2, 1, 2, 1)
    want = so.upfirdn2d(x, k4, 2, 2, 1, 1, 2, 1, 2, 1)
    assert got.shape[1:3] == (1024, 1024)
    np.testing.assert_allclose(got.cpu().numpy()[..., 0], want.numpy(), rtol=1e-5, atol=1e-6)
    # the 4 x 4, up = down = 1 fast path (fir4x4_kernel): an ASYMMETRIC kernel (a flipped or transposed tap table must show), ragged
    # sizes (widths that are not multiples of 4, single rows), every pad combination the network and its adjoints use and more
    ka = torch.randn(4, 4, generator=g)
    for (c, h, w) in [(3, 17, 19), (2, 5, 4), (1, 1, 9), (4, 64, 66), (2, 33, 7)]:
        x = torch.randn(c, h, w, generator=g)
        for pads in [(1, 1, 1, 1), (2, 2, 2, 2), (2, 1, 2, 1), (0, 3, 3, 0), (3, 3, 3, 3), (1, 2, 2, 1)]:
            if h + pads[2] + pads[3] < 4 or w + pads[0] + pads[1] < 4:
                continue
            got = ops.upfirdn2d(x.cuda()[..., None].contiguous(), ka.cuda(), 1, 1, 1, 1, *pads)
            want = so.upfirdn2d(x, ka, 1, 1, 1, 1, *pads)
            assert got.shape[:3] == want.shape, (got.shape, want.shape)
            np.testing.assert_allclose(got.cpu().numpy()[..., 0], want.numpy(), rtol=1e-5, atol=1e-5)
    with pytest.raises(RuntimeError):
        ops.upfirdn2d(torch.zeros(1, 2, 2, 1, device="cuda"), torch.zeros(4, 4, device="cuda"), 1, 1, 1, 1, 0, 0, 0, 0)   # empty output


@pytest.mark.gpu
@pytest.mark.parametrize("shape,with_noise,with_bias", [((1, 5, 33, 31), True, True), ((1, 8, 64, 64), True, True),
                                                         ((1, 3, 16, 20), False, True), ((1, 4, 72, 72), True, False)])
def test_noise_bias_act_matches_the_unfused_oracle(shape, with_noise, with_bias):
    """StyledConv tail (NoiseInjection + FusedLeakyReLU, dual_styleunet.py:303-313,598-604) fused into one kernel each way,
    vs `x + w * noise` followed by the fused_bias_act oracle on CPU with autograd.  Odd pixel counts take the scalar path."""
    import torch
    from animatablegaussians_amd.styleunet_ops import noise_bias_act
    from oracle import styleunet_oracle as so
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    noise = torch.randn(1, 1, shape[2], shape[3], generator=g) if with_noise else None
    nw = torch.randn(1, generator=g)
    b = torch.randn(shape[1], generator=g) if with_bias else None
    gy = torch.randn(*shape, generator=g)
    xc = x.clone().requires_grad_(True)
    nwc = nw.clone().requires_grad_(True)
    bc = b.clone().requires_grad_(True) if with_bias else None
    pre = xc + nwc * noise if with_noise else xc
    yc = so.fused_bias_act(pre, bc, None, 3, 0, 0.2, 2 ** 0.5)
    yc.backward(gy)
    xg = x.cuda().requires_grad_(True)
    nwg = nw.cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True) if with_bias else None
    yg = noise_bias_act(xg, noise.cuda() if with_noise else None, nwg, bg)
    yg.backward(gy.cuda())
    np.testing.assert_allclose(yg.detach().cpu().numpy(), yc.detach().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=1e-6, atol=1e-6)
    if with_bias:
        np.testing.assert_allclose(bg.grad.cpu().numpy(), bc.grad.numpy(), rtol=1e-4, atol=1e-4)
    if with_noise:
        np.testing.assert_allclose(nwg.grad.cpu().numpy(), nwc.grad.numpy(), rtol=1e-4, atol=2e-3)
    else:
        assert nwg.grad is None


@pytest.mark.gpu
@pytest.mark.parametrize("co,ci,k,demod,transposed", [(48, 40, 3, True, False), (12, 64, 1, False, False),
                                                       (40, 33, 3, True, True), (300, 20, 3, True, False)])
def test_modulate_weight_matches_the_reference_formulation(co, ci, k, demod, transposed):
    """ModulatedConv2d's fused weight path (dual_styleunet.py:254-259, :268-272 for the transposed layout) in one kernel each
    way vs the same torch expressions on CPU with autograd (fp32; the demodulation sum order differs)."""
    import math
    import torch
    from animatablegaussians_amd.styleunet_ops import modulate_weight
    g = torch.Generator().manual_seed(co * 7 + ci)
    W = torch.randn(1, co, ci, k, k, generator=g)
    style = torch.randn(1, ci, generator=g) + 1.0
    scale = 1 / math.sqrt(ci * k * k)
    Wc, sc = W.clone().requires_grad_(True), style.clone().requires_grad_(True)
    w = scale * Wc * sc.view(1, 1, ci, 1, 1)
    if demod:
        w = w * torch.rsqrt(w.pow(2).sum([2, 3, 4]) + 1e-8).view(1, co, 1, 1, 1)
    ref = w[0].transpose(0, 1) if transposed else w[0]
    up = torch.randn(ref.shape, generator=g)
    (ref * up).sum().backward()
    Wg, sg = W.cuda().requires_grad_(True), style.cuda().requires_grad_(True)
    out = modulate_weight(Wg, sg, scale, demod, transposed)
    assert out.shape == ref.shape and out.is_contiguous()
    (out * up.cuda()).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(Wg.grad.cpu().numpy(), Wc.grad.numpy(), rtol=1e-4, atol=2e-5 * float(Wc.grad.abs().max()))
    np.testing.assert_allclose(sg.grad.cpu().numpy(), sc.grad.numpy(), rtol=1e-4, atol=2e-5 * float(sc.grad.abs().max()))


@pytest.mark.gpu
def test_haar_split_merge_match_the_upfirdn_formulation():
    """HaarTransform / InverseHaarTransform as one 2x2-block kernel each vs the reference's formulation (four upfirdn2d calls
    with the kernels of get_haar_wavelet, dual_styleunet.py:374-425) on the CPU oracle, forward and backward."""
    import torch
    from animatablegaussians_amd.styleunet_ops import haar_merge, haar_split
    from oracle import styleunet_oracle as so
    a = 1 / (2 ** 0.5)
    lo, hi = torch.tensor([[a, a]]), torch.tensor([[-a, a]])
    k = {"ll": lo.T * lo, "lh": hi.T * lo, "hl": lo.T * hi, "hh": hi.T * hi}
    g = torch.Generator().manual_seed(12)
    x = torch.randn(1, 3, 20, 28, generator=g)
    xc = x.clone().requires_grad_(True)
    ref = torch.cat([so.upfirdn2d(xc[0], k[n], 1, 1, 2, 2, 0, 0, 0, 0)[None] for n in ("ll", "lh", "hl", "hh")], 1)
    up = torch.randn(ref.shape, generator=g)
    (ref * up).sum().backward()
    xg = x.cuda().requires_grad_(True)
    got = haar_split(xg)
    (got * up.cuda()).sum().backward()
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=1e-6, atol=1e-6)

    y = torch.randn(1, 12, 10, 14, generator=g)
    yc = y.clone().requires_grad_(True)
    sign = {"ll": 1.0, "lh": -1.0, "hl": -1.0, "hh": 1.0}
    parts = yc.chunk(4, 1)
    ref2 = sum(so.upfirdn2d(p_[0], sign[n] * k[n], 2, 2, 1, 1, 1, 0, 1, 0)[None] for p_, n in zip(parts, ("ll", "lh", "hl", "hh")))
    up2 = torch.randn(ref2.shape, generator=g)
    (ref2 * up2).sum().backward()
    yg = y.cuda().requires_grad_(True)
    got2 = haar_merge(yg)
    (got2 * up2.cuda()).sum().backward()
    np.testing.assert_allclose(got2.detach().cpu().numpy(), ref2.detach().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(yg.grad.cpu().numpy(), yc.grad.numpy(), rtol=1e-6, atol=1e-6)
    # perfect reconstruction
    np.testing.assert_allclose(haar_merge(haar_split(xg.detach())).cpu().numpy(), x.numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("C,H,W,with_noise", [(5, 33, 21, True), (64, 256, 256, True), (512, 16, 16, True), (3, 17, 260, False)])
def test_activation_parameter_sums_are_deterministic_and_match_the_fp64_oracle(C, H, W, with_noise):
    """ag_noise_bias_act_backward (NoiseInjection + FusedLeakyReLU backward, dual_styleunet.py:303-313, fused_act.py:33-97): the bias sums
    and the noise-strength sum -- ONE number summed over the whole feature map with mixed signs -- come from per-workgroup partial sums and a
    fixed-order finish (round 4; float atomics in arrival order before): repeated launches give the same BITS, and the values sit within
    fp32 summation noise of the fp64 sums (bound: 2e-6 of the sum of the magnitudes of the terms)."""
    import ctypes
    import torch
    from animatablegaussians_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(C * 1000 + H)
    HW = H * W
    y = torch.randn(C, HW, generator=g)                       # the saved forward output: its sign selects the slope
    gy = torch.randn(C, HW, generator=g)
    noise = torch.randn(HW, generator=g) if with_noise else None
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    yd, gyd, nd = y.cuda(), gy.cuda(), (noise.cuda() if with_noise else None)
    part = torch.empty(int(L.ag_noise_bias_act_partial_floats(C, HW)), device="cuda")
    runs = []
    for _ in range(4):
        gx, gb, gw = torch.empty(C, HW, device="cuda"), torch.full((C,), 7.0, device="cuda"), torch.full((1,), 7.0, device="cuda")
        _lib.check(L.ag_noise_bias_act_backward(p(gx), p(gyd), p(yd), p(nd), p(gb), p(gw) if with_noise else None, p(part), C, HW, 0.2, 2 ** 0.5, st),
                   "ag_noise_bias_act_backward")
        runs.append((gx.cpu(), gb.cpu(), gw.cpu()))
    for r in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(runs[0], r)), "the parameter sums changed from launch to launch"
    gx64 = gy.double() * torch.where(y > 0, 1.0, 0.2).double() * 2 ** 0.5
    np.testing.assert_allclose(runs[0][0].numpy(), gx64.float().numpy(), rtol=2e-7, atol=0)
    gx32 = runs[0][0].double()                                 # the sums are of the fp32 gx the kernel wrote
    assert float((runs[0][1].double() - gx32.sum(1)).abs().max()) <= 2e-6 * float(gx32.abs().sum(1).max())
    if with_noise:
        terms = gx32 * noise.double()[None]
        assert abs(float(runs[0][2]) - float(terms.sum())) <= 2e-6 * float(terms.abs().sum())
    # without the sums no scratch is needed
    gx = torch.empty(C, HW, device="cuda")
    _lib.check(L.ag_noise_bias_act_backward(p(gx), p(gyd), p(yd), None, None, None, None, C, HW, 0.2, 2 ** 0.5, st), "no sums")
    assert torch.equal(gx.cpu(), runs[0][0])
    assert L.ag_noise_bias_act_backward(p(gx), p(gyd), p(yd), None, p(torch.empty(C, device="cuda")), None, None, C, HW, 0.2, 2 ** 0.5, st) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("Co,Ci,K2,demod,transposed", [(512, 512, 9, True, False), (64, 128, 9, True, True), (12, 64, 1, False, False), (7, 5, 9, True, False)])
def test_style_gradient_of_the_weight_modulation_is_deterministic(Co, Ci, K2, demod, transposed):
    """ag_modulate_weight_backward (ModulatedConv2d's fused branch, dual_styleunet.py:254-259): dstyle[ci] is a sum over the output channels;
    per-(co, ci) partial sums + a fixed-order finish give the same bits on every launch, values vs torch autograd in fp64."""
    import ctypes
    import torch
    from animatablegaussians_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(Co * 31 + Ci)
    W = torch.randn(Co, Ci, K2, generator=g)
    style = torch.randn(Ci, generator=g) + 1.0
    gout = torch.randn((Ci, Co, K2) if transposed else (Co, Ci, K2), generator=g)
    scale = 1.0 / (Ci * K2) ** 0.5
    Wr, sr = W.double().requires_grad_(True), style.double().requires_grad_(True)
    wm = (scale * Wr) * sr[None, :, None]
    if demod:
        wm = wm * torch.rsqrt(wm.pow(2).sum((1, 2), keepdim=True) + 1e-8)
    ((wm.transpose(0, 1) if transposed else wm) * gout.double()).sum().backward()
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    Wd, sd, gd = W.cuda(), style.cuda(), gout.cuda().contiguous()
    out = torch.empty_like(gd)
    dcoef = torch.empty(Co, device="cuda") if demod else None
    _lib.check(L.ag_modulate_weight_forward(p(out), p(dcoef), p(Wd), p(sd), scale, int(demod), Co, Ci, K2, int(transposed), st), "fwd")
    part = torch.empty(int(L.ag_modulate_weight_partial_floats(Co, Ci)), device="cuda")
    runs = []
    for _ in range(4):
        dW, ds = torch.empty(Co, Ci, K2, device="cuda"), torch.full((Ci,), 3.0, device="cuda")
        _lib.check(L.ag_modulate_weight_backward(p(dW), p(ds), p(part), p(gd), p(Wd), p(sd), p(dcoef), scale, int(demod), Co, Ci, K2, int(transposed), st),
                   "bwd")
        runs.append((dW.cpu(), ds.cpu()))
    for r in runs[1:]:
        assert torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]), "dstyle changed from launch to launch"
    np.testing.assert_allclose(runs[0][0].numpy(), Wr.grad.float().numpy(), rtol=2e-4, atol=2e-5 * float(Wr.grad.abs().max()))
    np.testing.assert_allclose(runs[0][1].numpy(), sr.grad.float().numpy(), rtol=2e-4, atol=2e-5 * float(sr.grad.abs().max()))


def test_skip_chain_taps_reproduce_the_reference_chain_on_the_cpu():
    """The 576 coefficients of ToRGB's composed skip map (styleunet_ops.skip_chain_taps: InverseHaarTransform -> Upsample -> HaarTransform,
    dual_styleunet.py:607-633) against the three stages run one after the other by the oracle (the reference's own upfirdn2d CPU path
    restated), float64, odd sizes, every border."""
    import torch
    from animatablegaussians_amd.styleunet_ops import skip_chain_taps
    from oracle import dual_styleunet_oracle as do
    k = do._fir(gain=4.0, dtype=torch.float64)
    taps = np.array(skip_chain_taps(k.numpy()), np.float64).reshape(4, 2, 2, 4, 3, 3)
    assert int((taps != 0).sum()) == 256                      # two of the three row (column) offsets per parity: 4 x 4 x 4 x (2 x 2)
    rs = np.random.RandomState(0)
    for C, h, w in ((3, 7, 6), (1, 1, 1), (2, 2, 9)):
        x = torch.from_numpy(rs.normal(size=(1, 4 * C, h, w)))
        ref = do.haar_split(do.upfirdn2d(do.haar_merge(x), k, up=2, pad=(2, 1))).numpy()
        xp = np.pad(x.numpy()[0].reshape(4, C, h, w), ((0, 0), (0, 0), (1, 1), (1, 1)))
        out = np.zeros((4, C, 2 * h, 2 * w))
        for so in range(4):
            for py in range(2):
                for px in range(2):
                    acc = np.zeros((C, h, w))
                    for s in range(4):
                        for a in range(3):
                            for b in range(3):
                                acc += taps[so, py, px, s, a, b] * xp[s, :, a:a + h, b:b + w]
                    out[so, :, py::2, px::2] = acc
        assert np.abs(out.reshape(ref.shape) - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())     # taps are stored as float32
    # the two 1-D factors the kernels take reproduce the 2-D coefficients; a kernel that is not an outer product is refused
    from animatablegaussians_amd.styleunet_ops import skip_chain_taps_1d
    f = np.array(skip_chain_taps_1d(k.numpy()), np.float64)
    wy, wx = f[:24].reshape(2, 2, 2, 3), f[24:].reshape(2, 2, 2, 3)
    for so in range(4):
        for s in range(4):
            want = taps[so, :, :, s]
            got = np.einsum("pa,qb->pqab", wy[so & 1, :, s & 1, :], wx[so >> 1, :, s >> 1, :])
            assert np.abs(got - want).max() <= 1e-6
    with pytest.raises(RuntimeError):
        skip_chain_taps_1d(np.arange(16.0).reshape(4, 4) + np.eye(4))


@pytest.mark.gpu
@pytest.mark.parametrize("C,h,w", [(3, 8, 8), (3, 33, 17), (1, 1, 1), (3, 256, 256)])
def test_skip_chain_kernel_equals_the_three_kernel_path_and_the_oracle(C, h, w):
    """ag_skip_chain_forward / _backward against (a) the three kernels they replace (block merge, upfirdn2d up = 2, block split) with
    their autograd, (b) the float64 oracle chain; accumulate mode adds into an existing output."""
    import torch
    from animatablegaussians_amd import styleunet_ops as so
    from oracle import dual_styleunet_oracle as do
    rs = np.random.RandomState(C * 1000 + h)
    x_np = rs.normal(size=(1, 4 * C, h, w)).astype(np.float32)
    g_np = rs.normal(size=(1, 4 * C, 2 * h, 2 * w)).astype(np.float32)
    k_up = do._fir(gain=4.0).cuda()
    x1 = torch.from_numpy(x_np).cuda().requires_grad_(True)
    y1 = so.skip_chain(x1, k_up)
    y1.backward(torch.from_numpy(g_np).cuda())
    x2 = torch.from_numpy(x_np).cuda().requires_grad_(True)
    y2 = so.haar_split(so.upfirdn2d_nchw(so.haar_merge(x2), k_up, up=2, pad=(2, 1)))
    y2.backward(torch.from_numpy(g_np).cuda())
    x3 = torch.from_numpy(x_np).double().requires_grad_(True)
    y3 = do.haar_split(do.upfirdn2d(do.haar_merge(x3), do._fir(gain=4.0, dtype=torch.float64), up=2, pad=(2, 1)))
    y3.backward(torch.from_numpy(g_np).double())
    torch.cuda.synchronize()
    for got, three, ref, nm in ((y1, y2, y3, "forward"), (x1.grad, x2.grad, x3.grad, "gradient")):
        ref = ref.detach().numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        assert float(np.abs(got.detach().cpu().numpy() - ref).max()) <= 2e-6 * scale, nm
        assert float((got.detach() - three.detach()).abs().max()) <= 2e-6 * scale, nm + " vs the three kernels"
    base = torch.from_numpy(g_np).cuda().clone()
    so.skip_chain_forward_(base, x1.detach(), k_up, accumulate=True)
    assert float((base - (torch.from_numpy(g_np).cuda() + y1.detach())).abs().max()) <= 1e-6 * max(1.0, float(y1.abs().max()))
