"""include/ag_linear.h on the GPU: groups of EqualLinear layers on one-row inputs (reference network/styleunet/dual_styleunet.py:131-165, the
mapping network :594-610) and the bilinear resize of the view-direction feature (:881-883), against the reference formulas evaluated by
torch on the CPU (float64 as the yardstick where the summation order differs)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_equal_linear(x, w, b, lr_mul, activation):
    """EqualLinear.forward, dual_styleunet.py:152-165, with fused_leaky_relu's CPU branch (fused_act.py:118-129) spelled out."""
    scale = (1 / math.sqrt(w.shape[1])) * lr_mul
    if activation:
        out = F.linear(x, w * scale)
        return F.leaky_relu(out + (b * lr_mul)[None], 0.2) * math.sqrt(2)
    return F.linear(x, w * scale, bias=b * lr_mul if b is not None else None)


@pytest.mark.parametrize("B", [1, 3])
@pytest.mark.parametrize("activation", [False, True])
def test_equal_linear_group_forward_backward_equals_the_reference_formula(B, activation):
    from animatablegaussians_amd.linear_ops import equal_linear_group
    g = torch.Generator().manual_seed(5)
    fin, outs = 512, [512, 12, 100, 256, 17]           # (17 and 100: ragged 16-row chunks of the backward)
    lr_mul = 0.01 if activation else 1.0
    x1, x2 = torch.randn(B, fin, generator=g), torch.randn(B, fin, generator=g)
    ws = [torch.randn(o, fin, generator=g) / lr_mul for o in outs]
    bs = [torch.randn(o, generator=g) for o in outs]
    bs[1] = None if not activation else bs[1]
    xs_of = lambda a, b: [a, a, a, b, b]               # noqa: E731  jobs 0-2 share the first input, jobs 3-4 the second
    up = torch.randn(B, sum(outs), generator=g)

    def run(dev, dtype, fn):
        leaf = lambda t: t.detach().clone().to(dev, dtype).requires_grad_(True)      # noqa: E731
        a, b = leaf(x1), leaf(x2)
        W = [leaf(w) for w in ws]
        Bi = [leaf(bb) if bb is not None else None for bb in bs]
        y = fn(xs_of(a, b), W, Bi)
        y.backward(up.to(dev, dtype))
        return (y.detach().cpu().double(), a.grad.cpu().double(), b.grad.cpu().double(), [w.grad.cpu().double() for w in W],
                [bb.grad.cpu().double() if bb is not None else None for bb in Bi])

    ref = lambda xs, W, Bi: torch.cat([_ref_equal_linear(x, w, bb, lr_mul, activation) for x, w, bb in zip(xs, W, Bi)], 1)     # noqa: E731
    want = run("cpu", torch.float64, ref)
    ref32 = run("cpu", torch.float32, ref)
    got = run("cuda", torch.float32, lambda xs, W, Bi: torch.cat(equal_linear_group(xs, W, Bi, lr_mul=lr_mul, activation=activation), 1))
    again = run("cuda", torch.float32, lambda xs, W, Bi: torch.cat(equal_linear_group(xs, W, Bi, lr_mul=lr_mul, activation=activation), 1))

    def close(a, b, r32, what):
        scale = float(b.abs().max()) + 1e-30
        err, err32 = float((a - b).abs().max()) / scale, float((r32 - b).abs().max()) / scale
        assert err <= max(4 * err32, 2e-6), f"{what}: {err:.2e} of the scale (torch fp32: {err32:.2e})"

    close(got[0], want[0], ref32[0], "y")
    close(got[1], want[1], ref32[1], "g_x (shared by three jobs)")
    close(got[2], want[2], ref32[2], "g_x (shared by two jobs)")
    for j in range(len(outs)):
        close(got[3][j], want[3][j], ref32[3][j], f"g_weight[{j}]")
        if bs[j] is not None:
            close(got[4][j], want[4][j], ref32[4][j], f"g_bias[{j}]")
    # fixed summation orders: the same bits on every run
    assert torch.equal(got[0], again[0]) and torch.equal(got[1], again[1]) and all(torch.equal(a, b) for a, b in zip(got[3], again[3]))


def test_mapping_network_equals_the_reference_formula():
    """styleunet.latents_of: PixelNorm + n_mlp x EqualLinear(lr_mul 0.01, fused leaky ReLU) of three networks, two launches in all."""
    from animatablegaussians_amd.styleunet import DualStyleUNet, latents_of
    torch.manual_seed(11)
    nets = [DualStyleUNet(inp_size=512, inp_ch=3, out_ch=c, out_size=1024, style_dim=512, n_mlp=2).cuda() for c in (3, 3, 8)]
    zs = [torch.randn(1, 512).cuda() for _ in nets]
    got = latents_of(nets, zs)
    loss = sum((w * (i + 1)).sum() for i, w in enumerate(got))
    loss.backward()
    for n, z, w in zip(nets, zs, got):
        x = z.double().cpu()
        x = x * torch.rsqrt(torch.mean(x ** 2, dim=1, keepdim=True) + 1e-8)
        params = []
        for i in range(n.n_mlp):
            W = n._p(f"style.{i + 1}.weight").detach().double().cpu().requires_grad_(True)
            b = n._p(f"style.{i + 1}.bias").detach().double().cpu().requires_grad_(True)
            x = _ref_equal_linear(x, W, b, n.lr_mlp, True)
            params.append((W, b))
        assert float((w.detach().cpu().double() - x.detach()).abs().max()) <= 2e-6 * float(x.detach().abs().max())
        (x * (nets.index(n) + 1)).sum().backward()
        for i, (W, b) in enumerate(params):
            gw, gb = n._p(f"style.{i + 1}.weight").grad.cpu().double(), n._p(f"style.{i + 1}.bias").grad.cpu().double()
            assert float((gw - W.grad).abs().max()) <= 1e-5 * float(W.grad.abs().max())
            assert float((gb - b.grad).abs().max()) <= 1e-5 * float(b.grad.abs().max())


@pytest.mark.parametrize("shape,size", [((1, 128, 128, 128), (256, 256)), ((2, 5, 37, 53), (64, 101)), ((1, 3, 64, 48), (20, 31)), ((1, 2, 9, 9), (9, 9)),
                                        ((1, 4, 1, 7), (5, 3))])
def test_bilinear_resize_equals_torch_interpolate(shape, size):
    from animatablegaussians_amd.linear_ops import bilinear_resize
    g = torch.Generator().manual_seed(2)
    x = torch.randn(*shape, generator=g)
    up = torch.randn(shape[0], shape[1], *size, generator=g)
    xc = x.clone().requires_grad_(True)
    want = F.interpolate(xc, size, mode="bilinear")
    want.backward(up)
    xg = x.cuda().requires_grad_(True)
    got = bilinear_resize(xg, size)
    got.backward(up.cuda())
    assert got.shape == want.shape
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=0, atol=2e-6 * float(want.abs().max()))
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=0, atol=4e-6 * float(xc.grad.abs().max()))
    # the gather backward has a fixed order: same bits again
    xg2 = x.cuda().requires_grad_(True)
    bilinear_resize(xg2, size).backward(up.cuda())
    assert torch.equal(xg.grad, xg2.grad)


def test_cpu_tensors_are_refused():
    from animatablegaussians_amd.linear_ops import bilinear_resize, equal_linear_group
    with pytest.raises(RuntimeError):
        equal_linear_group([torch.randn(1, 8)], [torch.randn(4, 8)], [None])
    with pytest.raises(RuntimeError):
        bilinear_resize(torch.randn(1, 1, 4, 4), (8, 8))


@pytest.mark.parametrize("shape", [(2, 32, 512, 512), (4, 12, 64, 64), (3, 5, 7, 9), (1, 1, 1, 1), (6, 12, 16, 16)])
def test_plane_sums_equal_the_float64_sums_and_are_reproducible(shape):
    from animatablegaussians_amd.linear_ops import plane_sums
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(9))
    want = x.double().sum((2, 3))
    got = plane_sums(x.cuda())
    assert got.shape == want.shape
    n = shape[2] * shape[3]
    assert float((got.cpu().double() - want).abs().max()) <= 2e-6 * math.sqrt(n) * 4
    assert torch.equal(got, plane_sums(x.cuda()))
    # a non-contiguous / unaligned view takes the scalar path
    y = torch.randn(shape[0], shape[1], shape[2], shape[3] + 1, generator=torch.Generator().manual_seed(10)).cuda()[..., 1:]
    assert float((plane_sums(y).cpu().double() - y.cpu().double().sum((2, 3))).abs().max()) <= 2e-6 * math.sqrt(n) * 4


@pytest.mark.parametrize("vf_size", [(16, 12), (32, 24), None])
def test_select_add_rows_equals_index_select_interpolate_add(vf_size):
    """grouped._SelectAddRows (the input of a view-dependent decoder stage): x[m] = out[src[m]] + F.interpolate(vf[m - r0], bilinear) on the rows
    that have a view feature, forward and both gradients, against the torch composition on the CPU."""
    from animatablegaussians_amd.grouped import _SelectAddRows
    g = torch.Generator().manual_seed(4)
    out = torch.randn(3, 6, 32, 24, generator=g)
    src, rows = (0, 1, 2, 2, 1, 2), (2, 5)
    vf = torch.randn(3, 6, *vf_size, generator=g) if vf_size else None
    up = torch.randn(len(src), 6, 32, 24, generator=g)

    oc = out.clone().requires_grad_(True)
    vc = vf.clone().requires_grad_(True) if vf is not None else None
    want = oc.index_select(0, torch.tensor(src))
    if vc is not None:
        f = vc if vf_size == (32, 24) else F.interpolate(vc, (32, 24), mode="bilinear")
        want = torch.cat([want[:rows[0]], want[rows[0]:rows[1]] + f, want[rows[1]:]], 0)
    want.backward(up)

    og = out.cuda().requires_grad_(True)
    vg = vf.cuda().requires_grad_(True) if vf is not None else None
    got = _SelectAddRows.apply(og, vg, src, rows if vf is not None else None)
    got.backward(up.cuda())
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=0, atol=3e-6 * float(want.abs().max()))
    np.testing.assert_allclose(og.grad.cpu().numpy(), oc.grad.numpy(), rtol=0, atol=3e-6 * float(oc.grad.abs().max()))
    if vf is not None:
        np.testing.assert_allclose(vg.grad.cpu().numpy(), vc.grad.numpy(), rtol=0, atol=4e-6 * float(vc.grad.abs().max()))
