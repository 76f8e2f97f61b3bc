"""Grouped execution of the StyleUNet layers (round 4: include/ag_layers.h AgGroupedLayerArgs, animatablegaussians_amd/grouped.py).

The avatar's three DualStyleUNets (network/avatar.py:34-36) run as one launch chain with a group dimension.  The reference for a grouped
call is the SAME layer run instance by instance through the single-instance path, whose parity with the reference module is pinned in
test_styleunet_net.py / test_styleunet_ops.py / test_conv_gpu.py; grouped and one-by-one run the same kernels in the same order and may
differ only in the split-K slice count of a convolution (fp32 summation order): 2e-5 of the tensor's largest value here."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _close(a, b, what, tol=TOL):
    scale = float(b.abs().max())
    err = float((a - b).abs().max())
    assert err <= tol * scale + 1e-30, (what, err, scale)


@pytest.mark.parametrize("G,Cin,Cout,H,down,shared", [(3, 16, 24, 20, False, False), (2, 3, 32, 32, True, True), (6, 32, 16, 12, False, False),
                                                       (3, 16, 32, 16, True, False), (3, 3, 16, 64, False, True)])
def test_grouped_conv_layer_equals_the_layers_one_by_one(G, Cin, Cout, H, down, shared):
    import torch
    from animatablegaussians_amd import fused_layers as fl, grouped as gr
    g = torch.Generator().manual_seed(G * 100 + Cin)
    k = 1 if (shared and not down) else 3
    x = torch.randn(1 if shared else G, Cin, H, H, generator=g).cuda()
    ws = [(torch.randn(Cout, Cin, k, k, generator=g)).cuda().requires_grad_(True) for _ in range(G)]
    bs = [(torch.randn(Cout, generator=g) * 0.1).cuda().requires_grad_(True) for _ in range(G)]
    kb = torch.tensor([1., 3., 3., 1.])
    kb = (kb[None] * kb[:, None] / 64).cuda()
    scale = 1 / (Cin * k * k) ** 0.5
    xg = x.clone().requires_grad_(not shared)
    out = gr.grouped_conv_layer(xg, ws, bs, kb, scale, down, shared)
    up = torch.randn(out.shape, generator=g).cuda()
    out.backward(up)
    got = [out.detach()] + ([xg.grad] if not shared else []) + [w.grad.clone() for w in ws] + [b.grad.clone() for b in bs]
    for t in ws + bs:
        t.grad = None
    xs = [x[0 if shared else i][None].clone().requires_grad_(not shared) for i in range(G)]
    outs = [fl.conv_layer(xs[i], ws[i], bs[i], kb, scale, down) for i in range(G)]
    torch.cat(outs, 0).backward(up)
    want = [torch.cat(outs, 0).detach()] + ([torch.cat([t.grad for t in xs], 0)] if not shared else []) + [w.grad for w in ws] + [b.grad for b in bs]
    for i, (a, b) in enumerate(zip(got, want)):
        _close(a, b, i)


@pytest.mark.parametrize("G,Cin,Cout,H,up", [(3, 32, 16, 10, True), (6, 16, 16, 24, False), (4, 64, 32, 8, True), (2, 16, 48, 33, False)])
def test_grouped_styled_conv_equals_the_layers_one_by_one(G, Cin, Cout, H, up):
    import torch
    from animatablegaussians_amd import fused_layers as fl, grouped as gr
    g = torch.Generator().manual_seed(G * 100 + Cin)
    OH = 2 * H if up else H
    x = torch.randn(G, Cin, H, H, generator=g).cuda()
    ws = [torch.randn(1, Cout, Cin, 3, 3, generator=g).cuda().requires_grad_(True) for _ in range(G)]
    st = [(torch.randn(1, Cin, generator=g) + 1).cuda().requires_grad_(True) for _ in range(G)]
    nz = [torch.randn(1, 1, OH, OH, generator=g).cuda() for _ in range(G)]
    nw = [torch.randn(1, generator=g).cuda().requires_grad_(True) for _ in range(G)]
    bs = [(torch.randn(Cout, generator=g) * 0.1).cuda().requires_grad_(True) for _ in range(G)]
    kb = torch.tensor([1., 3., 3., 1.])
    kb = (kb[None] * kb[:, None] / 64 * 4).cuda()
    scale = 1 / (Cin * 9) ** 0.5
    params = ws + st + nw + bs
    xg = x.clone().requires_grad_(True)
    out = gr.grouped_styled_conv(xg, ws, st, nz, nw, bs, kb if up else None, scale, up)
    upg = torch.randn(out.shape, generator=g).cuda()
    out.backward(upg)
    got = [out.detach(), xg.grad] + [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    xs = [x[i][None].clone().requires_grad_(True) for i in range(G)]
    outs = [fl.styled_conv(xs[i], ws[i], st[i], nz[i], nw[i], bs[i], kb if up else None, scale, up) for i in range(G)]
    torch.cat(outs, 0).backward(upg)
    want = [torch.cat(outs, 0).detach(), torch.cat([t.grad for t in xs], 0)] + [p.grad for p in params]
    names = ["out", "gx"] + [f"{n}{i}" for n in ("w", "style", "nw", "bias") for i in range(G)]
    for n, a, b in zip(names, got, want):
        _close(a, b, n, tol=5e-5 if n.startswith(("style", "nw")) else TOL)


@pytest.mark.parametrize("up_first", [True, False])
def test_maxima_handed_over_by_the_producer_equal_the_swept_ones(up_first):
    """AgGroupedLayerArgs.out_maxima / x_maxima (fp16 split form): the kernel that writes a layer's output (Blur + activation, the convolution
    epilogue, the split-K finish) leaves its largest magnitudes for the next grouped call, which then does not sweep its input.  The maxima
    themselves are exact (a maximum does not depend on who takes it), so a chain of two StyledConvs and a comb convolution gives the SAME BITS,
    forward and backward, whether the second and third call take the maxima from the tensor they are handed or sweep it (attribute removed)."""
    import torch
    from animatablegaussians_amd import conv as agc, grouped as gr
    if agc.get_math() != "split_f16":
        pytest.skip("operand maxima exist in the fp16 split form only")
    g = torch.Generator().manual_seed(77)
    G, C, H = 4, 32, 16
    kb = torch.tensor([1., 3., 3., 1.])
    kb = (kb[None] * kb[:, None] / 16).cuda()

    def params(cin, cout):
        return ([torch.randn(1, cout, cin, 3, 3, generator=g).cuda().requires_grad_(True) for _ in range(G)],
                [(torch.randn(1, cin, generator=g) + 1).cuda().requires_grad_(True) for _ in range(G)],
                [torch.randn(1, generator=g).cuda().requires_grad_(True) for _ in range(G)],
                [(torch.randn(cout, generator=g) * 0.1).cuda().requires_grad_(True) for _ in range(G)])
    p1, p2 = params(C, C), params(C, C)
    OH = 2 * H if up_first else H
    nz1 = [torch.randn(1, 1, OH, OH, generator=g).cuda() for _ in range(G)]
    nz2 = [torch.randn(1, 1, OH, OH, generator=g).cuda() for _ in range(G)]
    wc = [torch.randn(C, 2 * C, 3, 3, generator=g).cuda().requires_grad_(True) for _ in range(2)]
    bc = [(torch.randn(C, generator=g) * 0.1).cuda().requires_grad_(True) for _ in range(2)]
    lev = torch.randn(2, C, OH, OH, generator=g).cuda()
    x0 = torch.randn(G, C, H, H, generator=g).cuda()
    up = torch.randn(G, C, OH, OH, generator=g).cuda()
    leaves = p1[0] + p1[1] + p1[2] + p1[3] + p2[0] + p2[1] + p2[2] + p2[3] + wc + bc

    def run(hand_over):
        for t in leaves:
            t.grad = None
        x = x0.clone().requires_grad_(True)
        a = gr.grouped_styled_conv(x, p1[0], p1[1], nz1, p1[2], p1[3], kb, 1 / (C * 9) ** 0.5, up_first)
        assert getattr(a, "_ag_maxima", None) is not None            # the producer left them
        if not hand_over:
            del a._ag_maxima
        b = gr.grouped_styled_conv(a, p2[0], p2[1], nz2, p2[2], p2[3], None, 1 / (C * 9) ** 0.5, False)
        if not hand_over:
            del b._ag_maxima
        c = gr._GroupedComb.apply((0, 2, 4), 1 / (2 * C * 9) ** 0.5, b, lev, wc[0], wc[1], bc[0], bc[0], bc[1], bc[1])
        c.backward(up)
        return [c.detach().clone(), x.grad.clone()] + [t.grad.clone() for t in leaves]

    with_h, without = run(True), run(False)
    for i, (u, v) in enumerate(zip(with_h, without)):
        if i < 2:                                # the output and the input gradient: deterministic kernels all the way
            assert torch.equal(u, v), (i, float((u - v).abs().max()))
        else:                                    # parameter gradients: the weight gradients (and the style gradients derived from them) go through
            _close(u, v, i, tol=1e-5)            # split-K float atomics -- equal up to their summation order, as two runs of ONE path are


@pytest.mark.parametrize("G,Cin,Cout,H,with_skip", [(4, 64, 12, 32, True), (2, 32, 32, 16, True), (3, 16, 12, 8, False), (4, 64, 12, 128, True)])
def test_grouped_to_rgb_equals_the_heads_one_by_one(G, Cin, Cout, H, with_skip):
    import torch
    from animatablegaussians_amd import fused_layers as fl, grouped as gr
    g = torch.Generator().manual_seed(G * 100 + Cin)
    x = torch.randn(G, Cin, H, H, generator=g).cuda()
    ws = [torch.randn(1, Cout, Cin, 1, 1, generator=g).cuda().requires_grad_(True) for _ in range(G)]
    st = [(torch.randn(1, Cin, generator=g) + 1).cuda().requires_grad_(True) for _ in range(G)]
    bs = [(torch.randn(Cout, generator=g) * 0.1).cuda().requires_grad_(True) for _ in range(G)]
    skip = torch.randn(G, Cout, H // 2, H // 2, generator=g).cuda() if with_skip else None
    kb = torch.tensor([1., 3., 3., 1.])
    kb = (kb[None] * kb[:, None] / 64 * 4).cuda()
    scale = 1 / Cin ** 0.5
    params = ws + st + bs
    xg = x.clone().requires_grad_(True)
    sg = skip.clone().requires_grad_(True) if with_skip else None
    out = gr.grouped_to_rgb(xg, ws, st, bs, sg, kb, scale)
    upg = torch.randn(out.shape, generator=g).cuda()
    out.backward(upg)
    got = [out.detach(), xg.grad] + ([sg.grad] if with_skip else []) + [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    xs = [x[i][None].clone().requires_grad_(True) for i in range(G)]
    ss = [skip[i][None].clone().requires_grad_(True) if with_skip else None for i in range(G)]
    outs = [fl.to_rgb(xs[i], ws[i], st[i], bs[i], ss[i], kb, scale) for i in range(G)]
    torch.cat(outs, 0).backward(upg)
    want = [torch.cat(outs, 0).detach(), torch.cat([t.grad for t in xs], 0)] + ([torch.cat([t.grad for t in ss], 0)] if with_skip else []) + [p.grad for p in params]
    for i, (a, b) in enumerate(zip(got, want)):
        _close(a, b, i, tol=5e-5)


@pytest.mark.parametrize("begin,C1,C2,Cout,H", [((0, 2, 4, 6), 32, 32, 16, 12), ((0, 2, 6, 8), 16, 48, 32, 10), ((0, 4), 64, 64, 64, 8), ((0, 1, 2, 3), 16, 16, 128, 16)])
def test_comb_convolution_without_the_concatenation_equals_the_concatenated_one(begin, C1, C2, Cout, H):
    """ag_grouped_comb_* (dual_styleunet.py:877-879): conv(cat(out_m, lev_r), W_r) as conv(out_m, W_r[:, :C1]) + conv(lev_r, W_r[:, C1:]) with
    the second half once per network, against the concatenation + one grouped ConvLayer per member; uneven member counts per network
    (several camera views of the colour network) included.  Differences: the association of the channel sum (two partial sums added in
    fp32)."""
    import torch
    from animatablegaussians_amd import grouped as gr
    N, M = len(begin) - 1, begin[-1]
    net_of = [r for r in range(N) for _ in range(begin[r + 1] - begin[r])]
    g = torch.Generator().manual_seed(M * 100 + C1)
    x = torch.randn(M, C1, H, H, generator=g).cuda()
    lev = torch.randn(N, C2, H, H, generator=g).cuda()
    ws = [torch.randn(Cout, C1 + C2, 3, 3, generator=g).cuda().requires_grad_(True) for _ in range(N)]
    bs = [(torch.randn(Cout, generator=g) * 0.1).cuda().requires_grad_(True) for _ in range(N)]
    scale = 1 / ((C1 + C2) * 9) ** 0.5
    up = torch.randn(M, Cout, H, H, generator=g).cuda()
    xa, la = x.clone().requires_grad_(True), lev.clone().requires_grad_(True)
    out = gr._GroupedComb.apply(tuple(begin), scale, xa, la, *ws, *[bs[r] for r in net_of])
    out.backward(up)
    got = [out.detach(), xa.grad, la.grad] + [w.grad.clone() for w in ws] + [b.grad.clone() for b in bs]
    for t in ws + bs:
        t.grad = None
    xb, lb = x.clone().requires_grad_(True), lev.clone().requires_grad_(True)
    cat = torch.cat([xb, lb[net_of]], 1)
    ref = gr.grouped_conv_layer(cat, [ws[r] for r in net_of], [bs[r] for r in net_of], None, scale, False)
    ref.backward(up)
    want = [ref.detach(), xb.grad, lb.grad] + [w.grad for w in ws] + [b.grad for b in bs]
    for i, (a, b) in enumerate(zip(got, want)):
        _close(a, b, i, tol=5e-5)


@pytest.fixture(scope="module")
def net():
    import torch
    from animatablegaussians_amd.avatar import AvatarNet
    torch.manual_seed(31359)
    return AvatarNet.synthetic({'with_viewdirs': True})


def _items(net):
    import sys
    import os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_avatar_net_gpu import _items as make
    return make(net)


def test_three_networks_as_one_chain_equal_the_networks_one_at_a_time(net):
    """AvatarNet.get_maps (network/avatar.py:93-124): the grouped chain (G = 3 encoders, G = 6 decoders) against position_net / other_net /
    color_net run one after the other -- the maps, and after a backward through the same upstream gradients a sample of parameter gradients
    of every layer kind, every level and all three networks."""
    import torch
    items = _items(net)
    net.get_pose_map(items)
    net.eval()                                   # no view-direction jitter: both sides see the same features
    fv, bv = net.get_viewdir_feat(items)
    fv, bv = fv.detach(), bv.detach()
    pose = items['smpl_pos_map'][:3]
    g = torch.Generator().manual_seed(11)
    ups = [torch.randn(1, c, 1024, 1024, generator=g).cuda() for c in (6, 16, 6)]
    probe = ["conv_in.1.weight", "conv_in.2.bias", "from_rgbs.0.conv.0.weight", "from_rgbs.4.conv.1.bias", "cond_convs.0.conv1.0.weight",
             "cond_convs.2.conv2.1.weight", "cond_convs.4.conv2.2.bias", "comb_convs.0.0.weight", "comb_convs.3.0.weight", "comb_convs.5.1.bias",
             "convs1.0.conv.weight", "convs2.0.conv.modulation.weight", "convs1.3.noise.weight", "convs2.9.noise.weight", "convs1.10.conv.weight",
             "convs2.11.activate.bias", "convs1.11.conv.modulation.bias", "to_rgbs1.0.conv.weight", "to_rgbs2.3.bias", "to_rgbs1.5.conv.modulation.weight",
             "to_rgbs2.5.conv.weight", "style.1.weight", "style.2.bias"]
    res = {}
    for mode in (True, False):
        prev = net.set_grouped(mode)
        try:
            net.zero_grad(set_to_none=True)
            fvr, bvr = fv.clone().requires_grad_(True), bv.clone().requires_grad_(True)
            maps = net.get_maps(pose, fvr, bvr)
            torch.autograd.backward(list(maps), ups)
            torch.cuda.synchronize()
            res[mode] = ([m.detach().clone() for m in maps], {(n, k): getattr(net, n)._p(k).grad.clone()
                                                             for n in ("position_net", "other_net", "color_net") for k in probe},
                         [fvr.grad.clone(), bvr.grad.clone()])
        finally:
            net.set_grouped(prev)
    net.zero_grad(set_to_none=True)
    for a, b, name in zip(res[True][0], res[False][0], ("position", "other", "colour")):
        assert a.shape == b.shape
        _close(a, b, name, tol=2e-5)
    worst = 0.0
    for key, b in res[False][1].items():
        a = res[True][1][key]
        scale = float(b.abs().max())
        assert scale > 0, key
        err = float((a - b).abs().max()) / scale
        worst = max(worst, err)
        # The two paths differ in the split-K slice counts of their convolutions, i.e. in fp32 summation order; through ~45 layers that
        # flips leaky-ReLU slopes of activations within rounding of zero, and the reference's OWN fp32 run deviates from its fp64 run by
        # a median 9e-5 and up to 6e-2 of a tensor's largest gradient (test_styleunet_net.py).  A mis-wired member is O(1).
        assert err <= (2e-2 if key[1].endswith("noise.weight") else 5e-3), (key, err)
    for a, b in zip(res[True][2], res[False][2]):
        # an ACTIVATION gradient (16 M elements): where a pre-activation sits within rounding of zero the two summation orders select
        # different leaky-ReLU slopes and the element's gradient differs by a factor 5 -- isolated elements, so: relative L2 and the fraction
        # of elements off by more than 1e-3 of the largest, not the maximum
        l2 = float((a - b).norm() / b.norm())
        off = float(((a - b).abs() > 1e-3 * float(b.abs().max())).float().mean())
        assert l2 <= 3e-3 and off <= 5e-3, ("view feature gradient", l2, off)       # measured 1.0e-3 / 1.5e-3; a mis-wired member: l2 ~ 1
    print(f"grouped vs one-by-one: worst relative parameter-gradient deviation {worst:.2e} over {len(res[False][1])} probes")


def test_grouped_multi_view_equals_per_view_renders(net):
    """render_views on the grouped chain (shared stages G = 6, the view-dependent stage with the colour members once per view) against
    render() per view."""
    import torch
    from animatablegaussians_amd import synth
    base = _items(net)
    net.get_pose_map(base)
    cams = synth.free_view_cameras(3, img=1024)
    views = [{'extr': torch.from_numpy(np.ascontiguousarray(c["extr"])).float().cuda(),
              'intr': torch.from_numpy(np.ascontiguousarray(c["intr"])).float().cuda(), 'img_w': 1024, 'img_h': 1024} for c in cams]
    pose_items = {k: base[k] for k in ('smpl_pos_map', 'cano2live_jnt_mats', 'cano2live_jnt_mats_woRoot')}
    net.eval()
    with torch.no_grad():
        multi = net.render_views(pose_items, views, bg_color=(0.2, 0.1, 0.0))
        for v, m in zip(views, multi):
            single = net.render({**pose_items, **v}, bg_color=(0.2, 0.1, 0.0))
            for key in ('pos_map', 'cano_tex_map', 'offset'):
                _close(m[key], single[key], key, tol=2e-5)
            # the image goes through the rasterizer's discrete decisions: compare away from them
            d = (m['rgb_map'] - single['rgb_map']).abs()
            assert float((d > 1e-3).float().mean()) < 1e-3, float((d > 1e-3).float().mean())


def test_grouped_chain_without_view_directions_and_under_no_grad():
    """`with_viewdirs=False` (network/avatar.py:21,42): no view feature reaches the colour network -- the grouped chain then never copies the
    shared decoder state; eval / no-grad calls; one and three views."""
    import torch
    from animatablegaussians_amd import synth
    from animatablegaussians_amd.avatar import AvatarNet
    torch.manual_seed(31359)
    net = AvatarNet.synthetic({'with_viewdirs': False})
    items = _items(net)
    net.get_pose_map(items)
    net.eval()
    pose = items['smpl_pos_map'][:3]
    with torch.no_grad():
        got = net.get_maps(pose)
        prev = net.set_grouped(False)
        try:
            want = net.get_maps(pose)
        finally:
            net.set_grouped(prev)
        for a, b, name in zip(got, want, ("position", "other", "colour")):
            _close(a, b, name, tol=2e-5)
        cams = synth.free_view_cameras(3, img=1024)
        views = [{'extr': torch.from_numpy(np.ascontiguousarray(c["extr"])).float().cuda(),
                  'intr': torch.from_numpy(np.ascontiguousarray(c["intr"])).float().cuda(), 'img_w': 1024, 'img_h': 1024} for c in cams]
        multi = net.render_views(items, views)
        single = net.render({**items, **views[1]})
        _close(multi[1]['cano_tex_map'], single['cano_tex_map'], "colour map of view 1", tol=2e-5)


def test_a_stale_handed_over_maximum_is_reported_not_silent():
    """Range guard of the scaled fp16 forms (include/ag_conv.h ag_conv_status, round 5): a producer's maxima travel with the tensor
    (grouped._handed_maxima); writes that bypass the version counter leave them stale.  A maximum too small by 2^8 overflows fp16 in the
    consumer's loader: the kernel raises the host-visible flag, ``check_status`` (and the next convolution call) raise AgNativeError with
    AG_ERR_RANGE instead of inf / NaN travelling on silently; after that the library works as before."""
    import torch
    from animatablegaussians_amd import _lib, conv as agc, grouped as gr
    if not agc.needs_maxima():
        pytest.skip("the arithmetic mode of this process does not scale its operands")
    g = torch.Generator().manual_seed(3)
    G, Cin, Cout, H = 3, 32, 32, 24
    x = torch.randn(G, Cin, H, H, generator=g).cuda()
    ws = [torch.randn(Cout, Cin, 3, 3, generator=g).cuda() for _ in range(G)]
    bs = [torch.zeros(Cout).cuda() for _ in range(G)]
    kb = torch.ones(4, 4).cuda() / 16
    agc.check_status()                                           # clean before
    good = gr.grouped_conv_layer(x, ws, bs, kb, 0.05, False)
    torch.cuda.synchronize()
    agc.check_status()
    assert bool(torch.isfinite(good).all())
    stale = torch.zeros(gr._OUT_MAXIMA_FLOATS, device="cuda")
    stale.view(-1, 256)[:G] = float(x.abs().max()) / 256.0       # as if x had grown by 2^8 since its producer measured it
    x._ag_maxima, x._ag_maxima_version = stale, x._version
    bad = gr.grouped_conv_layer(x, ws, bs, kb, 0.05, False)
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(bad).all())                   # the overflow happened ...
    with pytest.raises(_lib.AgNativeError, match="code -5"):     # ... and is reported
        agc.check_status()
    del x._ag_maxima
    again = gr.grouped_conv_layer(x, ws, bs, kb, 0.05, False)    # flag cleared: business as usual, same bits as before
    torch.cuda.synchronize()
    agc.check_status()
    assert torch.equal(again, good)
    # the NEXT convolution call reports by itself when nobody asked in between
    x._ag_maxima, x._ag_maxima_version = stale, x._version
    gr.grouped_conv_layer(x, ws, bs, kb, 0.05, False)
    torch.cuda.synchronize()
    with pytest.raises(_lib.AgNativeError, match="code -5"):
        gr.grouped_conv_layer(x, ws, bs, kb, 0.05, False)
    del x._ag_maxima
    agc.check_status()


def test_multi_view_gradients_are_the_sum_over_the_views(net):
    """Three views of one pose through the grouped chain (the colour network's view-dependent tail runs once per view ON THE SAME PARAMETERS; round 5:
    grouped._merge_shared sums the members' parameter gradients inside the layer's backward instead of leaving ~40 pairwise additions per extra
    view to autograd): every probed parameter gradient equals the sum of the three single-view passes' gradients, up to summation order."""
    import torch
    items = _items(net)
    net.get_pose_map(items)
    net.eval()
    G = net._grouped_nets()
    assert G is not None
    pose = items['smpl_pos_map'][:3][None].contiguous()
    styles = [net.position_style, net.color_style, net.other_style]
    gen = torch.Generator().manual_seed(21)
    feats = [tuple((torch.randn(1, 128, 128, 128, generator=gen) * 0.1).cuda() for _ in range(2)) for _ in range(3)]
    ups_c = [torch.randn(1, 6, 1024, 1024, generator=gen).cuda() for _ in range(3)]
    ups_p, ups_o = torch.randn(1, 6, 1024, 1024, generator=gen).cuda(), torch.randn(1, 16, 1024, 1024, generator=gen).cuda()
    probe = ["convs1.10.conv.weight", "convs2.11.conv.weight", "convs1.11.conv.modulation.weight", "convs2.10.noise.weight", "convs1.11.activate.bias",
             "to_rgbs1.5.conv.weight", "to_rgbs2.5.bias", "to_rgbs1.5.conv.modulation.bias", "comb_convs.0.0.weight", "comb_convs.0.1.bias",
             "convs1.8.conv.weight", "comb_convs.2.0.weight", "cond_convs.1.conv1.0.weight", "style.1.weight"]

    def grads():
        return {k: net.color_net._p(k).grad.clone() for k in probe}
    net.zero_grad(set_to_none=True)
    pm, cms, om = G.forward(styles, pose, {1: feats})
    assert isinstance(cms, list) and len(cms) == 3
    torch.autograd.backward([pm, om] + cms, [ups_p, ups_o] + ups_c)
    multi = grads()
    net.zero_grad(set_to_none=True)
    for v in range(3):
        pm1, cm1, om1 = G.forward(styles, pose, {1: feats[v]})
        if v == 0:
            torch.autograd.backward([pm1, om1, cm1], [ups_p, ups_o, ups_c[v]])
        else:
            cm1.backward(ups_c[v])
    single = grads()
    net.zero_grad(set_to_none=True)
    worst = 0.0
    for k in probe:
        scale = float(single[k].abs().max())
        err = float((multi[k] - single[k]).abs().max()) / max(scale, 1e-30)
        worst = max(worst, err)
        assert err <= (2e-2 if k.endswith("noise.weight") else 5e-3), (k, err)
    print(f"three views at once vs three single-view passes: worst relative parameter-gradient deviation {worst:.2e} over {len(probe)} probes")
