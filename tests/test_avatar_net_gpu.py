"""AvatarNet re-host (SURVEY.md §8 B5, rows a4-a6 glue) on the GPU.

The three DualStyleUNets, the assembly / LBS kernels and the rasterizer each have their own parity tests; this file
checks the wiring around them against literal restatements of network/avatar.py (oracle/avatar_oracle.py):
pose-map and view-direction producers, the composition inside render(), state_dict loading, and that a training-mode
backward reaches every learnable tensor."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _items(net, S=1024, seed=3):
    import torch
    from animatablegaussians_amd import camera
    g = torch.Generator().manual_seed(seed)
    J = net.lbs.shape[1]
    ax = torch.nn.functional.normalize(torch.randn(J, 3, generator=g))
    ang = torch.rand(J, generator=g) * (np.pi / 6)
    K = torch.zeros(J, 3, 3)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -ax[:, 2], ax[:, 1], ax[:, 2], -ax[:, 0], -ax[:, 1], ax[:, 0]
    Rm = torch.eye(3)[None] + torch.sin(ang)[:, None, None] * K + (1 - torch.cos(ang))[:, None, None] * (K @ K)
    A = torch.eye(4)[None].repeat(J, 1, 1)
    A[:, :3, :3] = Rm
    A[:, :3, 3] = (torch.rand(J, 3, generator=g) - 0.5) * 0.02
    extr = torch.from_numpy(camera.calc_front_mv(np.zeros(3, np.float32), tar_pos=(0.0, 0.0, 2.5)))
    intr = torch.tensor([[1100.0, 0, S / 2], [0, 1100.0, S / 2], [0, 0, 1]])
    return {'cano2live_jnt_mats': A.cuda(), 'cano2live_jnt_mats_woRoot': A.cuda(), 'extr': extr.cuda(), 'intr': intr.cuda(),
            'img_w': S, 'img_h': S}


@pytest.fixture(scope="module")
def net():
    import torch
    from animatablegaussians_amd.avatar import AvatarNet
    torch.manual_seed(31359)
    return AvatarNet.synthetic({'with_viewdirs': True})


def test_pose_map_and_viewdir_features_match_reference_formulation(net):
    import torch
    from animatablegaussians_amd import synth
    from oracle import avatar_oracle as ao
    items = _items(net)
    S = 1024
    mask = net.cano_smpl_mask.cpu()
    cano = torch.zeros(S, 2 * S, 3)
    cano[mask] = net.init_points.cpu()
    nml = torch.zeros(S, 2 * S, 3)
    nml[mask] = net.cano_nmls.cpu()
    A = items['cano2live_jnt_mats'].cpu()
    ref_pose = ao.get_pose_map(cano, mask, net.lbs.cpu(), A)
    got_pose = net.get_pose_map(items)
    assert got_pose.shape == (6, 512, 512) and items['smpl_pos_map'] is got_pose
    np.testing.assert_allclose(got_pose.cpu().numpy(), ref_pose.numpy(), rtol=1e-5, atol=1e-6)

    net.eval()
    w = [getattr(net.viewdir_net[i], k).detach().cpu() for i in (0, 2) for k in ("weight", "bias")]
    ref_f, ref_b = ao.get_viewdir_feat(cano, nml, mask, net.lbs.cpu(), A, items['extr'].cpu(), *w)
    got_f, got_b = net.get_viewdir_feat(items)
    assert got_f.shape == (1, 128, 128, 128)
    for got, ref in ((got_f, ref_f), (got_b, ref_b)):
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-5 * float(ref.abs().max()))
    del synth


def test_render_composes_the_verified_pieces_and_trains(net):
    import torch
    from animatablegaussians_amd import avatar_ops as ops
    from animatablegaussians_amd.gaussian_renderer import render3
    items = _items(net)
    net.get_pose_map(items)
    net.eval()
    with torch.no_grad():
        out = net.render(items, bg_color=(0.1, 0.2, 0.3))
        assert set(out) == {'rgb_map', 'mask_map', 'offset', 'pos_map', 'cano_tex_map', 'posed_gaussians'}
        assert out['rgb_map'].shape == (1024, 1024, 3) and out['mask_map'].shape == (1024, 1024, 1)
        assert out['pos_map'].shape == (1024, 2048, 3) and out['cano_tex_map'].shape == (1024, 2048, 3)
        # the same computation from its parts
        fv, bv = net.get_viewdir_feat(items)
        pm, om, cm = net.get_maps(items['smpl_pos_map'][:3], fv, bv)
        g = net.core.assemble(pm, om, cm)
        np.testing.assert_array_equal((g['positions'] - net.init_points).cpu().numpy(), out['offset'].cpu().numpy())
        # pos_map is the raw position canvas: positions = 0.05 * canvas[mask] + xyz  (network/avatar.py:97-101)
        np.testing.assert_allclose((0.05 * out['pos_map'][net.cano_smpl_mask] + net.init_points).cpu().numpy(),
                                   g['positions'].cpu().numpy(), rtol=1e-6, atol=1e-7)
        g['positions'], g['rotations'] = ops.lbs_transform(g['positions'], g['rotations'], net.lbs, items['cano2live_jnt_mats'])
        r = render3(g, torch.tensor([0.1, 0.2, 0.3]).cuda(), items['extr'], items['intr'], 1024, 1024)
        np.testing.assert_array_equal(r['render'].permute(1, 2, 0).cpu().numpy(), out['rgb_map'].cpu().numpy())
        cover = float((out['mask_map'] > 0.5).float().mean())
        assert 0.03 < cover < 0.6, cover                                   # the subject is in view

    # training mode: one loss, every learnable tensor receives a finite, not-all-zero gradient
    net.train()
    out = net.render(items, bg_color=(0., 0., 0.))
    assert set(out) == {'rgb_map', 'mask_map', 'offset', 'pos_map'}
    target = torch.rand(1024, 1024, 3, generator=torch.Generator().manual_seed(5)).cuda()
    loss = (out['rgb_map'] - target).abs().mean() + 0.005 * torch.linalg.norm(out['offset'], dim=-1).mean()
    loss.backward()
    dead = []
    for name, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        if float(p.grad.abs().max()) == 0.0:
            dead.append(name)
    # the two wavelet-skip ToRGB paths of the coarsest stages still reach the output; nothing may be disconnected
    assert not dead, dead[:10]
    net.zero_grad(set_to_none=True)


def test_render_views_shares_the_pose_dependent_work_without_changing_results(net):
    """render_views == render per view (eval: bit-identical images; training: summed gradients agree)."""
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
            for key in ('rgb_map', 'mask_map', 'offset', 'pos_map', 'cano_tex_map'):
                assert torch.equal(single[key], m[key]), key
    # gradients: sum over views of separate backward passes vs one shared backward.  eval mode keeps the view-direction
    # jitter off so both sides see the same features; autograd does not care about the mode.
    probe = [("position_net", "convs1.0.conv.weight"), ("color_net", "convs1.2.activate.bias"),
             ("color_net", "convs2.10.conv.weight"), ("other_net", "to_rgbs1.5.bias"), ("color_net", "conv_in.1.weight")]
    target = torch.rand(1024, 1024, 3, generator=torch.Generator().manual_seed(5)).cuda()

    def loss_of(r):
        return (r['rgb_map'] - target).abs().mean() + 0.005 * torch.linalg.norm(r['offset'], dim=-1).mean()

    net.zero_grad(set_to_none=True)
    for v in views:
        loss_of(net.render({**pose_items, **v})).backward()
    want = {k: getattr(net, k[0])._p(k[1]).grad.clone() for k in probe}
    want_vd = net.viewdir_net[0].weight.grad.clone()
    net.zero_grad(set_to_none=True)
    sum(loss_of(r) for r in net.render_views(pose_items, views)).backward()
    for k in probe:
        got = getattr(net, k[0])._p(k[1]).grad
        scale = float(want[k].abs().max())
        assert float((got - want[k]).abs().max()) <= 2e-3 * scale + 1e-12, (k, float((got - want[k]).abs().max()), scale)
    assert float((net.viewdir_net[0].weight.grad - want_vd).abs().max()) <= 2e-3 * float(want_vd.abs().max())
    net.zero_grad(set_to_none=True)


def test_graph_captured_networks_reproduce_eager_results(net):
    """enable_graphs(): the hipGraph replays of the three networks (and of the shared / per-view colour decoder parts)
    give bit-identical renders, also after the parameters are updated in place."""
    import torch
    from animatablegaussians_amd import synth
    base = _items(net)
    net.get_pose_map(base)
    cams = synth.free_view_cameras(2, img=1024)
    views = [{'extr': torch.from_numpy(np.ascontiguousarray(c["extr"])).float().cuda(),
              'intr': torch.from_numpy(np.ascontiguousarray(c["intr"])).float().cuda(), 'img_w': 1024, 'img_h': 1024} for c in cams]
    net.eval()
    try:
        with torch.no_grad():
            eager1 = net.render(base)['rgb_map'].clone()
            eagerv = [r['rgb_map'].clone() for r in net.render_views(base, views)]
            net.enable_graphs(True)
            for _ in range(2):                      # first call captures, second replays
                assert torch.equal(net.render(base)['rgb_map'], eager1)
                got = net.render_views(base, views)
                assert all(torch.equal(g['rgb_map'], e) for g, e in zip(got, eagerv))
            # in-place parameter update: the captures see it
            p = net.color_net._p("to_rgbs1.5.bias")
            p.add_(0.05)
            changed = net.render(base)['rgb_map']
            net.enable_graphs(False)
            assert torch.equal(net.render(base)['rgb_map'], changed) and not torch.equal(changed, eager1)
            p.sub_(0.05)
    finally:
        net.enable_graphs(False)


def test_reference_state_dict_roundtrip(net):
    import torch
    const = (".kernel", ".ll", ".lh", ".hl", ".hh")
    full = {k: v.detach().clone() for k, v in net.state_dict().items()}
    assert "color_net.conv_in.0.kernel" in full and "viewdir_net.2.bias" in full and "other_net.noises.noise_11" in full
    sd = {k: (v if k.endswith(const) else v + 1.0) for k, v in full.items()}
    before = float(net.color_net._p("style.1.bias").detach().sum())
    net.load_state_dict(sd, strict=True)                                  # what main_avatar.py:797 does
    assert abs(float(net.color_net._p("style.1.bias").detach().sum()) - (before + 512)) < 1e-2
    assert abs(float(net.viewdir_net[0].bias.detach().sum()) - float(full["viewdir_net.0.bias"].sum()) - 64) < 1e-3
    with pytest.raises(RuntimeError):
        net.load_reference_state_dict({**sd, "bogus.weight": torch.zeros(1)})
    # a checkpoint without the constant buffers (written by round-1 code) still loads through load_reference_state_dict
    net.load_reference_state_dict({k: v for k, v in full.items() if not k.endswith(const)})
    for k, v in net.state_dict().items():
        assert torch.equal(v, full[k]), k


def test_fix_hand_fades_the_hands_into_the_mean_hands_frame(net):
    """Test-time `fix_hand` (network/avatar.py:52-77,183-200): generate_mean_hands stores the Gaussians of one frame, render
    cross-fades positions / opacity / scales / rotations into them inside the MANO boxes; opacity and scales (untouched by the
    skinning) must equal the torch restatement of the blend."""
    import torch
    from oracle import avatar_oracle as ao
    items = _items(net)
    net.get_pose_map(items)
    pose_b = items['smpl_pos_map']
    pose_a = pose_b.flip(-1).contiguous()                              # "another frame"
    xyz = net.init_points
    lo, hi = xyz.min(0)[0], xyz.max(0)[0]
    g = torch.Generator().manual_seed(8)
    box = lambda cx: (torch.tensor([cx, float(hi[1]) - 0.2, 0.0]) + (torch.rand(778, 3, generator=g) - 0.5) * torch.tensor([0.2, 0.12, 0.1])).cuda()  # noqa: E731
    items.update({'left_cano_mano_v': box(float(hi[0]) - 0.1), 'right_cano_mano_v': box(float(lo[0]) + 0.1),
                  'cano_smpl_center': (0.5 * (lo + hi))})
    net.eval()
    old = net.opt.get('fix_hand', False)
    try:
        with torch.no_grad():
            net.opt['fix_hand'] = False
            plain = net.render(items)['posed_gaussians']
            net.opt['fix_hand'] = True
            with pytest.raises(RuntimeError):
                net.hand_positions = None
                net.render(items)
            net.generate_mean_hands(pose_a)
            assert net.hand_mask.dtype == torch.bool and net.hand_mask.shape == (xyz.shape[0],)
            fused = net.render(items)['posed_gaussians']
            fv, bv = net.get_viewdir_feat(items)
            cur = net.core.assemble(*net.get_maps(pose_b[:3], fv, bv))
        hand = {'positions': net.hand_positions, 'opacity': net.hand_opacity, 'scales': net.hand_scales, 'rotations': net.hand_rotations}
        ref, w = ao.hand_fuse({k: cur[k].cpu() for k in hand}, xyz.cpu(), items['left_cano_mano_v'].cpu(), items['right_cano_mano_v'].cpu(),
                              items['cano_smpl_center'].cpu(), {k: v.cpu() for k, v in hand.items()})
        assert float((w > 0.5).float().mean()) > 0.005
        for k in ('opacity', 'scales'):
            np.testing.assert_allclose(fused[k].cpu().numpy(), ref[k].numpy(), rtol=1e-5, atol=1e-7)
            far = (w[:, 0] == 0)
            assert torch.equal(fused[k][far.cuda()], plain[k][far.cuda()])
        assert not torch.equal(fused['opacity'], plain['opacity'])
    finally:
        net.opt['fix_hand'] = old
