"""The surface of ``AvatarNet`` the REFERENCE TRAINER touches (SURVEY.md §1 "L5 -> L3", §8 B5): constructed the way
``main_avatar.py:45-48`` constructs it -- ``importlib.import_module(opt['model']['module']).AvatarNet(opt['model'])`` with the
drop-in module, a ``config`` module holding the data directory, assets read from EXR / NPY files -- and then driven by the bodies
of ``AvatarTrainer.forward_one_pass_pretrain`` (main_avatar.py:126-160) and ``forward_one_pass`` (:162-264), restated below
statement for statement.  Values are checked against ``oracle/avatar_oracle.py`` (the literal restatement of network/avatar.py)
and the torch CPU restatement of the loss tail."""
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "animatablegaussians_amd", "dropin")


def _write_subject(data_dir, S=1024, hand_frame=7):
    """The per-subject files network/avatar.py:27-43,61 reads, from the synthetic subject."""
    import torch
    from animatablegaussians_amd import exr
    from animatablegaussians_amd.avatar import AvatarNet
    src = AvatarNet.synthetic({'with_viewdirs': True}, S=S, device='cpu')
    d = os.path.join(data_dir, 'smpl_pos_map')
    os.makedirs(d, exist_ok=True)
    mask = src.cano_smpl_mask
    cano = torch.zeros(S, 2 * S, 3)
    cano[mask] = src.init_points
    nml = torch.zeros(S, 2 * S, 3)
    nml[mask] = src.cano_nmls
    exr.imwrite(os.path.join(d, 'cano_smpl_pos_map.exr'), cano.numpy())
    exr.imwrite(os.path.join(d, 'cano_smpl_nml_map.exr'), nml.numpy())
    np.save(os.path.join(d, 'init_pts_lbs.npy'), src.lbs.numpy())
    live = np.random.RandomState(hand_frame).standard_normal((S // 2, S, 3)).astype(np.float32) * 0.3   # [H, 2W, 3] front|back
    exr.imwrite(os.path.join(d, '%08d.exr' % hand_frame), live)
    return live


class _Config:
    """Installs a stand-in for the reference's global ``config`` module (config.py: ``opt``, ``device``)."""

    def __init__(self, data_dir, device, mode='train', fix_hand=False):
        self.mod = types.ModuleType("config")
        self.mod.opt = {'mode': mode, 'train': {'data': {'data_dir': data_dir}}, 'test': {'fix_hand': fix_hand, 'fix_hand_id': 7},
                        'model': {'module': 'avatar_module', 'with_viewdirs': True, 'random_style': False}}
        self.mod.device = device

    def __enter__(self):
        self.saved = sys.modules.get("config")
        sys.modules["config"] = self.mod
        sys.path.insert(0, DROPIN)
        return self.mod

    def __exit__(self, *exc):
        sys.path.remove(DROPIN)
        sys.modules.pop("avatar_module", None)
        if self.saved is not None:
            sys.modules["config"] = self.saved
        else:
            sys.modules.pop("config", None)


def test_trainer_facing_attributes_exist(tmp_path):
    """CPU: every attribute SURVEY.md §1 'L5 -> L3' lists, on the drop-in class built the reference trainer's way.  Properties
    that would launch a kernel are looked up on the type."""
    import importlib
    import torch
    live = _write_subject(str(tmp_path), S=64)
    with _Config(str(tmp_path), 'cpu', mode='test', fix_hand=True) as config:
        AvatarNet = importlib.import_module(config.opt['model']['module']).AvatarNet          # main_avatar.py:45-47
        net = AvatarNet(config.opt['model']).to(config.device)                               # :48
        for name in ("render", "get_positions", "get_others", "get_colors", "get_viewdir_feat", "get_pose_map", "transform_cano2live",
                     "generate_mean_hands", "state_dict", "load_state_dict", "parameters", "train", "eval"):
            assert callable(getattr(net, name)), name
        for name in ("color_net", "position_net", "other_net", "viewdir_net"):                # main_avatar.py:185-187
            assert isinstance(getattr(net, name), torch.nn.Module), name
        assert net.init_points.shape[1] == 3 and net.lbs.shape[0] == net.init_points.shape[0]                     # :484
        assert net.cano_smpl_mask.dtype == torch.bool and net.max_sh_degree == 0 and net.with_viewdirs and not net.random_style
        cgm = net.cano_gaussian_model                                                         # main_avatar.py:134-153
        assert cgm.get_xyz is net.init_points
        for name in ("get_xyz", "get_opacity", "get_scaling", "get_rotation", "get_opacity_raw", "get_scaling_raw", "get_rotation_raw"):
            assert isinstance(getattr(type(cgm), name), property), name
        assert cgm.get_opacity_raw.shape == (net.init_points.shape[0], 1) and cgm.get_rotation_raw.shape[1] == 4
        x = torch.tensor([[0.3, -0.2, 0.1, 0.9]])
        assert torch.equal(cgm.opacity_activation(x), torch.sigmoid(x)) and torch.equal(cgm.scaling_activation(x), torch.exp(x))
        assert torch.allclose(cgm.rotation_activation(x).norm(dim=-1), torch.ones(1))
        # requires_net_grad(net, flag) of the trainer iterates .parameters() of the sub-networks (main_avatar.py:67-73)
        assert sum(p.numel() for p in net.color_net.parameters()) == 74479016
        # test-time hand fusion is switched by config.opt['test'] + mode, not by the model opt (network/avatar.py:183)
        assert net._fix_hand_enabled()
        config.opt['mode'] = 'train'
        assert not net._fix_hand_enabled()
        pm = net._fix_hand_pose_map()                                                         # network/avatar.py:61-67
        assert pm.shape == (3, 32, 32)
        np.testing.assert_array_equal(pm.numpy(), live[:, :32].transpose(2, 0, 1))


@pytest.fixture(scope="module")
def trainer_net(tmp_path_factory):
    import importlib
    import torch
    d = str(tmp_path_factory.mktemp("subject"))
    _write_subject(d)
    torch.manual_seed(31359)
    with _Config(d, 'cuda:0') as config:
        AvatarNet = importlib.import_module(config.opt['model']['module']).AvatarNet
        net = AvatarNet(config.opt['model']).to(config.device)
        yield net, config


def _pose_items(net, seed=3):
    sys.path.insert(0, os.path.dirname(__file__))
    from test_avatar_net_gpu import _items
    items = _items(net, seed=seed)
    net.get_pose_map(items)
    return items


@pytest.mark.gpu
def test_forward_one_pass_pretrain_body(trainer_net):
    """main_avatar.py:126-160 against the drop-in class; the per-part values against avatar_oracle on the same network outputs."""
    import torch
    from oracle import avatar_oracle as ao
    net, _ = trainer_net
    net.train()
    optm = torch.optim.Adam(net.parameters(), lr=5e-4)
    items = _pose_items(net)

    # ---- the reference's statements -----------------------------------------------------------------------------------------
    total_loss = 0
    batch_losses = {}
    l1_loss = torch.nn.L1Loss()
    pose_map = items['smpl_pos_map'][:3]
    position_loss = l1_loss(net.get_positions(pose_map), net.cano_gaussian_model.get_xyz)
    total_loss += position_loss
    batch_losses.update({'position': position_loss.item()})
    opacity, scales, rotations = net.get_others(pose_map)
    opacity_loss = l1_loss(opacity, net.cano_gaussian_model.get_opacity)
    total_loss += opacity_loss
    batch_losses.update({'opacity': opacity_loss.item()})
    scale_loss = l1_loss(scales, net.cano_gaussian_model.get_scaling)
    total_loss += scale_loss
    batch_losses.update({'scale': scale_loss.item()})
    rotation_loss = l1_loss(rotations, net.cano_gaussian_model.get_rotation)
    total_loss += rotation_loss
    batch_losses.update({'rotation': rotation_loss.item()})
    total_loss.backward()
    probe = net.position_net._p("convs1.0.conv.weight")
    g_probe, before = probe.grad.clone(), probe.detach().clone()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n in (net.position_net, net.other_net) for p in n.parameters())
    assert all(p.grad is None for p in net.color_net.parameters())          # the colour network is not part of this pass
    optm.step()
    optm.zero_grad()
    # ---------------------------------------------------------------------------------------------------------------------------
    assert float(g_probe.abs().max()) > 0 and not torch.equal(probe.detach(), before) and all(np.isfinite(v) for v in batch_losses.values())

    # values: the same network outputs through the oracle's get_positions / get_others, and GaussianModel's getters
    net.eval()
    mask = net.cano_smpl_mask.cpu()
    cgm = net.cano_gaussian_model
    xyz, o_raw, s_raw, r_raw = (t.cpu() for t in (cgm.get_xyz, cgm.get_opacity_raw, cgm.get_scaling_raw, cgm.get_rotation_raw))
    with torch.no_grad():
        x = pose_map[None].contiguous()
        pm = net.position_net([net.position_style], x, randomize_noise=False)[0]
        om = net.other_net([net.other_style], x, randomize_noise=False)[0]
        fv, bv = net.get_viewdir_feat(items)
        cm = net.color_net([net.color_style], x, randomize_noise=False, view_feature1=fv, view_feature2=bv)[0]
        got_pos, got_pmap = net.get_positions(pose_map, return_map=True)
        got_o, got_s, got_r = net.get_others(pose_map)
        got_c, got_cmap = net.get_colors(pose_map, fv, bv)
    ref = ao.gather_activate(pm.cpu(), om.cpu(), cm.cpu(), mask, xyz, o_raw, s_raw, r_raw)
    for name, got, want in zip(("positions", "opacity", "scales", "rotations", "colors"), (got_pos, got_o, got_s, got_r, got_c), ref):
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-6, err_msg=name)
    np.testing.assert_array_equal(got_pmap.cpu().numpy(), ao.canvas(pm.cpu(), 3).numpy())
    np.testing.assert_array_equal(got_cmap.cpu().numpy(), ao.canvas(cm.cpu(), 3).numpy())
    np.testing.assert_allclose(cgm.get_opacity.cpu().numpy(), torch.sigmoid(o_raw).numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(cgm.get_scaling.cpu().numpy(), torch.exp(s_raw).numpy(), rtol=1e-6)
    np.testing.assert_allclose(cgm.get_rotation.cpu().numpy(), torch.nn.functional.normalize(r_raw).numpy(), rtol=1e-6, atol=1e-7)
    assert cgm.get_opacity is cgm.get_opacity                                       # cached until a raw buffer changes

    # gradients of the part-wise accessors == autograd of the oracle's expressions
    from animatablegaussians_amd import avatar_ops as ops
    gen = torch.Generator().manual_seed(11)
    leaf = {k: (v.detach().clone().requires_grad_(True), v.detach().cpu().clone().requires_grad_(True)) for k, v in (("p", pm), ("o", om), ("c", cm))}
    N = xyz.shape[0]
    G = [torch.randn(N, c, generator=gen) for c in (3, 1, 3, 4, 3)]
    core = net.core
    outs = (ops.gather_positions(leaf["p"][0], core.pix, core.xyz),) + tuple(ops.gather_others(leaf["o"][0], core.pix, core.opacity_raw, core.scaling_raw, core.rotation_raw)) \
        + (ops.gather_colors(leaf["c"][0], core.pix),)
    sum((o * g.cuda()).sum() for o, g in zip(outs, G)).backward()
    refs = ao.gather_activate(leaf["p"][1], leaf["o"][1], leaf["c"][1], mask, xyz, o_raw, s_raw, r_raw)
    sum((o * g).sum() for o, g in zip(refs, G)).backward()
    for k in ("p", "o", "c"):
        want = leaf[k][1].grad.numpy()
        np.testing.assert_allclose(leaf[k][0].grad.cpu().numpy(), want, rtol=1e-5, atol=1e-6 * np.abs(want).max(), err_msg=k)

    # transform_cano2live mutates and returns the dict (network/avatar.py:84-91)
    vals = {'positions': got_pos.clone(), 'rotations': got_r.clone(), 'opacity': got_o}
    back = net.transform_cano2live(vals, items)
    assert back is vals
    lp, lr = ao.transform_cano2live(ref[0], ref[3], net.lbs.cpu(), items['cano2live_jnt_mats'].cpu())
    np.testing.assert_allclose(vals['positions'].cpu().numpy(), lp.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(vals['rotations'].cpu().numpy(), lr.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_forward_one_pass_body_with_the_full_loss_tail(trainer_net):
    """main_avatar.py:162-264: render, boundary compositing, L1 + mask + LPIPS (512^2 crop) + offset losses, backward, Adam step --
    the loss parts recomputed on the CPU from the rendered maps (torch restatement + oracle/lpips_oracle)."""
    import torch
    from animatablegaussians_amd import losses
    from animatablegaussians_amd.lpips import LPIPS, lpips_named_fill
    from oracle import lpips_oracle as lo
    net, config = trainer_net
    net.train()
    optm = torch.optim.Adam(net.parameters(), lr=5e-4)
    lp = LPIPS(net='vgg')
    lp_sd = lpips_named_fill({k: v for k, v in lp.reference_state_dict().items() if not k.startswith("scaling_layer")})
    lp.load_reference_state_dict({**lp_sd, "scaling_layer.shift": lp.scaling_layer__shift, "scaling_layer.scale": lp.scaling_layer__scale})
    lp = lp.cuda()
    loss_weight = {'l1': 1.0, 'lpips': 0.1, 'offset': 0.005, 'mask': 0.1}          # configs/avatarrex_zzr/avatar.yaml
    items = _pose_items(net, seed=4)
    H = W = 1024
    gen = torch.Generator().manual_seed(9)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    mask_img = ((yy - 520.0) ** 2 / 400.0 ** 2 + (xx - 500.0) ** 2 / 230.0 ** 2) < 1.0
    ring = ((yy - 520.0) ** 2 / 410.0 ** 2 + (xx - 500.0) ** 2 / 240.0 ** 2) < 1.0
    items.update({'color_img': torch.rand(H, W, 3, generator=gen).cuda(), 'mask_img': mask_img.cuda(), 'boundary_mask_img': (ring & ~mask_img).cuda()})
    bg_color = (0.2, 0.5, 0.7)
    bg_color_cuda = torch.tensor(bg_color).cuda()
    before = net.color_net._p("convs1.10.conv.weight").detach().clone()

    # ---- the reference's statements (render + loss via losses.training_loss, which restates :196-245) ---------------------------
    render_output = net.render(items, bg_color)
    total_loss, parts = losses.training_loss(render_output, items, bg_color_cuda, loss_weight, lpips=lp, patch_size=512, random_patch=False)
    total_loss.backward()
    grads_ok = all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    optm.step()
    optm.zero_grad()
    # ---------------------------------------------------------------------------------------------------------------------------
    assert grads_ok and not torch.equal(net.color_net._p("convs1.10.conv.weight").detach(), before)
    assert set(render_output) == {'rgb_map', 'mask_map', 'offset', 'pos_map'} and set(parts) == {'l1_loss', 'mask_loss', 'lpips_loss', 'offset_loss'}

    # CPU restatement of main_avatar.py:196-245 on the rendered maps
    image = render_output['rgb_map'].detach().cpu().permute(2, 0, 1)
    bgc = bg_color_cuda.cpu()
    color = items['color_img'].cpu().clone()
    color[~mask_img] = bgc
    gt_image = color.permute(2, 0, 1)
    m32 = mask_img.to(torch.float32)
    boundary = 1. - items['boundary_mask_img'].cpu().to(torch.float32)
    image = image * boundary[None] + (1. - boundary[None]) * bgc[:, None, None]
    gt_image = gt_image * boundary[None] + (1. - boundary[None]) * bgc[:, None, None]
    l1 = torch.abs(image - gt_image).mean()
    ml = torch.abs(render_output['mask_map'].detach().cpu().squeeze(-1) * boundary - m32 * boundary).mean()
    uv = torch.argwhere(m32 > 0.)                                                      # crop_image, :75-115 (not random below 300k its)
    (min_v, min_u), (max_v, max_u) = uv.min(0)[0].tolist(), uv.max(0)[0].tolist()
    len_v, len_u = max_v - min_v, max_u - min_u
    size = max(len_v, len_u)
    crops = []
    for im in (image, gt_image):
        c = bgc[:, None, None] * torch.ones(3, size, size)
        if len_v > len_u:
            s = (size - len_u) // 2
            c[:, :, s:s + len_u] = im[:, min_v:max_v, min_u:max_u]
        else:
            s = (size - len_v) // 2
            c[:, s:s + len_v, :] = im[:, min_v:max_v, min_u:max_u]
        crops.append(torch.nn.functional.interpolate(c[None], size=(512, 512), mode='bilinear')[0])
    lpv, _ = lo.lpips(crops[0][None, [2, 1, 0]], crops[1][None, [2, 1, 0]], lp_sd, normalize=True)
    off = torch.linalg.norm(render_output['offset'].detach().cpu(), dim=-1).mean()
    for name, want in (("l1_loss", l1), ("mask_loss", ml), ("lpips_loss", lpv.mean()), ("offset_loss", off)):
        got = float(parts[name])
        assert abs(got - float(want)) <= 2e-4 * abs(float(want)) + 1e-7, (name, got, float(want))
    want_total = loss_weight['l1'] * l1 + loss_weight['mask'] * ml + loss_weight['lpips'] * lpv.mean() + loss_weight['offset'] * off
    assert abs(float(total_loss) - float(want_total)) <= 2e-4 * abs(float(want_total))
    del config
