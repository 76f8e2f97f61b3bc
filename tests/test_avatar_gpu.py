"""GPU parity of the gather/activation and LBS kernels against the torch-CPU oracle (oracle/avatar_oracle.py).

fp32 tolerance: 1e-5 relative to the row magnitude for forward values (a handful of fused fp32 ops; the LBS blend is a
55-term dot product whose summation order differs from the oracle's BLAS matmul), 1e-4 for gradients."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _synthetic(S=256, J=55, seed=0):
    import torch
    from animatablegaussians_amd import synth
    g = torch.Generator().manual_seed(seed)
    m = synth.body_mask(S)
    mask = torch.from_numpy(np.concatenate([m, m[:, ::-1]], axis=1).copy())
    N = int(mask.sum())
    d = dict(
        mask=mask,
        position_map=torch.randn(1, 6, S, S, generator=g),
        other_map=torch.randn(1, 16, S, S, generator=g) * 0.5,
        color_map=torch.rand(1, 6, S, S, generator=g),
        xyz=torch.randn(N, 3, generator=g) * 0.5,
        opacity_raw=torch.randn(N, 1, generator=g),
        scaling_raw=torch.randn(N, 3, generator=g) * 0.3 - 5.0,
        rotation_raw=torch.nn.functional.normalize(torch.randn(N, 4, generator=g)),
    )
    w = torch.softmax(torch.randn(N, J, generator=g) * 4, dim=1)
    top = torch.topk(w, 4, dim=1)
    lbs = torch.zeros_like(w).scatter_(1, top.indices, top.values)
    d["lbs"] = lbs / lbs.sum(1, keepdim=True)
    # random rigid joint transforms: rotations <= 30 deg, translations <= 5 cm
    ax = torch.nn.functional.normalize(torch.randn(J, 3, generator=g))
    ang = torch.rand(J, generator=g) * (np.pi / 6)
    K = torch.zeros(J, 3, 3)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -ax[:, 2], ax[:, 1], ax[:, 2], -ax[:, 0], -ax[:, 1], ax[:, 0]
    Rm = torch.eye(3)[None] + torch.sin(ang)[:, None, None] * K + (1 - torch.cos(ang))[:, None, None] * (K @ K)
    A = torch.eye(4)[None].repeat(J, 1, 1)
    A[:, :3, :3] = Rm
    A[:, :3, 3] = (torch.rand(J, 3, generator=g) - 0.5) * 0.1
    d["jnt_mats"] = A
    return d


def _close(got, ref, name, rtol):
    got, ref = got.detach().cpu().double().numpy(), ref.detach().double().numpy()
    d = np.abs(got - ref)
    lim = rtol * np.maximum(np.abs(ref), np.abs(ref).max(axis=-1, keepdims=True) if ref.ndim > 1 else np.abs(ref)) + 1e-7
    assert np.isfinite(got).all(), name
    assert (d <= lim).all(), f"{name}: max diff {d.max():.3e}, worst ratio {(d / lim).max():.2f}"


def test_gather_activate_forward_backward():
    import torch
    from animatablegaussians_amd import avatar_ops as ops
    from oracle import avatar_oracle as ao
    d = _synthetic()
    maps = [d[k].clone().requires_grad_(True) for k in ("position_map", "other_map", "color_map")]
    ref = ao.gather_activate(*maps, d["mask"], d["xyz"], d["opacity_raw"], d["scaling_raw"], d["rotation_raw"])
    g = torch.Generator().manual_seed(1)
    ups = [torch.randn(r.shape, generator=g) for r in ref]
    torch.autograd.backward(list(ref), ups)

    gm = [d[k].cuda().requires_grad_(True) for k in ("position_map", "other_map", "color_map")]
    pix = ops.mask_to_pix(d["mask"].cuda())
    got = ops.gather_activate(*gm, pix, d["xyz"].cuda(), d["opacity_raw"].cuda(), d["scaling_raw"].cuda(), d["rotation_raw"].cuda())
    for name, a, b in zip(("positions", "opacity", "scales", "rotations", "colors"), got, ref):
        _close(a, b, name, 1e-5)
    torch.autograd.backward(list(got), [u.cuda() for u in ups])
    for name, a, b in zip(("dL_dposition_map", "dL_dother_map", "dL_dcolor_map"), gm, maps):
        ga, gb = a.grad.cpu(), b.grad
        assert torch.equal(ga == 0, gb == 0) or (ga[gb == 0].abs().max() == 0), name + ": gradient outside the mask"
        np.testing.assert_allclose(ga.numpy(), gb.numpy(), rtol=1e-4, atol=1e-6, err_msg=name)


@pytest.mark.parametrize("J", [55, 24])
def test_lbs_forward_backward(J):
    import torch
    from animatablegaussians_amd import avatar_ops as ops
    from oracle import avatar_oracle as ao
    d = _synthetic(J=J, seed=2)
    N = d["xyz"].shape[0]
    g = torch.Generator().manual_seed(3)
    pos = (torch.randn(N, 3, generator=g) * 0.5).requires_grad_(True)
    rot = torch.nn.functional.normalize(torch.randn(N, 4, generator=g)).requires_grad_(True)
    rp, rr = ao.transform_cano2live(pos, rot, d["lbs"], d["jnt_mats"])
    up, ur = torch.randn(N, 3, generator=g), torch.randn(N, 4, generator=g)
    torch.autograd.backward([rp, rr], [up, ur])

    gpos, grot = pos.detach().cuda().requires_grad_(True), rot.detach().cuda().requires_grad_(True)
    gp, gr = ops.lbs_transform(gpos, grot, d["lbs"].cuda(), d["jnt_mats"].cuda())
    _close(gp, rp, "live positions", 1e-5)
    _close(gr, rr, "live rotations", 1e-5)
    # blended matrices are not orthonormal: the output quaternions are generally NOT unit (the rasterizer consumes them raw)
    assert (gr.detach().norm(dim=1) - 1).abs().max() > 1e-4
    torch.autograd.backward([gp, gr], [up.cuda(), ur.cuda()])
    _close(gpos.grad, pos.grad, "dL_dpositions", 1e-4)
    _close(grot.grad, rot.grad, "dL_drotations", 1e-4)


def test_lbs_all_matrix_to_quaternion_branches():
    """Force each of the four arg-max candidates of matrix_to_quaternion (rotations by ~180 deg about x, y, z and a
    small one) plus the 0.1 floor (a strongly shrunk blend)."""
    import torch
    from animatablegaussians_amd import avatar_ops as ops
    from oracle import avatar_oracle as ao
    qs = torch.tensor([[1.0, 0.02, 0.01, 0.03], [0.02, 1.0, 0.03, 0.01], [0.01, 0.02, 1.0, 0.03], [0.03, 0.01, 0.02, 1.0]])
    qs = torch.nn.functional.normalize(qs).repeat(16, 1)
    N = qs.shape[0]
    lbs = torch.zeros(N, 3)
    lbs[:, 0] = 1.0
    lbs[N // 2:, 0] = 0.004          # shrinks M3 so every q_abs candidate falls under the 0.1 floor... (identity * 0.004)
    A = torch.eye(4)[None].repeat(3, 1, 1)
    pos = torch.randn(N, 3)
    rot = qs.clone().requires_grad_(True)
    rp, rr = ao.transform_cano2live(pos, rot, lbs, A)
    up = torch.randn(N, 4)
    rr.backward(up)
    grot = qs.clone().cuda().requires_grad_(True)
    gp, gr = ops.lbs_transform(pos.cuda(), grot, lbs.cuda(), A.cuda())
    _close(gr, rr, "rotations", 1e-5)
    gr.backward(up.cuda())
    _close(grot.grad, rot.grad, "dL_drotations", 1e-4)


def test_sparse_lbs_rows_give_the_dense_results():
    """blend-weight rows are mostly exact zeros; the (joint, weight)-pair form reads 5 K instead of 4 J bytes per
    Gaussian and must give the DENSE kernels' results bit for bit (a skipped term is + 0 * A), forward and backward; rows with too
    many non-zeros (weights sampled from the diffused volume) fall back to the dense path."""
    import torch
    from animatablegaussians_amd import avatar_ops as ops
    d = _synthetic(seed=3)
    lbs = d["lbs"].cuda()
    # ragged sparsity: 1..6 non-zeros per row, a few rows with a single joint
    N, J = lbs.shape
    g = torch.Generator().manual_seed(5)
    extra = torch.rand(N, J, generator=g).cuda() * (torch.rand(N, J, generator=g).cuda() < 0.03)
    lbs = lbs + extra
    lbs[::7] = 0
    lbs[::7, 11] = 1.0
    lbs = (lbs / lbs.sum(1, keepdim=True)).contiguous()
    sp = ops.SparseLbs.build(lbs)
    assert sp is not None and 4 <= sp.K <= 16 and sp.idx.dtype == torch.uint8 and tuple(sp.idx.shape) == (sp.K, N)
    # the pairs reproduce the rows
    back = torch.zeros_like(lbs).t().contiguous()
    back.scatter_add_(0, sp.idx.long(), sp.w)
    assert torch.equal(back.t(), lbs)
    assert bool((sp.idx[1:].long() >= sp.idx[:-1].long())[sp.w[1:] != 0].all())           # ascending joints within a row
    pos = d["xyz"].cuda().requires_grad_(True)
    rot = torch.nn.functional.normalize(torch.randn(N, 4, generator=g)).cuda().requires_grad_(True)
    A = d["jnt_mats"].cuda()
    up = [torch.randn(N, 3, generator=g).cuda(), torch.randn(N, 4, generator=g).cuda()]
    outs = {}
    for name, sparse in (("dense", None), ("sparse", sp)):
        pos.grad = rot.grad = None
        p, r = ops.lbs_transform(pos, rot, lbs, A, sparse)
        torch.autograd.backward([p, r], up)
        outs[name] = [t.detach().clone() for t in (p, r, pos.grad, rot.grad)]
    for a, b, what in zip(outs["dense"], outs["sparse"], ("positions", "rotations", "dL_dpositions", "dL_drotations")):
        assert torch.equal(a, b), f"{what}: sparse differs from dense by {float((a - b).abs().max()):.3e}"
    # dense rows: no sparse form
    assert ops.SparseLbs.build(torch.softmax(torch.randn(100, J, generator=g), 1).cuda()) is None
    # the avatar core picks it up by itself
    from animatablegaussians_amd.avatar import AvatarRenderCore
    core = AvatarRenderCore(d["mask"].cuda(), d["xyz"].cuda(), d["opacity_raw"].cuda(), d["scaling_raw"].cuda(), d["rotation_raw"].cuda(),
                            d["lbs"].cuda())
    assert core.lbs_sparse is not None and core.lbs_sparse.K == 4


def test_avatar_render_core_end_to_end():
    """maps -> gather/activations -> LBS -> render3 -> rasterizer, forward and backward, against the composition of the
    torch oracle (assembly + skinning) with the C rasterizer oracle."""
    import torch
    import helpers as h
    from animatablegaussians_amd import camera
    from animatablegaussians_amd.avatar import AvatarRenderCore
    from oracle import avatar_oracle as ao
    from oracle import raster_oracle as ro
    S, W = 256, 256
    d = _synthetic(S=S, seed=7)
    # canonical points on a body-sized sheet so the camera sees them; small scales
    N = d["xyz"].shape[0]
    vv, uu = torch.nonzero(d["mask"], as_tuple=True)
    front = uu < S
    ul = torch.where(front, uu, 2 * S - 1 - uu).float()
    d["xyz"] = torch.stack([(ul + 0.5 - S / 2) * (2.0 / S), (S / 2 - (vv.float() + 0.5)) * (2.0 / S),
                            torch.where(front, 0.05, -0.05) * torch.ones(N)], 1)
    d["scaling_raw"] = torch.full((N, 3), float(np.log(2.0 / S)))
    d["position_map"] = d["position_map"] * 0.2
    d["jnt_mats"][:, :3, 3] *= 0.2
    extr = torch.from_numpy(camera.calc_front_mv(np.zeros(3, np.float32), tar_pos=(0.0, 0.0, 2.5)))
    intr = torch.tensor([[275.0, 0, W / 2], [0, 275.0, W / 2], [0, 0, 1]])
    bg = torch.tensor([0.2, 0.5, 0.8])
    # ---- oracle: forward ----
    maps = [d[k].clone().requires_grad_(True) for k in ("position_map", "other_map", "color_map")]
    pos, opa, sca, rot, col = ao.gather_activate(*maps, d["mask"], d["xyz"], d["opacity_raw"], d["scaling_raw"], d["rotation_raw"])
    lpos, lrot = ao.transform_cano2live(pos, rot, d["lbs"], d["jnt_mats"])
    cm = camera.camera_from_intr_extr(extr.numpy(), intr.numpy(), W, W)
    npy = lambda t: t.detach().numpy()  # noqa: E731
    st = ro.forward(npy(lpos), npy(col), npy(opa), npy(sca), npy(lrot), bg.numpy(), cm["viewmatrix"], cm["projmatrix"],
                    cm["tanfovx"], cm["tanfovy"], W, W)
    assert st["num_rendered"] > 5000
    frag = st["fragile"].astype(bool)
    # ---- GPU: forward ----
    core = AvatarRenderCore(d["mask"].cuda(), d["xyz"].cuda(), d["opacity_raw"].cuda(), d["scaling_raw"].cuda(),
                            d["rotation_raw"].cuda(), d["lbs"].cuda())
    gm = [d[k].cuda().requires_grad_(True) for k in ("position_map", "other_map", "color_map")]
    out = core(*gm, d["jnt_mats"].cuda(), extr.cuda(), intr.cuda(), W, W, bg.cuda())
    rgb = out["rgb_map"].detach().cpu().numpy().transpose(2, 0, 1)
    err = np.abs(rgb - st["color"])[:, ~frag]
    assert err.max() <= 2e-4, f"rgb differs by {err.max()}"       # 1e-4 raster bar + upstream fp32 op-order noise
    np.testing.assert_allclose(out["offset"].detach().cpu().numpy(), npy(pos - d["xyz"]), rtol=1e-5, atol=1e-7)
    # ---- backward: same masked upstream gradients on both sides ----
    g = torch.Generator().manual_seed(9)
    keep = torch.from_numpy((~frag).astype(np.float32))
    g_rgb = torch.randn(W, W, 3, generator=g) * keep[..., None]
    g_msk = torch.randn(W, W, 1, generator=g) * keep[..., None]
    torch.autograd.backward([out["rgb_map"], out["mask_map"]], [g_rgb.cuda(), g_msk.cuda()])
    gr = ro.backward(st, npy(lpos), npy(col), npy(sca), npy(lrot), bg.numpy(), cm["viewmatrix"], cm["projmatrix"], cm["tanfovx"],
                     cm["tanfovy"], g_rgb.permute(2, 0, 1).contiguous().numpy(), np.zeros((1, W, W), np.float32),
                     g_msk.permute(2, 0, 1).contiguous().numpy())
    T = torch.from_numpy
    torch.autograd.backward([lpos, lrot, sca, opa, col],
                            [T(gr["dL_dmeans3D"]), T(gr["dL_drotations"]), T(gr["dL_dscales"]), T(gr["dL_dopacity"]), T(gr["dL_dcolors"])])
    for name, a, b in zip(("position_map", "other_map", "color_map"), gm, maps):
        ga, gb = a.grad.cpu().numpy(), b.grad.numpy()
        dd = np.abs(ga - gb)
        lim = 2e-3 * np.abs(gb) + 2e-4 * np.abs(gb).max()          # coarse: forward-rounding conditioning
        assert (dd > lim).mean() < 2e-3, f"dL/d{name}: {(dd > lim).mean():.2e} of elements off, max {dd.max():.3e}"


def test_hand_fuse_matches_reference_blend():
    """Eval-time hand fusion (network/avatar.py:183-200 + utils/geo_util.py:104-114) against the torch restatement: weights
    rise across the hand boxes, vanish below the body centre, and the four attribute arrays are cross-faded in place."""
    import torch
    from animatablegaussians_amd import avatar_ops as ops
    from oracle import avatar_oracle as ao
    g = torch.Generator().manual_seed(4)
    N = 20000
    xyz = (torch.rand(N, 3, generator=g) - 0.5) * torch.tensor([1.8, 1.8, 0.4])
    left = torch.tensor([0.72, 0.35, 0.0]) + (torch.rand(778, 3, generator=g) - 0.5) * torch.tensor([0.2, 0.1, 0.08])
    right = torch.tensor([-0.72, 0.35, 0.0]) + (torch.rand(778, 3, generator=g) - 0.5) * torch.tensor([0.2, 0.1, 0.08])
    centre = torch.tensor([0.0, 0.1, 0.0])
    cur = {'positions': torch.randn(N, 3, generator=g), 'opacity': torch.rand(N, 1, generator=g),
           'scales': torch.rand(N, 3, generator=g) * 0.01, 'rotations': torch.randn(N, 4, generator=g)}
    hand = {k: torch.randn(v.shape, generator=g) for k, v in cur.items()}
    ref, w = ao.hand_fuse({k: v.clone() for k, v in cur.items()}, xyz, left, right, centre, hand)
    assert 0.02 < float((w > 0.5).float().mean()) < 0.5 and float(w[xyz[:, 1] < centre[1]].abs().max()) == 0.0
    c = lambda t: t.cuda()  # noqa: E731
    got = ops.hand_fuse(c(cur['positions']), c(cur['opacity']), c(cur['scales']), c(cur['rotations']), c(xyz), c(left), c(right),
                        centre, c(hand['positions']), c(hand['opacity']), c(hand['scales']), c(hand['rotations']))
    for k, t in zip(('positions', 'opacity', 'scales', 'rotations'), got):
        np.testing.assert_allclose(t.cpu().numpy(), ref[k].numpy(), rtol=1e-5, atol=2e-6)
    with pytest.raises(RuntimeError):
        ops.hand_fuse(c(cur['positions'])[:5], c(cur['opacity']), c(cur['scales']), c(cur['rotations']), c(xyz), c(left), c(right),
                      centre, c(hand['positions']), c(hand['opacity']), c(hand['scales']), c(hand['rotations']))
