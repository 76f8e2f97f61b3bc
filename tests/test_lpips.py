"""LPIPS(net='vgg') -- SURVEY.md §8(f)-1 -- against the reference's own class.

Fixture tests/golden/lpips_vgg_64.npz: the reference's LPIPS run on CPU (stand-in torchvision VGG16 `features` stack, name-seeded
parameters from lpips.lpips_named_fill) on two 64x64 image pairs: value, per-level values, gradient w.r.t. the first image.
CPU: the torch oracle restatement reproduces it.  GPU: the MI355X module (MFMA convolutions, fused bias+ReLU, max-pool and
per-level distance kernels) reproduces it within fp32 summation-order noise (1e-4 relative on values, 2e-3 of the max on the
image gradient -- thirteen convolutions and ReLU / max-pool selections deep), plus unit tests of the two new kernels."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "lpips_vgg_64.npz")


def _state(module_sd):
    from animatablegaussians_amd.lpips import lpips_named_fill
    return lpips_named_fill({k: v for k, v in module_sd.items() if not k.startswith("scaling_layer")})


def test_oracle_reproduces_the_reference_class():
    import torch
    from animatablegaussians_amd.lpips import LPIPS
    from oracle import lpips_oracle as lo
    gold = np.load(GOLD)
    sd = _state(LPIPS().reference_state_dict())
    for tag, normalize in (("n", True), ("r", False)):
        a = torch.from_numpy(gold[f"{tag}_in0"]).requires_grad_(True)
        val, res = lo.lpips(a, torch.from_numpy(gold[f"{tag}_in1"]), sd, normalize=normalize)
        val.sum().backward()
        np.testing.assert_allclose(val.detach().numpy(), gold[f"{tag}_val"], rtol=1e-5)
        np.testing.assert_allclose([float(r.detach()) for r in res[1:]], gold[f"{tag}_res"][1:], rtol=1e-5)
        np.testing.assert_allclose(a.grad.numpy(), gold[f"{tag}_grad"], rtol=1e-4, atol=1e-6 * np.abs(gold[f"{tag}_grad"]).max())


def test_state_dict_surface_and_freezing():
    import torch
    from animatablegaussians_amd.lpips import LPIPS
    m = LPIPS(net='vgg')
    keys = set(m.reference_state_dict())
    assert {"net.slice1.0.weight", "net.slice3.14.bias", "net.slice5.28.weight", "lin4.model.1.weight", "scaling_layer.shift"} <= keys
    assert len([k for k in keys if k.startswith("net.")]) == 26 and not m.training
    assert all(not p.requires_grad for p in m.parameters())
    m.load_reference_state_dict(m.reference_state_dict())
    with pytest.raises(RuntimeError):
        m.load_reference_state_dict({**m.reference_state_dict(), "bogus": torch.zeros(1)})
    with pytest.raises(RuntimeError):
        LPIPS(net='alex')


@pytest.mark.gpu
def test_gpu_module_matches_the_reference_class():
    import torch
    from animatablegaussians_amd.lpips import LPIPS
    gold = np.load(GOLD)
    m = LPIPS(net='vgg')
    m.load_reference_state_dict({**_state(m.reference_state_dict()), "scaling_layer.shift": m.scaling_layer__shift,
                                 "scaling_layer.scale": m.scaling_layer__scale})
    m = m.cuda()
    for tag, normalize in (("n", True), ("r", False)):
        a = torch.from_numpy(gold[f"{tag}_in0"]).cuda().requires_grad_(True)
        val, res = m(a, torch.from_numpy(gold[f"{tag}_in1"]).cuda(), retPerLayer=True, normalize=normalize)
        val.sum().backward()
        np.testing.assert_allclose(val.detach().cpu().numpy(), gold[f"{tag}_val"], rtol=1e-4)
        np.testing.assert_allclose([float(r.detach()) for r in res], gold[f"{tag}_res"], rtol=2e-4)      # slot 0 = total, as the reference
        gmax = np.abs(gold[f"{tag}_grad"]).max()
        assert np.abs(a.grad.cpu().numpy() - gold[f"{tag}_grad"]).max() <= 2e-3 * gmax


@pytest.mark.gpu
def test_maxpool_and_level_kernels_vs_torch():
    import torch
    import torch.nn.functional as F
    from animatablegaussians_amd.lpips import _LpipsLevel, maxpool2x2
    from oracle import lpips_oracle as lo
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 5, 11, 14, generator=g)
    x[0, 0, 0, 0] = x[0, 0, 0, 1] = 7.0                      # a tie: the first maximum wins
    xc = x.clone().requires_grad_(True)
    yc = F.max_pool2d(xc, 2, 2)
    up = torch.randn(yc.shape, generator=g)
    (yc * up).sum().backward()
    xg = x.cuda().requires_grad_(True)
    yg = maxpool2x2(xg)
    (yg * up.cuda()).sum().backward()
    assert torch.equal(yg.cpu(), yc.detach()) and torch.equal(xg.grad.cpu(), xc.grad)

    f0, f1 = torch.randn(1, 37, 9, 13, generator=g), torch.randn(1, 37, 9, 13, generator=g)
    f0[0, :, 0, 0] = 0.0                                       # a zero feature vector: the eps terms matter
    lin = torch.rand(37, generator=g)
    f0c = f0.clone().requires_grad_(True)
    ref = (F.conv2d((lo.normalize_tensor(f0c) - lo.normalize_tensor(f1)) ** 2, lin.view(1, -1, 1, 1))).mean()
    (3.0 * ref).backward()
    f0g = f0.cuda().requires_grad_(True)
    got = _LpipsLevel.apply(f0g, f1.cuda(), lin.cuda())
    (3.0 * got).sum().backward()
    np.testing.assert_allclose(got.item(), ref.item(), rtol=1e-5)
    gref = f0c.grad.numpy()
    np.testing.assert_allclose(f0g.grad.cpu().numpy(), gref, rtol=1e-4, atol=1e-6 * np.abs(gref).max())


@pytest.mark.gpu
def test_training_loss_tail_runs_and_differentiates():
    """composite + L1 + mask + crop + LPIPS + offset (main_avatar.py:196-245) on synthetic images; gradient reaches the render."""
    import torch
    from animatablegaussians_amd import losses
    from animatablegaussians_amd.lpips import LPIPS
    g = torch.Generator().manual_seed(8)
    H = W = 160
    rgb = torch.rand(H, W, 3, generator=g).cuda().requires_grad_(True)
    mask_map = torch.rand(H, W, 1, generator=g).cuda().requires_grad_(True)
    offset = (torch.randn(100, 3, generator=g) * 0.01).cuda().requires_grad_(True)
    mask = torch.zeros(H, W, dtype=torch.bool)
    mask[30:130, 50:110] = True
    boundary = torch.zeros(H, W, dtype=torch.bool)
    boundary[28:32, 48:112] = True
    items = {'color_img': torch.rand(H, W, 3, generator=g).cuda(), 'mask_img': mask.cuda(), 'boundary_mask_img': boundary.cuda()}
    bg = torch.tensor([1.0, 1.0, 1.0]).cuda()
    lp = LPIPS(net='vgg').cuda()
    total, parts = losses.training_loss({'rgb_map': rgb, 'mask_map': mask_map, 'offset': offset}, items, bg,
                                        {'l1': 1.0, 'mask': 0.1, 'lpips': 0.1, 'offset': 0.005}, lpips=lp, patch_size=64)
    assert set(parts) == {'l1_loss', 'mask_loss', 'lpips_loss', 'offset_loss'} and torch.isfinite(total)
    total.backward()
    for t in (rgb, mask_map, offset):
        assert t.grad is not None and torch.isfinite(t.grad).all() and float(t.grad.abs().max()) > 0
    assert not rgb.grad[boundary.cuda()].any()              # the boundary band is replaced by the background on both images
    # the same step with the bounding box computed on the host copy of the mask (no device read-back): identical loss
    assert losses.mask_bbox(mask.numpy()) == (30, 50, 129, 109)
    total2, _ = losses.training_loss({'rgb_map': rgb, 'mask_map': mask_map, 'offset': offset}, {**items, 'mask_bbox': losses.mask_bbox(mask)},
                                     bg, {'l1': 1.0, 'mask': 0.1, 'lpips': 0.1, 'offset': 0.005}, lpips=lp, patch_size=64)
    assert torch.equal(total2, total)


def test_crop_with_host_bbox_equals_crop_with_device_readback():
    import torch
    from animatablegaussians_amd import losses
    g = torch.Generator().manual_seed(1)
    mask = torch.zeros(90, 120)
    mask[10:71, 33:60] = 1.0
    mask[40, 80] = 1.0
    img, gt = torch.rand(3, 90, 120, generator=g), torch.rand(3, 90, 120, generator=g)
    bg = torch.tensor([0.2, 0.4, 0.6])
    a = losses.crop_image(mask, 32, False, bg, img, gt)
    b = losses.crop_image(mask, 32, False, bg, img, gt, bbox=losses.mask_bbox(mask.numpy()))
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and a[0].shape == (3, 32, 32)
