"""The loss tail of one training iteration (reference ``main_avatar.py:196-245``): boundary-mask
compositing, L1, mask loss, the square crop around the subject and the LPIPS term.  Everything here except LPIPS is a handful of
element-wise torch operations on two images; LPIPS runs on this package's kernels (``lpips.py``)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def composite(image, gt_image, boundary_mask_img, bg_color):
    """Replace the boundary band of both images by the background colour (main_avatar.py:199-202).  image, gt_image [3, H, W];
    boundary_mask_img [H, W] bool (True on the band); bg_color [3]."""
    keep = 1. - boundary_mask_img.to(torch.float32)
    fill = (1. - keep[None]) * bg_color[:, None, None]
    return image * keep[None] + fill, gt_image * keep[None] + fill


def mask_bbox(mask) -> tuple:
    """(min_v, min_u, max_v, max_u) of the set pixels of a HOST mask (numpy array / CPU tensor), as ``crop_image`` derives them from
    ``torch.argwhere`` (main_avatar.py:82-84).  The ground-truth mask is data: the loader has it on the host anyway, and a box
    computed there (``items['mask_bbox']``) spares the training step the device read-back the reference pays in every iteration."""
    import numpy as np
    m = np.asarray(mask) > 0
    vs, us = np.nonzero(m.any(1))[0], np.nonzero(m.any(0))[0]
    return int(vs[0]), int(us[0]), int(vs[-1]), int(us[-1])


def crop_image(gt_mask, patch_size, randomly, bg_color, *images, bbox=None):
    """Square crop around the mask's bounding box, padded with the background, then a random ``patch_size`` window or a
    bilinear resize to ``patch_size`` (main_avatar.py:75-115).  One host synchronisation (the bounding box) unless ``bbox`` =
    ``mask_bbox(host mask)`` is passed."""
    if bbox is None:
        mask_uv = torch.argwhere(gt_mask > 0.)
        min_v, min_u = (int(v) for v in mask_uv.min(0)[0])
        max_v, max_u = (int(v) for v in mask_uv.max(0)[0])
    else:
        min_v, min_u, max_v, max_u = bbox
    len_v, len_u = max_v - min_v, max_u - min_u
    max_size = max(len_v, len_u)
    rnd = randomly and max_size > patch_size
    if rnd:
        rv = int(torch.randint(0, max_size - patch_size + 1, (1,)))
        ru = int(torch.randint(0, max_size - patch_size + 1, (1,)))
    out = []
    for image in images:
        canvas = bg_color[:, None, None] * torch.ones((3, max_size, max_size), dtype=image.dtype, device=image.device)
        if len_v > len_u:
            s = (max_size - len_u) // 2
            canvas[:, :, s:s + len_u] = image[:, min_v:max_v, min_u:max_u]
        else:
            s = (max_size - len_v) // 2
            canvas[:, s:s + len_v, :] = image[:, min_v:max_v, min_u:max_u]
        if rnd:
            canvas = canvas[:, rv:rv + patch_size, ru:ru + patch_size]
        else:
            canvas = F.interpolate(canvas[None], size=(patch_size, patch_size), mode='bilinear')[0]
        out.append(canvas)
    return out if len(out) > 1 else out[0]


def lpips_loss(lpips, image, gt_image):
    """main_avatar.py:117-124: square [3, S, S] images in [0, 1], channels flipped to the order the metric was trained on."""
    assert image.shape[1] == image.shape[2] and gt_image.shape[1] == gt_image.shape[2]
    return lpips(image[None, [2, 1, 0]], gt_image[None, [2, 1, 0]], normalize=True).mean()


def training_loss(render_output, items, bg_color, loss_weight, lpips=None, patch_size=512, random_patch=False):
    """The scalar the reference back-propagates (main_avatar.py:196-245) from ``AvatarNet.render`` output and the dataset
    item (``color_img`` [H, W, 3], ``mask_img`` [H, W] bool, ``boundary_mask_img`` [H, W] bool; optional ``mask_bbox`` from
    ``mask_bbox`` on the host copy of the mask: no read-back in the step).  Returns (loss, parts)."""
    image = render_output['rgb_map'].permute(2, 0, 1)
    color = items['color_img'].clone()
    color[~items['mask_img']] = bg_color
    gt_image = color.permute(2, 0, 1)
    mask_img = items['mask_img'].to(torch.float32)
    image, gt_image = composite(image, gt_image, items['boundary_mask_img'], bg_color)
    total, parts = 0., {}
    if loss_weight.get('l1', 0.) > 0.:
        parts['l1_loss'] = torch.abs(image - gt_image).mean()
        total = total + loss_weight['l1'] * parts['l1_loss']
    if loss_weight.get('mask', 0.) and 'mask_map' in render_output:
        keep = 1. - items['boundary_mask_img'].to(torch.float32)
        parts['mask_loss'] = torch.abs(render_output['mask_map'].squeeze(-1) * keep - mask_img * keep).mean()
        total = total + loss_weight['mask'] * parts['mask_loss']
    if loss_weight.get('lpips', 0.) > 0. and lpips is not None:
        ci, cg = crop_image(mask_img, patch_size, random_patch, bg_color, image, gt_image, bbox=items.get('mask_bbox'))
        parts['lpips_loss'] = lpips_loss(lpips, ci, cg)
        total = total + loss_weight['lpips'] * parts['lpips_loss']
    parts['offset_loss'] = torch.linalg.norm(render_output['offset'], dim=-1).mean()
    total = total + loss_weight.get('offset', 0.) * parts['offset_loss']
    return total, parts
