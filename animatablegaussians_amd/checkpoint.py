"""Checkpoints in the reference's layout -- SURVEY.md §8(f)-3; ``main_avatar.py:777-813``.

``<dir>/net.pt`` = ``{'epoch_idx', 'iter_idx', 'avatar_net': state_dict}`` with the REFERENCE's ``AvatarNet.state_dict()`` keys
(``color_net.*``, ``position_net.*``, ``other_net.*``, ``viewdir_net.*``), so a reference run can resume from a file written here
(``load_state_dict(..., strict=False)``: the reference's constant blur / Haar buffers are not stored, it recomputes them in its
constructors) and ``load_ckpt`` reads the reference's own ``net.pt`` (``PRETRAINED_MODEL.md``) through ``load_reference_state_dict``.
``<dir>/optm.pt`` = ``{'avatar_net': optimizer.state_dict()}``: Adam moments are indexed by parameter ORDER, which differs between
the two module trees, so optimizer files are only exchanged between runs of the same implementation."""
from __future__ import annotations

import os

import torch


def avatar_state_dict(net) -> dict:
    sd = {}
    for prefix in ("color_net", "position_net", "other_net"):
        for k, v in getattr(net, prefix).reference_state_dict().items():
            sd[f"{prefix}.{k}"] = v.detach()
    if net.with_viewdirs:
        for i in (0, 2):
            for k in ("weight", "bias"):
                sd[f"viewdir_net.{i}.{k}"] = getattr(net, f"viewdir_net__{i}__{k}").detach()
    return sd


def save_ckpt(path: str, net, optm=None, epoch_idx: int = 0, iter_idx: int = 0, save_optm: bool = True) -> None:
    os.makedirs(path, exist_ok=True)
    torch.save({'epoch_idx': epoch_idx, 'iter_idx': iter_idx,
                'avatar_net': {k: v.cpu() for k, v in avatar_state_dict(net).items()}}, os.path.join(path, 'net.pt'))
    if save_optm and optm is not None:
        torch.save({'avatar_net': optm.state_dict()}, os.path.join(path, 'optm.pt'))


def load_ckpt(path: str, net, optm=None, load_optm: bool = True):
    """-> (epoch_idx, iter_idx), as the reference's ``load_ckpt``."""
    net_dict = torch.load(os.path.join(path, 'net.pt'), map_location='cpu')
    if 'avatar_net' in net_dict:
        net.load_reference_state_dict(net_dict['avatar_net'])
    else:
        print('[WARNING] Cannot find "avatar_net" from the network checkpoint!')
    if load_optm and optm is not None and os.path.exists(os.path.join(path, 'optm.pt')):
        optm_dict = torch.load(os.path.join(path, 'optm.pt'), map_location='cpu')
        if 'avatar_net' in optm_dict:
            optm.load_state_dict(optm_dict['avatar_net'])
        else:
            print('[WARNING] Cannot find "avatar_net" from the optimizer checkpoint!')
    return net_dict['epoch_idx'], net_dict['iter_idx']
