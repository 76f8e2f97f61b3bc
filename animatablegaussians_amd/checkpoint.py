"""Checkpoints in the reference's layout -- SURVEY.md §8(f)-3; ``main_avatar.py:777-813``.

``<dir>/net.pt`` = ``{'epoch_idx', 'iter_idx', 'avatar_net': avatar_net.state_dict()}``.  ``AvatarNet.state_dict()`` here has the
REFERENCE's keys in the reference's order (``color_net.*``, ``position_net.*``, ``other_net.*``, ``viewdir_net.*``, the constant
blur / Haar buffers included; pinned by ``tests/golden/state_layout.json``, produced by the reference module), so the reference's
strict ``load_state_dict`` takes a file written here and ``load_ckpt`` takes the reference's own ``net.pt`` (``PRETRAINED_MODEL.md``).
``<dir>/optm.pt`` = ``{'avatar_net': optimizer.state_dict()}``: Adam moments are indexed by parameter ORDER, which is the
reference's too (same fixture), so optimizer files are interchangeable as well."""
from __future__ import annotations

import os

import torch


def avatar_state_dict(net) -> dict:
    return {k: v.detach() for k, v in net.state_dict().items()}


def save_ckpt(path: str, net, optm=None, epoch_idx: int = 0, iter_idx: int = 0, save_optm: bool = True) -> None:
    os.makedirs(path, exist_ok=True)
    torch.save({'epoch_idx': epoch_idx, 'iter_idx': iter_idx,
                'avatar_net': {k: v.cpu() for k, v in avatar_state_dict(net).items()}}, os.path.join(path, 'net.pt'))
    if save_optm and optm is not None:
        torch.save({'avatar_net': optm.state_dict()}, os.path.join(path, 'optm.pt'))


def load_ckpt(path: str, net, optm=None, load_optm: bool = True):
    """-> (epoch_idx, iter_idx), as the reference's ``load_ckpt``."""
    net_dict = torch.load(os.path.join(path, 'net.pt'), map_location='cpu')
    legacy = False
    if 'avatar_net' in net_dict:
        sd = net_dict['avatar_net']
        own = net.state_dict()
        const = ('.kernel', '.ll', '.lh', '.hl', '.hh')
        legacy = any(k.endswith(const) and k not in sd for k in own)
        if legacy and hasattr(net, 'load_reference_state_dict'):
            # written by this package before round 2 (no constant blur / Haar buffers, other parameter order): the constants are the same
            # here, so the learnable tensors load by name; its optm.pt is indexed by the OLD parameter order and is refused below
            print('[WARNING] net.pt without the constant FIR / Haar buffers (pre-round-2 layout): loading by name; optm.pt is skipped')
            net.load_reference_state_dict(sd)
        else:
            net.load_state_dict(sd)
    else:
        print('[WARNING] Cannot find "avatar_net" from the network checkpoint!')
    if load_optm and not legacy and optm is not None and os.path.exists(os.path.join(path, 'optm.pt')):
        optm_dict = torch.load(os.path.join(path, 'optm.pt'), map_location='cpu')
        if 'avatar_net' in optm_dict:
            optm.load_state_dict(optm_dict['avatar_net'])
        else:
            print('[WARNING] Cannot find "avatar_net" from the optimizer checkpoint!')
    return net_dict['epoch_idx'], net_dict['iter_idx']
