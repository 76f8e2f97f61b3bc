"""``AvatarNet`` under the constructor the reference trainer uses (``main_avatar.py:45-48``:
``importlib.import_module(opt['model']['module']).AvatarNet(opt['model'])``), backed by
``animatablegaussians_amd.avatar.AvatarNet``.  Select it from the YAML config::

    model:
      module: avatar_module          # with animatablegaussians_amd/dropin on PYTHONPATH

The per-subject assets are read exactly where the reference reads them (``network/avatar.py:27,31,43``), with the
OpenCV-free EXR reader (``animatablegaussians_amd/exr.py``); the only reference dependency left is its global ``config``
module (data directory and device), imported at call time."""
from animatablegaussians_amd.avatar import AvatarNet as _AvatarNet


class AvatarNet(_AvatarNet):
    def __new__(cls, opt):
        import config               # the reference's global config module
        net = _AvatarNet.from_data_dir(opt, config.opt['train']['data']['data_dir'], device=config.device)
        net.__class__ = cls
        return net

    def __init__(self, opt):         # built in __new__ (main_avatar.py:48 then calls .to(config.device): a no-op)
        pass

    def _fix_hand_enabled(self):     # network/avatar.py:183
        import config
        return bool(config.opt['test'].get('fix_hand', False)) and config.opt['mode'] == 'test'

    def _fix_hand_pose_map(self):    # network/avatar.py:61-67 (called without arguments at main_avatar.py:584)
        import glob

        import config
        import numpy as np
        import torch

        from animatablegaussians_amd import exr
        paths = sorted(glob.glob(config.opt['train']['data']['data_dir'] + '/smpl_pos_map/%08d.exr' % config.opt['test']['fix_hand_id']))
        m = exr.imread(paths[0])
        half = m.shape[1] // 2
        m = np.concatenate([m[:, :half], m[:, half:]], 2).transpose((2, 0, 1))
        return torch.from_numpy(np.ascontiguousarray(m)).to(torch.float32).to(config.device)[:3]
