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
