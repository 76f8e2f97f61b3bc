"""``AvatarNet`` under the constructor the reference trainer uses (``main_avatar.py:45-48``:
``importlib.import_module(opt['model']['module']).AvatarNet(opt['model'])``), backed by
``animatablegaussians_amd.avatar.AvatarNet``.  Select it from the YAML config::

    model:
      module: avatar_module          # with animatablegaussians_amd/dropin on PYTHONPATH

The per-subject assets are read exactly where the reference reads them (``network/avatar.py:27,31,43``) -- this is the
one place that needs OpenCV's EXR reader and the reference's ``config`` module, both imported at call time.  NOT
exercised by this repository's tests (no OpenCV, no dataset in the build image); everything behind it is."""
import numpy as np
import torch

from animatablegaussians_amd.avatar import AvatarNet as _AvatarNet


class AvatarNet(_AvatarNet):
    def __init__(self, opt):
        import cv2 as cv            # noqa: F401  (reference dependency; EXR support must be enabled as the reference does)
        import config               # the reference's global config module
        data_dir = config.opt['train']['data']['data_dir']
        cano = cv.imread(data_dir + '/smpl_pos_map/cano_smpl_pos_map.exr', cv.IMREAD_UNCHANGED)
        lbs = np.load(data_dir + '/smpl_pos_map/init_pts_lbs.npy')
        nml = None
        if opt.get('with_viewdirs', True):
            nml = torch.from_numpy(cv.imread(data_dir + '/smpl_pos_map/cano_smpl_nml_map.exr', cv.IMREAD_UNCHANGED))
        super().__init__(opt, cano_smpl_map=torch.from_numpy(cano), lbs=torch.from_numpy(lbs).float(), cano_nml_map=nml,
                         device=config.device)

    def to(self, *args, **kwargs):   # main_avatar.py:48 calls .to(config.device); buffers are already there
        return super().to(*args, **kwargs)
