"""Drop-in wiring that can be checked without a GPU (and, for the in-place patch, only where the reference tree exists)."""
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "animatablegaussians_amd", "dropin")
REF = "/root/reference"


def test_dropin_modules_export_the_reference_names():
    sys.path.insert(0, DROPIN)
    try:
        import fused
        import upfirdn2d
        import diff_gaussian_rasterization_depth_alpha as d
        assert callable(fused.fused_bias_act) and callable(upfirdn2d.upfirdn2d)
        assert hasattr(d, "GaussianRasterizationSettings") and hasattr(d, "GaussianRasterizer")
        from diff_gaussian_rasterization_depth_alpha import _C
        assert callable(_C.rasterize_gaussians) and callable(_C.rasterize_gaussians_backward) and callable(_C.mark_visible)
    finally:
        sys.path.remove(DROPIN)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_patch_reference_routes_conv2d_gradfix_to_the_mfma_path():
    import torch
    sys.path.insert(0, DROPIN)
    sys.path.insert(0, REF)
    try:
        import patch_reference
        g = patch_reference.apply()
        from animatablegaussians_amd import conv as agc
        assert g.conv2d.__module__ == "patch_reference" and g.conv_transpose2d.__module__ == "patch_reference"
        # CPU tensors must fail loudly (there is no fallback), with the MFMA path's own message
        with pytest.raises(RuntimeError, match="GPU"):
            g.conv2d(torch.zeros(1, 4, 8, 8), torch.zeros(4, 4, 3, 3), padding=1, groups=1)
        del agc
    finally:
        sys.path.remove(DROPIN)
        sys.path.remove(REF)
        for m in [m for m in sys.modules if m == "network" or m.startswith("network.")]:
            del sys.modules[m]
