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


def test_flipped_fir_kernel_cache_never_serves_a_recycled_address():
    """styleunet_ops._flipped caches the flipped taps of an upfirdn2d kernel for its backward.  A key made of the address alone would be
    handed to a DIFFERENT kernel once the first is freed and the allocator re-uses its block (round-2 advisor finding: 5 of 50 freshly
    allocated 4 x 4 kernels got a stale result on the CPU allocator).  The entry now holds the source tensor and is used only for that
    very object."""
    import torch
    from animatablegaussians_amd.styleunet_ops import _flipped
    stale = 0
    for i in range(200):
        k = torch.full((4, 4), float(i)) + torch.arange(16.).reshape(4, 4)
        f = _flipped(k)
        stale += int(not torch.equal(f, torch.flip(k, [0, 1])))
        del k, f
    assert stale == 0
    k = torch.arange(16.).reshape(4, 4)
    assert _flipped(k) is _flipped(k)                                # still a cache for a kernel that stays alive
    k.mul_(2.0)                                                      # in-place change bumps the version: recomputed
    assert torch.equal(_flipped(k), torch.flip(k, [0, 1]))


def test_camera_identity_cache_follows_the_tensors_values():
    """gaussian_renderer._camera_tensors: the same extr / intr TENSORS again -> no device->host copy (the cached matrices, same objects); an in-place
    change of the values, or other tensors, -> the matrices of the new values; a dead tensor whose address is handed out again cannot alias."""
    import gc

    import numpy as np
    import torch

    from animatablegaussians_amd import gaussian_renderer as gr

    extr = torch.eye(4)
    extr[:3, 3] = torch.tensor([0.1, -0.2, 2.5])
    intr = torch.tensor([[1100., 0., 512.], [0., 1100., 512.], [0., 0., 1.]])
    a = gr._camera_tensors(extr, intr, 1024, 1024, "cpu")
    calls = []
    real = gr._camera_tensors_by_value
    gr._camera_tensors_by_value = lambda *args: (calls.append(1), real(*args))[1]
    try:
        assert gr._camera_tensors(extr, intr, 1024, 1024, "cpu") is a and not calls            # identity hit: the by-value path (the copy) is not taken
        assert gr._camera_tensors(extr, intr, 512, 512, "cpu") is not a and len(calls) == 1     # another image size: another camera
        extr[2, 3] = 3.0                                                                        # in place: the version counter moves
        b = gr._camera_tensors(extr, intr, 1024, 1024, "cpu")
        assert len(calls) == 2 and not torch.equal(b["viewmatrix"], a["viewmatrix"])
        c = gr._camera_tensors(extr.clone(), intr.clone(), 1024, 1024, "cpu")                   # other tensors, same values: the by-value cache
        assert len(calls) == 3 and c is b
        # a stale identity entry (its tensors are gone) never hits, whatever address and version a new tensor has
        e2, i2 = extr.clone(), intr.clone()
        gr._camera_tensors(e2, i2, 1024, 1024, "cpu")
        key = [k for k, v in gr._camera_ident.items() if v[0]() is e2][0]
        stale = gr._camera_ident[key]
        del e2, i2
        gc.collect()
        assert stale[0]() is None and stale[1]() is None
        e3 = torch.eye(4)
        gr._camera_ident[(id(e3), e3.data_ptr(), e3._version, id(intr), intr.data_ptr(), intr._version, 1024, 1024, "cpu")] = stale
        n = len(calls)
        d = gr._camera_tensors(e3, intr, 1024, 1024, "cpu")
        assert len(calls) == n + 1 and np.allclose(d["viewmatrix"].numpy()[3, :3], 0.0)         # identity extrinsics: no translation row
        # tensors created under torch.inference_mode() have no version counter (reading ._version raises): they take the by-value path
        with torch.inference_mode():
            ei, ii = extr.clone(), intr.clone()
            n = len(calls)
            f = gr._camera_tensors(ei, ii, 1024, 1024, "cpu")
            assert len(calls) == n + 1 and f is b
    finally:
        gr._camera_tensors_by_value = real
