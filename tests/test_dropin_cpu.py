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


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_the_references_render3_through_the_dropin_issues_the_native_call_of_our_rehost():
    """Boundary B1 / row a6, one level deeper than the import check (round 6).  The REFERENCE'S OWN `gaussians/gaussian_renderer.render3`
    (its file, unmodified, imported from /root/reference) is run with `diff_gaussian_rasterization_depth_alpha` resolving to our drop-in package,
    for a colour call and a spherical-harmonics call; so is our re-host `animatablegaussians_amd.gaussian_renderer.render3` on the same inputs.
    The native entry point both end in (`rasterizer.native_rasterize_gaussians`, the `_C.rasterize_gaussians` equivalent) is replaced by a recorder:
    both paths must hand it the same 19-argument tuple -- same order, same tensors to 1e-6 (view / projection matrices, camera centre, tan fov,
    colours incl. the SH -> RGB the reference does in Python), same flags -- and route the outputs back under the reference's keys.  No GPU here: the
    reference's two `.cuda()` calls and its `device="cuda"` are mapped to the CPU for the duration of the test.  (Running the reference's Python ON the
    GPU box is ruled out by the task: it cannot travel there.)"""
    import numpy as np
    import torch
    from animatablegaussians_amd import gaussian_renderer as ours
    from animatablegaussians_amd import rasterizer as rz
    sys.path.insert(0, DROPIN)
    sys.path.insert(0, REF)
    calls = []

    def recorder(*args, **kw):
        calls.append((args, kw))
        H, W, P = int(args[12]), int(args[13]), args[1].shape[0]
        z = lambda *s: torch.zeros(*s)  # noqa: E731
        return (0, 0), z(3, H, W), z(1, H, W), z(1, H, W), torch.zeros(P, dtype=torch.int32), z(1), z(1), z(1), z(1)[:0]

    real_native, real_cuda, real_zl = rz.native_rasterize_gaussians, torch.Tensor.cuda, torch.zeros_like

    def zeros_like_cpu(t, **kw):
        kw.pop("device", None)
        return real_zl(t, **kw)

    try:
        rz.native_rasterize_gaussians = lambda *a, **k: recorder(*a, **k)[:8]
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.zeros_like = zeros_like_cpu
        ref_mod = __import__("gaussians.gaussian_renderer", fromlist=["render3"])
        g = torch.Generator().manual_seed(5)
        P = 37
        base = {"positions": torch.randn(P, 3, generator=g) + torch.tensor([0., 0., 3.]), "opacity": torch.rand(P, 1, generator=g),
                "scales": torch.rand(P, 3, generator=g) * 0.05, "rotations": torch.nn.functional.normalize(torch.randn(P, 4, generator=g)),
                "max_sh_degree": 0}
        extr = torch.eye(4)
        extr[:3, :3] = torch.tensor([[0.8, 0., 0.6], [0., 1., 0.], [-0.6, 0., 0.8]])
        extr[:3, 3] = torch.tensor([0.1, -0.2, 2.5])
        intr = torch.tensor([[1100., 0., 250.], [0., 1090., 260.], [0., 0., 1.]])
        bg = torch.tensor([0.1, 0.2, 0.3])
        for variant in ("colors", "shs"):
            vals = dict(base)
            if variant == "colors":
                vals["colors"] = torch.rand(P, 3, generator=g)
            else:
                vals["shs"] = torch.randn(P, 3, 4, generator=g) * 0.3
                vals["max_sh_degree"] = 1
            del calls[:]
            out_ref = ref_mod.render3(vals, bg, extr, intr, 512, 480, 1.0)
            out_our = ours.render3(vals, bg, extr, intr, 512, 480, 1.0)
            assert len(calls) == 2
            (a_ref, k_ref), (a_our, k_our) = calls
            assert len(a_ref) == len(a_our) == 19 and k_ref == k_our
            for i, (x, y) in enumerate(zip(a_ref, a_our)):
                if torch.is_tensor(x):
                    assert torch.is_tensor(y) and x.shape == y.shape and x.dtype == y.dtype, (variant, i)
                    if x.numel():
                        assert float((x.double() - y.double()).abs().max()) <= 1e-6 * max(1.0, float(x.double().abs().max())), (variant, i)
                else:
                    assert (abs(x - y) <= 1e-7 * max(1.0, abs(x))) if isinstance(x, float) else x == y, (variant, i, x, y)
            assert set(out_ref) == set(out_our) == {"render", "depth", "mask", "viewspace_points", "visibility_filter", "radii"}
            assert out_ref["render"].shape == (3, 480, 512) and out_ref["visibility_filter"].dtype == torch.bool
    finally:
        rz.native_rasterize_gaussians, torch.Tensor.cuda, torch.zeros_like = real_native, real_cuda, real_zl
        sys.path.remove(DROPIN)
        sys.path.remove(REF)
        for m in [m for m in sys.modules if m in ("gaussians", "utils") or m.startswith(("gaussians.", "utils."))]:
            del sys.modules[m]
