"""Shared helpers of the parity tests (tests only)."""
import ctypes

import numpy as np

from animatablegaussians_amd import camera


def cam_of(scene_or_cam):
    return camera.camera_from_intr_extr(scene_or_cam["extr"], scene_or_cam["intr"], scene_or_cam["img_w"], scene_or_cam["img_h"])


def oracle_forward(scene, cam, **kw):
    from oracle import raster_oracle as ro
    return ro.forward(scene["means3D"], scene["colors"], scene["opacities"], scene.get("scales"), scene.get("rotations"),
                      scene["bg"], cam["viewmatrix"], cam["projmatrix"], cam["tanfovx"], cam["tanfovy"],
                      cam["img_w"], cam["img_h"], cov3D_precomp=scene.get("cov3D_precomp"), **kw)


def oracle_backward(st, scene, cam, grads):
    from oracle import raster_oracle as ro
    return ro.backward(st, scene["means3D"], scene["colors"], scene.get("scales"), scene.get("rotations"), scene["bg"],
                       cam["viewmatrix"], cam["projmatrix"], cam["tanfovx"], cam["tanfovy"],
                       grads["dL_dcolor"], grads["dL_ddepth"], grads["dL_dalpha"],
                       cov3D_precomp=scene.get("cov3D_precomp"))


def gpu_settings(scene, cam, device="cuda", debug=False):
    import torch
    from animatablegaussians_amd.rasterizer import GaussianRasterizationSettings
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    return GaussianRasterizationSettings(
        image_height=cam["img_h"], image_width=cam["img_w"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
        bg=t(scene["bg"]), scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]),
        sh_degree=0, campos=t(cam["campos"]), prefiltered=False, debug=debug)


def gpu_inputs(scene, device="cuda", requires_grad=False):
    import torch
    out = {}
    for k in ("means3D", "colors", "opacities", "scales", "rotations", "cov3D_precomp"):
        if scene.get(k) is None:
            out[k] = None
            continue
        v = torch.from_numpy(np.ascontiguousarray(scene[k])).to(device)
        if requires_grad:
            v.requires_grad_(True)
        out[k] = v
    return out


def _scratch_view(buf, off, nbytes, dtype):
    base = buf.data_ptr()
    start = ((base + 255) & ~255) - base + off
    return buf[start:start + nbytes].cpu().numpy().view(dtype)


def gpu_native_forward(scene, cam, device="cuda"):
    """Call the `_C.rasterize_gaussians` equivalent and unpack the private scratch for comparison."""
    import torch
    from animatablegaussians_amd import _lib
    from animatablegaussians_amd.rasterizer import native_rasterize_gaussians
    rs = gpu_settings(scene, cam, device)
    inp = gpu_inputs(scene, device)
    empty = torch.Tensor([])
    e = lambda v: empty if v is None else v  # noqa: E731
    R, color, depth, alpha, radii, geom, binning, img = native_rasterize_gaussians(
        rs.bg, inp["means3D"], inp["colors"], inp["opacities"], e(inp["scales"]), e(inp["rotations"]), 1.0,
        e(inp["cov3D_precomp"]), rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width,
        empty, 0, rs.campos, False, True)
    P, W, H = scene["means3D"].shape[0], cam["img_w"], cam["img_h"]
    st = {"num_rendered": R, "color": color.cpu().numpy(), "depth": depth.cpu().numpy(), "alpha": alpha.cpu().numpy(),
          "radii": radii.cpu().numpy(),
          "_torch": dict(rs=rs, inp=inp, radii=radii, geom=geom, binning=binning, img=img, alpha=alpha)}
    if P == 0:
        return st
    L = _lib.lib()
    lay = _lib.AgRasterScratchLayout()
    _lib.check(L.ag_raster_describe_scratch(P, W, H, R, ctypes.byref(lay)), "describe")
    view = _scratch_view
    rec = view(geom, lay.geom_rec_off, P * lay.geom_rec_stride, np.float32).reshape(P, lay.geom_rec_stride // 4)
    vis = st["radii"] > 0
    z = lambda a: np.where(vis.reshape((-1,) + (1,) * (a.ndim - 1)), a, 0).astype(a.dtype)  # noqa: E731
    st["means2D"] = z(rec[:, 0:2].copy())
    st["conic_opacity"] = z(rec[:, 2:6].copy())
    st["depths"] = z(rec[:, 9].copy())
    st["r2cut"] = rec[:, 10].copy()
    st["cov3D"] = z(view(geom, lay.geom_cov3d_off, P * 24, np.float32).reshape(P, 6).copy())
    st["tiles_touched"] = view(geom, lay.geom_tiles_touched_off, P * 4, np.uint32).copy()
    T = ((W + 15) // 16) * ((H + 15) // 16)
    st["ranges"] = view(img, lay.img_ranges_off, T * 8, np.uint32).reshape(T, 2).copy()
    st["n_contrib"] = view(img, lay.img_n_contrib_off, W * H * 4, np.uint32).reshape(H, W).copy()
    st["tile_count"] = view(img, lay.img_tile_count_off, T * 4, np.uint32).copy()
    if R > 0:
        st["point_list"] = view(binning, lay.bin_point_list_off, R * 4, np.uint32).copy()
        st["keys"] = view(binning, lay.bin_keys_off, R * 8, np.uint64).copy()
    else:
        st["point_list"] = np.zeros(0, np.uint32)
        st["keys"] = np.zeros(0, np.uint64)
    return st


def gpu_native_backward(fw, grads, alphas=None):
    """`_C.rasterize_gaussians_backward` equivalent on the forward state `fw` (from gpu_native_forward).
    `alphas` overrides the saved forward alpha map (an explicit input of the reference's backward too).
    Returns the 8 API gradients plus the internal accumulators dL_dconic [P,4] and dL_ddepths [P,1]."""
    import torch
    from animatablegaussians_amd import _lib
    from animatablegaussians_amd.rasterizer import native_rasterize_gaussians_backward
    t = fw["_torch"]
    rs, inp = t["rs"], t["inp"]
    dev = inp["means3D"].device
    empty = torch.Tensor([])
    e = lambda v: empty if v is None else v  # noqa: E731
    P = inp["means3D"].shape[0]
    accum = torch.empty((_lib.lib().ag_raster_accum_bytes(P),), dtype=torch.uint8, device=dev)
    al = t["alpha"] if alphas is None else torch.from_numpy(np.ascontiguousarray(alphas)).to(dev)
    g = lambda k: torch.from_numpy(np.ascontiguousarray(grads[k])).to(dev)  # noqa: E731
    out = native_rasterize_gaussians_backward(
        rs.bg, inp["means3D"], t["radii"], inp["colors"], e(inp["scales"]), e(inp["rotations"]), 1.0,
        e(inp["cov3D_precomp"]), rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, g("dL_dcolor"), g("dL_ddepth"),
        g("dL_dalpha"), empty, 0, rs.campos, t["geom"], fw["num_rendered"], t["binning"], t["img"], al, True,
        _accum_buffer=accum)
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
    res = {n: o.cpu().numpy() for n, o in zip(names, out)}
    acc = _scratch_view(accum, 0, P * 64, np.float32).reshape(P, 16)
    # the accumulator row holds the MOMENTS of q = G dL/dalpha (csrc/ag_common.h AccumSlot); dL/dconic = -0.5 * opacity * (q dx^2, q dx dy,
    # q dy^2) is applied by the preprocess backward and restated here in float64 for the comparison with the oracle's dL_dconic
    nhop = -0.5 * inp["opacities"].detach().cpu().numpy().astype(np.float64).reshape(P)
    res["dL_dconic"] = np.stack([nhop * acc[:, 2], nhop * acc[:, 3], np.zeros(P), nhop * acc[:, 4]], 1)
    res["dL_ddepths"] = acc[:, 9:10].copy()
    return res


def assert_image_parity(gpu, ref, atol=1e-4, fragile_atol=6e-3, max_fragile_frac=5e-3):
    """colour/depth/alpha within atol (fp32 tolerance stated by BASELINE.json north_star: 1e-4) at every pixel the
    oracle did not flag as sitting on a discrete blend threshold; flagged pixels may flip one 1/255-sized term."""
    frag = ref["fragile"].astype(bool)
    assert frag.mean() <= max_fragile_frac, f"too many fragile pixels: {frag.mean()}"
    for k in ("color", "depth", "alpha"):
        d = np.abs(gpu[k] - ref[k])
        scale = np.maximum(1.0, np.abs(ref[k]))
        bad = (d > atol * scale) & ~frag[None]
        assert not bad.any(), f"{k}: {bad.sum()} non-fragile pixels differ, max {d[~np.broadcast_to(frag[None], d.shape)].max()}"
        assert (d[np.broadcast_to(frag[None], d.shape)] <= fragile_atol * scale[np.broadcast_to(frag[None], d.shape)]).all(), f"{k}: fragile pixel off by more than one threshold term"
    nc = gpu["n_contrib"] != ref["n_contrib"]
    assert not (nc & ~frag).any(), f"n_contrib differs at {int((nc & ~frag).sum())} non-fragile pixels"
    worst = max(float((np.abs(gpu[k] - ref[k]) / np.maximum(1.0, np.abs(ref[k])))[:, ~frag].max()) for k in ("color", "depth", "alpha"))
    print(f"\n[parity] image {ref['color'].shape[2]}x{ref['color'].shape[1]}: {frag.mean():.2e} of pixels fragile (cap {max_fragile_frac:g}), "
          f"worst non-fragile image difference {worst:.2e} (bar {atol:g})")


_SLOT_OF = {"dL_dmeans2D": (0, 1, None), "dL_dconic": (2, 3, None, 4), "dL_dopacity": (5,), "dL_dcolors": (6, 7, 8),
            "dL_ddepths": (9,)}


def assert_accum_parity(got, ref, rtol=1e-4, k_eps=64.0, ref_perturbed=None, k_sens=0.0, max_ratio=1.0):
    """Blend-backward accumulators vs the fp64-accumulated oracle.

    |got - ref| <= rtol*|ref| + k_eps*eps_fp32*sum|term| + 1e-7: the first term is the stated 1e-4 fp32 bar, the
    second is the spread between admissible float summation orders of the reference's atomicAdds (any order is
    "the reference"), with abs_sum measured by the oracle.  ``ref_perturbed`` (end-to-end comparisons only): the oracle's result
    with every exp() scaled by 1 + 2^-20 -- k_sens * |ref_perturbed - ref| is the reference algorithm's own movement under a
    rounding-sized change of its transcendental, added per element.  ``max_ratio``: the cap on the worst ratio to the limit (1.0 = the limit
    itself; a measured-and-capped comparison passes its cap here).  Returns the worst ratio to the limit."""
    eps = float(np.finfo(np.float32).eps)
    worst_all = 0.0
    for name, slots in _SLOT_OF.items():
        g = np.asarray(got[name], np.float64)
        r = np.asarray(ref[name], np.float64)
        assert np.isfinite(g).all(), f"{name}: non-finite"
        for col, slot in enumerate(slots):
            if slot is None:
                assert not g[:, col].any(), f"{name}[:, {col}] must stay zero"
                continue
            d = np.abs(g[:, col] - r[:, col])
            lim = rtol * np.abs(r[:, col]) + k_eps * eps * ref["abs_sum"][:, slot].astype(np.float64) + 1e-7
            if ref_perturbed is not None:
                lim = lim + k_sens * np.abs(np.asarray(ref_perturbed[name], np.float64)[:, col] - r[:, col])
            worst = float((d / lim).max()) if d.size else 0.0
            worst_all = max(worst_all, worst)
            assert worst <= max_ratio, (f"{name}[:, {col}]: {int((d > lim * max_ratio).sum())} of {d.size} over tolerance x {max_ratio:g}, worst ratio "
                                  f"{worst:.2f}, max |diff| {d.max():.3e}, ref max {np.abs(r[:, col]).max():.3e}")
    return worst_all


def assert_rows_close(got, ref, name, rtol=1e-4, row_rtol=1e-5):
    """Per-Gaussian outputs of the streaming backward: |got - ref| <= rtol*|ref| + row_rtol*max|ref row| + 1e-9.
    The row term covers cancellation inside one Gaussian's chain rule (fp32 op order / FMA differ)."""
    got = np.asarray(got, np.float64).reshape(ref.shape)
    ref = np.asarray(ref, np.float64)
    assert np.isfinite(got).all(), f"{name}: non-finite"
    d = np.abs(got - ref)
    lim = rtol * np.abs(ref) + row_rtol * np.abs(ref).max(axis=1, keepdims=True) + 1e-9
    worst = float((d / lim).max()) if d.size else 0.0
    assert worst <= 1.0, f"{name}: {int((d > lim).sum())} of {d.size} over tolerance, worst ratio {worst:.2f}, max |diff| {d.max():.3e}"


