"""GPU parity tests of the HIP rasterizer against the CPU oracle, through the C ABI (pytest -m gpu).

Bars (BASELINE.json north_star): bit-exact tile/sort indices (radii, per-Gaussian tile counts, projected means, depth
keys, conics, cov3D, sorted point_list, tile ranges); rendered colour/depth/alpha and every gradient within 1e-4
(fp32; relative to magnitude for gradients, which are unordered float sums).
"""
import ctypes

import os

import numpy as np
import pytest

import helpers as h
from animatablegaussians_amd import synth

pytestmark = pytest.mark.gpu


def _bitexact(gpu, ref):
    for k in ("radii", "tiles_touched"):
        assert np.array_equal(gpu[k], ref[k]), f"{k} not bit-exact: {(gpu[k] != ref[k]).sum()} differ"
    vis = ref["radii"] > 0   # per-Gaussian state is only defined (and only consumed) for rasterized Gaussians
    for k in ("means2D", "depths", "conic_opacity", "cov3D"):
        a, b = gpu[k][vis].view(np.uint32), ref[k][vis].view(np.uint32)
        assert np.array_equal(a, b), f"{k} not bit-exact: {(a != b).sum()} words differ, max abs {np.abs(gpu[k] - ref[k]).max()}"
    assert gpu["num_rendered"] == ref["num_rendered"]
    assert np.array_equal(gpu["ranges"], ref["ranges"]), "tile ranges differ"
    assert np.array_equal(gpu["point_list"], ref["point_list"]), "sorted point_list differs"


@pytest.mark.parametrize("P,img", [(10000, 512), (3000, 500)])
def test_forward_config1_bitexact_and_images(P, img):
    scene = synth.random_gaussians(P=P, img=img)
    if img == 500:   # ragged: W, H not multiples of the 16x16 tile
        scene["img_w"], scene["img_h"] = 500, 300
    cam = h.cam_of(scene)
    ref = h.oracle_forward(scene, cam)
    gpu = h.gpu_native_forward(scene, cam)
    _bitexact(gpu, ref)
    h.assert_image_parity(gpu, ref)


@pytest.mark.parametrize("name", ["raster_random_p1500_128", "raster_ragged_p800_100x60_cov3d"])
def test_against_reference_golden_fixtures(name):
    """HIP path vs the fixtures produced by the reference's own code (tests/golden/make_golden.py)."""
    from test_oracle_cpu import load_golden
    scene, cam, st_ref, g_ref = load_golden(name)
    gpu = h.gpu_native_forward(scene, cam)
    ref = dict(st_ref, fragile=h.oracle_forward(scene, cam)["fragile"])
    if scene.get("cov3D_precomp") is not None:
        ref["cov3D"] = np.where((st_ref["radii"] > 0)[:, None], scene["cov3D_precomp"], 0).astype(np.float32)
    _bitexact(gpu, ref)
    h.assert_image_parity(gpu, ref)
    # gradients on the golden's saved alpha map: accumulators vs the golden (fp32 sums of the same terms in the
    # reference's sequential order) with the summation-order slack measured by the oracle
    from oracle import raster_oracle as ro
    keep = (~ref["fragile"].astype(bool)).astype(np.float32)[None]
    grads = {k: np.ascontiguousarray(scene[k] * keep) for k in ("dL_dcolor", "dL_ddepth", "dL_dalpha")}
    acc = ro.backward_blend(st_ref, scene["colors"], scene["bg"], grads["dL_dcolor"], grads["dL_ddepth"], grads["dL_dalpha"])
    got = h.gpu_native_backward(gpu, grads, alphas=st_ref["alpha"])
    h.assert_accum_parity(got, acc)
    if not ref["fragile"].any():   # no masked pixel: the golden gradients themselves are directly comparable
        for k in ("dL_dmeans3D", "dL_dcov3D") + (("dL_dscales", "dL_drotations") if scene.get("scales") is not None else ()):
            h.assert_rows_close(got[k], g_ref[k], k, rtol=1e-3, row_rtol=1e-4)


def test_forward_cov3d_precomp_path():
    scene = synth.random_gaussians(P=2000)
    cam = h.cam_of(scene)
    ref0 = h.oracle_forward(scene, cam)
    scene2 = dict(scene, cov3D_precomp=ref0["cov3D"].copy(), scales=None, rotations=None)
    ref = h.oracle_forward(scene2, cam)
    gpu = h.gpu_native_forward(scene2, cam)
    _bitexact(gpu, ref)
    h.assert_image_parity(gpu, ref)


def test_forward_empty_and_all_culled():
    import torch
    scene = synth.random_gaussians(P=64)
    cam = h.cam_of(scene)
    # P == 0: the reference skips the native call -> all-zero colour (NOT bg), rasterize_points.cu:68-83
    empty = {k: (v[:0] if isinstance(v, np.ndarray) and v.shape[:1] == (64,) else v) for k, v in scene.items()}
    gpu = h.gpu_native_forward(empty, cam)
    assert gpu["num_rendered"] == 0 and not gpu["color"].any() and not gpu["alpha"].any()
    # everything behind the near plane: num_rendered 0, blend still runs -> colour == bg, alpha == depth == 0
    behind = dict(scene, means3D=scene["means3D"] + np.array([0, 0, 10.0], np.float32))
    ref = h.oracle_forward(behind, cam)
    gpu = h.gpu_native_forward(behind, cam)
    assert ref["num_rendered"] == 0 and gpu["num_rendered"] == 0
    assert not gpu["radii"].any() and not gpu["alpha"].any() and not gpu["depth"].any()
    np.testing.assert_array_equal(gpu["color"], np.broadcast_to(scene["bg"][:, None, None], gpu["color"].shape))
    assert not gpu["ranges"].any() and not gpu["n_contrib"].any()
    torch.cuda.synchronize()


def test_forward_oversized_tile_uses_global_sort():
    """> 4096 instances in one tile: exercises the global-memory bitonic fallback and heavy depth ties."""
    rs = np.random.RandomState(7)
    P = 9000
    scene = synth.random_gaussians(P=P)
    scene["means3D"] = (rs.normal(0, 0.004, (P, 3)) + np.array([0.0, 0.0, 0.0])).astype(np.float32)
    scene["means3D"][: P // 2, 2] = 0.0          # exact depth ties -> order must fall back to the Gaussian index
    scene["scales"] = np.full((P, 3), 0.002, np.float32)
    scene["opacities"] = np.full((P, 1), 0.02, np.float32)
    cam = h.cam_of(scene)
    ref = h.oracle_forward(scene, cam)
    assert (ref["ranges"][:, 1] - ref["ranges"][:, 0]).max() > 4096
    gpu = h.gpu_native_forward(scene, cam)
    _bitexact(gpu, ref)
    h.assert_image_parity(gpu, ref, max_fragile_frac=0.05)


def test_tile_sort_every_size_class_is_bit_exact():
    """One scene whose tile lists span every path of the tile sort -- shorter than one thread's 4 keys, inside one wave's 256-key
    network, several waves' runs merged through LDS (257 .. 2048), the chunked large class (2049 .. 8192) and the global network
    beyond -- with a block of exact depth ties (order falls back to the Gaussian index).  Sorted lists and ranges bit-exact against the oracle."""
    rs = np.random.RandomState(11)
    P, n_halo = 50000, 400
    scene = synth.random_gaussians(P=P, img=256)
    # a dense core (tiles beyond 8192 entries), a wider cloud around it (hundreds to thousands) and a sparse halo (a handful per tile)
    n1 = P // 3
    n2 = P - n1 - n_halo
    halo = np.stack([rs.uniform(-0.5, 0.5, n_halo), rs.uniform(-0.5, 0.5, n_halo), rs.uniform(-0.1, 0.1, n_halo)], 1)
    scene["means3D"] = np.concatenate([rs.normal(0, 0.025, (n1, 3)), rs.normal(0, 0.12, (n2, 3)), halo]).astype(np.float32)
    scene["means3D"][: P // 8, 2] = 0.0          # exact depth ties inside the long tiles
    scene["scales"] = np.full((P, 3), 0.0015, np.float32)
    scene["opacities"] = np.full((P, 1), 0.05, np.float32)
    cam = h.cam_of(scene)
    ref = h.oracle_forward(scene, cam)
    n = (ref["ranges"][:, 1] - ref["ranges"][:, 0]).astype(np.int64)
    classes = [(1, 4), (5, 256), (257, 512), (513, 1024), (1025, 2048), (2049, 8192), (8193, 10 ** 9)]
    have = [int(((n >= lo) & (n <= hi)).sum()) for lo, hi in classes]
    assert all(have), f"tile-length classes {classes} -> counts {have}: the scene must reach every path"
    print(f"\n[sort] tiles per length class {dict(zip([c[1] for c in classes], have))}, longest {int(n.max())}")
    gpu = h.gpu_native_forward(scene, cam)
    _bitexact(gpu, ref)


def _run_autograd(scene, cam, grads):
    import torch
    from animatablegaussians_amd.rasterizer import GaussianRasterizer
    rs = h.gpu_settings(scene, cam)
    inp = h.gpu_inputs(scene, requires_grad=True)
    means2D = torch.zeros_like(inp["means3D"], requires_grad=True)
    color, radii, depth, alpha = GaussianRasterizer(rs)(
        means3D=inp["means3D"], means2D=means2D, opacities=inp["opacities"], shs=None, colors_precomp=inp["colors"],
        scales=inp["scales"], rotations=inp["rotations"], cov3D_precomp=inp["cov3D_precomp"])
    t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    loss = (color * t(grads["dL_dcolor"])).sum() + (depth * t(grads["dL_ddepth"])).sum() + (alpha * t(grads["dL_dalpha"])).sum()
    loss.backward()
    torch.cuda.synchronize()
    g = {"dL_dmeans2D": means2D.grad, "dL_dmeans3D": inp["means3D"].grad, "dL_dcolors": inp["colors"].grad,
         "dL_dopacity": inp["opacities"].grad}
    if inp["scales"] is not None:
        g["dL_dscales"] = inp["scales"].grad
        g["dL_drotations"] = inp["rotations"].grad
    if inp["cov3D_precomp"] is not None:
        g["dL_dcov3D"] = inp["cov3D_precomp"].grad
    return {k: v.cpu().numpy() for k, v in g.items()}


def _masked_grads(scene, ref):
    """Upstream gradients with fragile pixels zeroed: every gradient is linear in them, so a pixel whose discrete
    blend decision may legitimately flip is removed from both sides of the comparison."""
    keep = (~ref["fragile"].astype(bool)).astype(np.float32)[None]
    return {k: np.ascontiguousarray(scene[k] * keep) for k in ("dL_dcolor", "dL_ddepth", "dL_dalpha")}


def _check_backward(scene, cam, ill_expected, exempt_cap=1.25):
    """Three-level backward parity:
    (1) blend-backward accumulators on the SAME saved forward state (the oracle's alpha map is passed as the
        `alphas` input, exactly as the reference's backward takes it) vs the fp64-accumulated oracle;
    (2) the streaming preprocess-backward on the GPU's own accumulators vs the oracle's restatement of it;
    (3) end to end through torch.autograd: identical to the native call on the GPU's own state (plumbing), and a coarse
        sanity bound against oracle forward -> oracle backward."""
    from oracle import raster_oracle as ro
    ref = h.oracle_forward(scene, cam)
    grads = _masked_grads(scene, ref)
    fw = h.gpu_native_forward(scene, cam)
    assert np.array_equal(fw["point_list"], ref["point_list"]) and np.array_equal(fw["ranges"], ref["ranges"])
    nc = fw["n_contrib"] != ref["n_contrib"]
    assert not (nc & ~ref["fragile"].astype(bool)).any()

    # (1)
    acc_ref = ro.backward_blend(ref, scene["colors"], scene["bg"], grads["dL_dcolor"], grads["dL_ddepth"], grads["dL_dalpha"])
    got = h.gpu_native_backward(fw, grads, alphas=ref["alpha"])
    h.assert_accum_parity(got, acc_ref)
    for k in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity"):
        assert got[k].shape == acc_ref[k].shape

    # (2)
    pre = ro.backward_preprocess(ref, got, scene["means3D"], scene.get("scales"), scene.get("rotations"),
                                 cam["viewmatrix"], cam["projmatrix"], cam["tanfovx"], cam["tanfovy"],
                                 cov3D_precomp=scene.get("cov3D_precomp"))
    h.assert_rows_close(got["dL_dmeans3D"], pre["dL_dmeans3D"], "dL_dmeans3D")
    h.assert_rows_close(got["dL_dcov3D"], pre["dL_dcov3D"], "dL_dcov3D")
    if scene.get("scales") is not None:
        h.assert_rows_close(got["dL_dscales"], pre["dL_dscales"], "dL_dscales")
        # quaternion chain rule: sums of +-2 q_i s_j dL/dM_jk products that largely cancel, so the slack is relative to
        # the row's largest entry, three times looser than for the other rows
        h.assert_rows_close(got["dL_drotations"], pre["dL_drotations"], "dL_drotations", row_rtol=3e-5)
    vis = ref["radii"] > 0
    for k in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"):
        assert not got[k][~vis].any(), f"{k} must be zero for culled Gaussians"

    # (3) autograd plumbing: same kernels on the GPU's own saved state; only the atomic order may differ
    own = h.gpu_native_backward(fw, grads)
    e2e = _run_autograd(scene, cam, grads)
    for k, v in e2e.items():   # routing check: a mis-wired gradient is off by O(1), atomic-order noise by ~1e-6
        np.testing.assert_allclose(v.reshape(own[k].shape), own[k], rtol=1e-3, atol=1e-4 * max(1.0, np.abs(own[k]).max()),
                                   err_msg="autograd " + k)
    # (4) TRUE end to end -- GPU forward -> GPU blend backward on its OWN saved state against oracle forward -> oracle blend backward
    # -- at the stated bar, with no outlier budget.  The reference's backward restarts from T_final = 1 - alpha_out and divides by
    # it, so a forward rounding difference d_alpha of alpha_out (itself far inside the 1e-4 image bar) enters EVERY backward term of
    # that pixel as d_alpha / T_final -- for the reference against itself (atomics in another order, FMA contraction, another exp)
    # as much as for us (DESIGN.md "backward conditioning").  A pixel is ill-conditioned when T_final < 1e-2 (a 1e-6 forward
    # difference, ~10 ulp over a long list, already costs 1e-4), or when the MEASURED forward difference moves its terms by more
    # than 2e-5 relative.  With the upstream gradients of those (and of the fragile pixels) zeroed on both sides, every accumulator
    # element must meet 1e-4*|ref| + the float summation-order slack + 16x the reference algorithm's own movement under a
    # rounding-sized change of exp() (oracle re-run with every exp() scaled by 1 + 2^-20: per-element condition estimate).
    # Together with (2) -- the streaming stage is a verified function of the accumulators -- this bounds every API gradient.  The
    # masked fractions are printed and capped so they cannot grow silently.
    frag = ref["fragile"].astype(bool)
    ro.set_exp_scale(1.0 + 2.0 ** -20)
    try:
        ref_p = h.oracle_forward(scene, cam)
    finally:
        ro.set_exp_scale(1.0)
    flips = ref_p["fragile"].astype(bool) | (ref_p["n_contrib"] != ref["n_contrib"])
    t_fin = 1.0 - ref["alpha"][0].astype(np.float64)
    sat = t_fin < 1e-2
    amp = np.abs(fw["alpha"][0].astype(np.float64) - ref["alpha"][0]) > 2e-5 * t_fin
    ill = sat | amp
    # cap = 1.5 x the fraction measured for this scene (`ill_expected`, printed below by `pytest -s`) + 5e-4: the exemption cannot grow silently
    assert ill.mean() <= 1.5 * ill_expected + 5e-4 and (flips & ~frag).mean() < 5e-3, \
        f"{ill.mean():.4f} of pixels ill-conditioned (measured when the cap was set: {ill_expected}), {flips.mean():.2e} flip"
    # Evidence for the exemption, printed per scene: the REFERENCE'S OWN gradients (oracle against oracle with every exp() scaled by
    # 1 + 2^-20, i.e. one rounding of alpha) on the exempted pixels and on the kept ones.  Measured: 268 k avatar @1024^2 -- exempted
    # pixels: the reference moves by 7.7e-4 / 4.6e-3 / 4.5e-2 of an element's value (50th / 90th / 99th percentile), kept pixels
    # 1.9e-6 / 1.3e-5 / 1.1e-4; 10 k random scene -- 1.7e-5 / 1.2e-4 / 1.2e-3 against 2.2e-6 / 1.6e-5 / 1.5e-4.  The 1e-4 bar is not
    # defined on the exempted set: the reference does not meet it against itself there.
    evidence = {}
    for label, m in (("exempted", ill & ~(frag | flips)), ("kept", ~(ill | frag | flips))):
        km = m.astype(np.float32)[None]
        gm = {k: np.ascontiguousarray(scene[k] * km) for k in ("dL_dcolor", "dL_ddepth", "dL_dalpha")}
        a0 = ro.backward_blend(ref, scene["colors"], scene["bg"], gm["dL_dcolor"], gm["dL_ddepth"], gm["dL_dalpha"])
        ro.set_exp_scale(1.0 + 2.0 ** -20)
        try:
            a1 = ro.backward_blend(ref_p, scene["colors"], scene["bg"], gm["dL_dcolor"], gm["dL_ddepth"], gm["dL_dalpha"])
        finally:
            ro.set_exp_scale(1.0)
        rel = []
        for name, slots in h._SLOT_OF.items():
            for col, slot in enumerate(slots):
                if slot is not None:
                    r0, r1 = np.asarray(a0[name], np.float64)[:, col], np.asarray(a1[name], np.float64)[:, col]
                    big = np.abs(r0) > 1e-6 * max(np.abs(r0).max(), 1e-300)
                    rel.append(np.abs(r1 - r0)[big] / np.abs(r0)[big])
        rel = np.concatenate(rel) if rel else np.zeros(1)
        evidence[label] = np.percentile(rel, [50, 90, 99]) if rel.size else np.zeros(3)
    keep = (~(frag | flips | ill)).astype(np.float32)[None]
    wc = {k: np.ascontiguousarray(scene[k] * keep) for k in ("dL_dcolor", "dL_ddepth", "dL_dalpha")}
    acc_wc = ro.backward_blend(ref, scene["colors"], scene["bg"], wc["dL_dcolor"], wc["dL_ddepth"], wc["dL_dalpha"])
    ro.set_exp_scale(1.0 + 2.0 ** -20)
    try:
        acc_p = ro.backward_blend(ref_p, scene["colors"], scene["bg"], wc["dL_dcolor"], wc["dL_ddepth"], wc["dL_dalpha"])
    finally:
        ro.set_exp_scale(1.0)
    worst_wc = h.assert_accum_parity(h.gpu_native_backward(fw, wc), acc_wc, k_eps=512.0, ref_perturbed=acc_p, k_sens=16.0)
    # ... and the ill-conditioned rest, through autograd, against the oracle's backward evaluated on the alpha map the GPU forward
    # saved (which removes the forward-rounding amplification but not the repeated division by (1 - alpha) near saturation): loose.
    gref = h.oracle_backward(dict(ref, alpha=fw["alpha"]), scene, cam, grads)
    loose = 0.0
    for k, v in e2e.items():
        d = np.abs(v.astype(np.float64).reshape(gref[k].shape) - gref[k])
        lim = 1e-3 * np.abs(gref[k]) + 1e-4 * np.abs(gref[k]).max(axis=1, keepdims=True) + 1e-6
        loose = max(loose, float((d > lim).mean()))
        assert (d > lim).mean() < 1e-3, f"end-to-end {k}: {(d > lim).mean():.2e} of elements beyond the coarse bound"
    # (5) round 6: the EXEMPTED pixels at the strict bar where both sides share the alpha map.  Upstream gradients restricted to the ill-conditioned
    # (not fragile, not flipping) pixels; the GPU blend backward on its own saved state against the oracle's blend backward evaluated on the alpha
    # map the GPU forward saved -- the forward-rounding amplification d_alpha / T_final is then the same on both sides, what is left is the backward's
    # own arithmetic near saturation (the reference divides T by (1 - alpha) entry after entry, we carry the product and take one reciprocal).
    # EVERY accumulator element, no outlier budget: 1e-4 |ref| + 64 eps sum|term| (the summation-order slack of level 1), worst ratio printed and
    # capped at `exempt_cap` x that limit.  Measured (profiles/r06_parity_measure_raster.txt): 0.94 on the avatar (10.9 % of the image exempted), 0.91 on
    # the 10 k random scene, 0.22 / 0.03 on the small ones -- inside the level-1 limit itself; the cap of 1.25 leaves room for the order of the atomics.
    exempt_cap = float(os.environ.get("AG_TEST_EXEMPT_CAP", exempt_cap))       # (measuring runs)
    exempt = (ill & ~(frag | flips)).astype(np.float32)[None]
    ge = {k: np.ascontiguousarray(scene[k] * exempt) for k in ("dL_dcolor", "dL_ddepth", "dL_dalpha")}
    acc_e = ro.backward_blend(dict(ref, alpha=fw["alpha"]), scene["colors"], scene["bg"], ge["dL_dcolor"], ge["dL_ddepth"], ge["dL_dalpha"])
    worst_ex = h.assert_accum_parity(h.gpu_native_backward(fw, ge), acc_e, max_ratio=exempt_cap)
    print(f"\n[parity] exempted pixels ({exempt.mean():.3f} of the image) on the shared alpha map: worst ratio to 1e-4 |ref| + 64 eps sum|term| = {worst_ex:.2f} (cap {exempt_cap:g})")
    print(f"\n[parity] P={ref['radii'].shape[0]} {cam['img_w']}x{cam['img_h']}: fragile pixels {frag.mean():.2e} (+ {(flips & ~frag).mean():.2e} "
          f"that flip under the exp probe), ill-conditioned {ill.mean():.2e} (T_final < 1e-2: {sat.mean():.2e}, forward difference "
          f"amplified past 2e-5: {(amp & ~sat).mean():.2e}); well-conditioned set: every accumulator element within the 1e-4 bound "
          f"(worst ratio {worst_wc:.2f}); all pixels: {loose:.2e} of gradient elements beyond 1e-3*|ref| + 1e-4*rowmax\n"
          f"         the reference against itself under exp * (1 + 2^-20), relative movement of its accumulator elements p50/p90/p99: "
          + "; ".join(f"{k} pixels {v[0]:.1e}/{v[1]:.1e}/{v[2]:.1e}" for k, v in evidence.items()))


@pytest.mark.parametrize("P,img", [(10000, 512), (3000, 500)])
def test_backward_config1(P, img):
    scene = synth.random_gaussians(P=P, img=img)
    if img == 500:
        scene["img_w"], scene["img_h"] = 500, 300
        scene.update(synth.upstream_grads(500, 300, 5))
    _check_backward(scene, h.cam_of(scene), ill_expected=0.0292 if img == 512 else 0.0003)


def test_backward_cov3d_precomp():
    scene = synth.random_gaussians(P=2000)
    cam = h.cam_of(scene)
    ref0 = h.oracle_forward(scene, cam)
    scene2 = dict(scene, cov3D_precomp=ref0["cov3D"].copy(), scales=None, rotations=None)
    _check_backward(scene2, cam, ill_expected=0.0)


def test_avatar_config2_forward_backward():
    """BASELINE.json configs[1]: ~268 k Gaussians on the front|back map, one free-view camera at 1024^2."""
    av = synth.avatar_map_gaussians()
    camd = synth.free_view_cameras()[1]
    scene = dict(av, **camd)
    scene.update(synth.upstream_grads(1024, 1024, 11))
    cam = h.cam_of(scene)
    ref = h.oracle_forward(scene, cam)
    gpu = h.gpu_native_forward(scene, cam)
    _bitexact(gpu, ref)
    h.assert_image_parity(gpu, ref)
    _check_backward(scene, cam, ill_expected=0.1101)


def test_mark_visible():
    import torch
    from animatablegaussians_amd.rasterizer import GaussianRasterizer
    from oracle import raster_oracle as ro
    scene = synth.random_gaussians(P=5000)
    scene["means3D"][:, 2] += np.random.RandomState(3).uniform(0, 4.6, 5000).astype(np.float32)
    cam = h.cam_of(scene)
    want = ro.mark_visible(scene["means3D"], cam["viewmatrix"], cam["projmatrix"])
    got = GaussianRasterizer(h.gpu_settings(scene, cam)).markVisible(torch.from_numpy(scene["means3D"]).cuda())
    assert 0 < want.sum() < 5000
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_argument_contract_errors():
    import torch
    from animatablegaussians_amd.rasterizer import GaussianRasterizer
    scene = synth.random_gaussians(P=16)
    cam = h.cam_of(scene)
    r = GaussianRasterizer(h.gpu_settings(scene, cam))
    inp = h.gpu_inputs(scene)
    m2 = torch.zeros_like(inp["means3D"])
    with pytest.raises(Exception, match="excatly one of either SHs"):
        r(inp["means3D"], m2, inp["opacities"], shs=None, colors_precomp=None, scales=inp["scales"], rotations=inp["rotations"])
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(inp["means3D"], m2, inp["opacities"], colors_precomp=inp["colors"], scales=inp["scales"], rotations=None)
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        r(inp["means3D"][:, :2], m2, inp["opacities"], colors_precomp=inp["colors"], scales=inp["scales"], rotations=inp["rotations"])


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_colour_path_matches_reference_golden(deg):
    """shs= path of GaussianRasterizer (forward.cu:20-71, backward.cu:20-139) against the fixture the reference's own code
    produced (tests/golden/make_golden_sh.py): per-Gaussian colours are bit-identical (same fp32 operation order, no FMA
    contraction in the preprocess kernels), images within 1e-4, gradients within the blend-backward tolerances."""
    import os
    import torch
    from animatablegaussians_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "raster_sh_p600_96x80.npz"))
    dev = "cuda:0"
    t = lambda k: torch.from_numpy(np.ascontiguousarray(gold[k])).to(dev)  # noqa: E731
    W, H = (int(v) for v in gold["in_img_wh"])
    tanx, tany = (float(v) for v in gold["cam_tanfov"])
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=tanx, tanfovy=tany, bg=t("in_bg"), scale_modifier=1.0,
                                       viewmatrix=t("cam_viewmatrix"), projmatrix=t("cam_projmatrix"), sh_degree=deg,
                                       campos=t("cam_campos"), prefiltered=False, debug=False)
    leaves = {k: t("in_" + k).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    m2 = torch.zeros_like(leaves["means3D"], requires_grad=True)
    color, radii, depth, alpha = GaussianRasterizer(rs)(leaves["means3D"], m2, leaves["opacities"], shs=leaves["shs"],
                                                        scales=leaves["scales"], rotations=leaves["rotations"])
    assert np.array_equal(radii.cpu().numpy(), gold[f"d{deg}_st_radii"])
    for got, key in ((color, "color"), (depth, "depth"), (alpha, "alpha")):
        assert np.abs(got.detach().cpu().numpy() - gold[f"d{deg}_st_{key}"]).max() <= 1e-4, key
    # backward on the reference's alpha (the reference's backward restarts from 1 - alpha_out)
    torch.autograd.backward([color, depth, alpha], [t("in_dL_dcolor"), t("in_dL_ddepth"), t("in_dL_dalpha")])
    ref_sh = gold[f"d{deg}_g_dL_dsh"]
    got_sh = leaves["shs"].grad.cpu().numpy()
    n = (deg + 1) ** 2
    assert not got_sh[:, n:, :].any() and not got_sh[gold[f"d{deg}_st_radii"] <= 0].any()
    h.assert_rows_close(got_sh.reshape(len(got_sh), -1), ref_sh.reshape(len(ref_sh), -1), "dL_dsh", rtol=2e-3, row_rtol=2e-4)
    h.assert_rows_close(leaves["means3D"].grad.cpu().numpy(), gold[f"d{deg}_g_dL_dmeans3D"], "dL_dmeans3D", rtol=2e-3, row_rtol=2e-4)
    # clamped channels get exactly zero gradient: rows whose three channels are all clamped have dL_dsh == 0
    all_clamped = gold[f"d{deg}_st_clamped"].all(axis=1) & (gold[f"d{deg}_st_radii"] > 0)
    assert all_clamped.any() and not got_sh[all_clamped].any()


def test_sh_argument_errors():
    import torch
    from animatablegaussians_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from animatablegaussians_amd._lib import AgNativeError
    dev = "cuda:0"
    z = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731
    rs = GaussianRasterizationSettings(image_height=32, image_width=32, tanfovx=0.5, tanfovy=0.5, bg=z(3), scale_modifier=1.0,
                                       viewmatrix=torch.eye(4, device=dev), projmatrix=torch.eye(4, device=dev), sh_degree=2,
                                       campos=z(3), prefiltered=False, debug=False)
    m = torch.rand(10, 3, device=dev) + 1.0
    with pytest.raises(AgNativeError):      # degree 2 needs 9 coefficients, 4 given
        GaussianRasterizer(rs)(m, torch.zeros_like(m), torch.rand(10, 1, device=dev), shs=z(10, 4, 3),
                               scales=torch.rand(10, 3, device=dev), rotations=torch.rand(10, 4, device=dev))


def test_full_size_1m_gaussians_2048_properties():
    """BASELINE configs[4] scale (1024^2 front|back maps = ~1.07 M Gaussians, one 2048^2 view): far too large for the CPU oracle
    inside a test, so the size-independent invariants of the path are checked instead:
      * ranges tile the instance list exactly, every tile segment is sorted by (depth bits, Gaussian index) -- the unique order
        the reference's stable radix sort produces -- and holds exactly the Gaussians whose rect covers the tile;
      * n_contrib never exceeds the tile's list length; alpha in [0, 1], images finite; empty tiles carry the background;
      * the forward is bit-reproducible; the backward is finite, zero for culled Gaussians, and linear in the upstream gradient."""
    import torch
    S, W = 2048, 2048
    av = synth.avatar_map_gaussians(S)
    camd = synth.free_view_cameras(8, img=W, focal=2200.0)[1]
    scene = dict(av, **camd)
    cam = h.cam_of(scene)
    P = av["means3D"].shape[0]
    assert P > 1_000_000
    fw = h.gpu_native_forward(scene, cam)
    R = fw["num_rendered"]
    rng, pl = fw["ranges"].astype(np.int64), fw["point_list"]
    # the reference's sort key inside a tile: raw bits of the view-space depth (positive floats: monotone as integers), ties by index
    keys = (fw["depths"].view(np.uint32).astype(np.uint64)[pl] << np.uint64(32)) | pl.astype(np.uint64)
    ln = rng[:, 1] - rng[:, 0]
    assert R == int(fw["tiles_touched"].astype(np.int64).sum()) == int(ln.sum()) and (ln >= 0).all()
    nz = np.nonzero(ln)[0]
    order = np.argsort(rng[nz, 0])
    b, e_ = rng[nz, 0][order], rng[nz, 1][order]
    assert b[0] == 0 and e_[-1] == R and np.array_equal(e_[:-1], b[1:])         # the non-empty segments tile [0, R)
    # sortedness inside every segment: keys strictly ascending except across segment starts
    d = np.diff(keys.astype(np.uint64).view(np.int64))            # depth bits of positive floats << 32 | index: monotone as int64
    starts = np.zeros(R, bool)
    starts[rng[nz, 0]] = True
    assert (d[~starts[1:]] > 0).all()
    # each Gaussian appears tiles_touched times
    assert np.array_equal(np.bincount(pl, minlength=P).astype(np.uint32), fw["tiles_touched"])
    # per-pixel invariants
    T_x = (W + 15) // 16
    tile_len = ln.reshape(-1, T_x)
    per_pix_len = np.repeat(np.repeat(tile_len, 16, 0), 16, 1)[:W, :W]
    assert (fw["n_contrib"] <= per_pix_len).all()
    assert np.isfinite(fw["color"]).all() and np.isfinite(fw["depth"]).all()
    assert fw["alpha"].min() >= 0.0 and fw["alpha"].max() <= 1.0 + 1e-5
    empty_pix = per_pix_len == 0
    assert empty_pix.any() and np.array_equal(fw["color"][:, empty_pix], np.broadcast_to(av["bg"][:, None], (3, int(empty_pix.sum()))))
    assert not fw["alpha"][0][empty_pix].any()
    # reproducibility of the forward
    fw2 = h.gpu_native_forward(scene, cam)
    for k in ("color", "depth", "alpha", "n_contrib", "point_list", "radii"):
        assert np.array_equal(fw[k], fw2[k]), k
    # backward: finite, zero where culled, linear in the upstream gradient (float atomics: to rounding)
    up = synth.upstream_grads(W, W, 5)
    g1 = h.gpu_native_backward(fw, up)
    g2 = h.gpu_native_backward(fw, {k: 2.0 * v for k, v in up.items()})
    culled = fw["radii"] <= 0
    for k in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dcolors"):
        assert np.isfinite(g1[k]).all(), k
        assert not g1[k][culled].any(), k
        scale = np.abs(g1[k]).max()
        assert np.abs(g2[k] - 2.0 * g1[k]).max() <= 2e-4 * scale, k
    del torch


def test_full_size_1m_gaussians_2048_vs_oracle():
    """BASELINE configs[4] scale against the oracle itself (not only invariants): 1.07 M Gaussians, one 2048^2 view, f = 2200.  The C
    oracle's blend loops run OpenMP-parallel over tiles, so the comparison is seconds on the GPU box's host cores: per-Gaussian
    state, sorted instance list and tile ranges bit-exact, images within 1e-4 outside oracle-flagged fragile pixels, blend-backward
    accumulators on the same saved state within the accumulator tolerance, streaming backward within the row tolerance."""
    import time
    from oracle import raster_oracle as ro
    S, W = 2048, 2048
    av = synth.avatar_map_gaussians(S)
    scene = dict(av, **synth.free_view_cameras(8, img=W, focal=2200.0)[3])
    scene.update(synth.upstream_grads(W, W, 17))
    cam = h.cam_of(scene)
    t0 = time.perf_counter()
    ref = h.oracle_forward(scene, cam)
    t_ref = time.perf_counter() - t0
    assert ref["radii"].shape[0] > 1_000_000 and ref["num_rendered"] > 2_000_000
    gpu = h.gpu_native_forward(scene, cam)
    _bitexact(gpu, ref)
    h.assert_image_parity(gpu, ref)
    frag = ref["fragile"].astype(bool)
    print(f"\nconfigs[4] vs oracle: P={ref['radii'].shape[0]} R={ref['num_rendered']} oracle forward {t_ref:.1f} s, "
          f"fragile pixels {frag.mean():.2e}, max colour diff outside them "
          f"{np.abs(gpu['color'] - ref['color'])[:, ~frag].max():.2e}")
    grads = _masked_grads(scene, ref)
    acc_ref = ro.backward_blend(ref, scene["colors"], scene["bg"], grads["dL_dcolor"], grads["dL_ddepth"], grads["dL_dalpha"])
    got = h.gpu_native_backward(gpu, grads, alphas=ref["alpha"])
    # Tile lists up to ~8000 entries: twice the summation-order slack of the small scenes.  And, at 10.7 M accumulator elements, the
    # far tail of the one error source the slack does not model: T is recovered by a chain of thousands of fp32 divisions, the
    # reference (one division per entry) and any other grouping of that chain (16 entries per step in the region kernel, 4 in the wave
    # kernel -- fewer roundings than the reference itself) drift apart like a random walk, and an element whose few terms all sit at
    # the end of such a chain inherits that drift relative to its own size (measured with both kernels: identical error percentiles,
    # one element of 10.7 M at 1.2x the bound).  The oracle prices it per element: its own movement when every exp() is scaled by
    # 1 + 2^-20 (one rounding of alpha), times 4.
    ro.set_exp_scale(1.0 + 2.0 ** -20)
    try:
        ref_p = h.oracle_forward(scene, cam)
        acc_p = ro.backward_blend(ref_p, scene["colors"], scene["bg"], grads["dL_dcolor"], grads["dL_ddepth"], grads["dL_dalpha"])
    finally:
        ro.set_exp_scale(1.0)
    h.assert_accum_parity(got, acc_ref, k_eps=128.0, ref_perturbed=acc_p, k_sens=4.0)
    pre = ro.backward_preprocess(ref, got, scene["means3D"], scene["scales"], scene["rotations"], cam["viewmatrix"], cam["projmatrix"],
                                 cam["tanfovx"], cam["tanfovy"])
    for k, rr in (("dL_dmeans3D", 1e-5), ("dL_dcov3D", 1e-5), ("dL_dscales", 1e-5), ("dL_drotations", 3e-5)):
        h.assert_rows_close(got[k], pre[k], k, row_rtol=rr)


def test_fused_forward_backward_step_equals_the_operator_path():
    """rasterizer.FusedRasterStep (ag_raster_forward_backward: one native call per view, internal streams, on-device sum over the
    views of a step) against GaussianRasterizer + torch.autograd on the same views: images bit-identical, gradients equal up to the
    float-atomic order of the blend backward; a capacity that is too small is outgrown without disturbing the sums."""
    import torch
    from animatablegaussians_amd import rasterizer as rz
    from animatablegaussians_amd.rasterizer import FusedRasterStep, GaussianRasterizer
    sc = synth.random_gaussians(6000, seed=9, img=256, focal=275.0)
    cams = synth.free_view_cameras(3, img=256, focal=275.0)
    ups = [synth.upstream_grads(256, 256, 20 + v) for v in range(3)]
    inp = h.gpu_inputs(sc, requires_grad=True)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    settings, want_img = [], []
    for cam_d, up in zip(cams, ups):
        cam = h.cam_of(dict(sc, **cam_d))
        rs = h.gpu_settings(sc, cam)
        settings.append(rs)
        m2d = torch.zeros_like(inp["means3D"], requires_grad=True)
        color, radii, depth, alpha = GaussianRasterizer(rs)(means3D=inp["means3D"], means2D=m2d, opacities=inp["opacities"],
                                                            colors_precomp=inp["colors"], scales=inp["scales"], rotations=inp["rotations"])
        torch.autograd.backward([color, depth, alpha], [t(up["dL_dcolor"]), t(up["dL_ddepth"]), t(up["dL_dalpha"])])   # leaves accumulate
        want_img.append([x.detach().clone() for x in (color, depth, alpha, radii)])
    want = {"dL_dmeans3D": inp["means3D"].grad, "dL_dcolors": inp["colors"].grad, "dL_dopacity": inp["opacities"].grad,
            "dL_dscales": inp["scales"].grad, "dL_drotations": inp["rotations"].grad}
    det = {k: v.detach() for k, v in inp.items() if v is not None}
    key = (6000, 256, 256, torch.cuda.current_device())
    for cap0 in (None, 500):                       # second round: planned capacity far too small -> every view is redone once
        if cap0 is None:
            rz._capacity.pop(key, None)
        else:
            rz._capacity[key] = cap0
        step = FusedRasterStep(6000, 256, 256, "cuda", n_streams=2)
        for v, (rs, up) in enumerate(zip(settings, ups)):
            color, depth, alpha, radii, _ = step.view(rs, det["means3D"], det["colors"], det["opacities"], det["scales"], det["rotations"],
                                                      t(up["dL_dcolor"]), t(up["dL_ddepth"]), t(up["dL_dalpha"]), accumulate=True)
            torch.cuda.synchronize()
            for a, b, nm in zip((color, depth, alpha, radii), want_img[v], ("color", "depth", "alpha", "radii")):
                assert torch.equal(a, b), f"view {v}: {nm} differs from the operator path"
        got = step.join()
        torch.cuda.synchronize()
        for k, w in want.items():
            scale = float(w.abs().max())
            assert float((got[k] - w.reshape(got[k].shape)).abs().max()) <= 2e-5 * scale + 1e-9, k
        assert rz._capacity[key] > 500
    # without accumulate every view overwrites its slot: the last view on a slot is what it holds
    step = FusedRasterStep(6000, 256, 256, "cuda", n_streams=1)
    for rs, up in zip(settings, ups):
        g = step.view(rs, det["means3D"], det["colors"], det["opacities"], det["scales"], det["rotations"], t(up["dL_dcolor"]),
                      t(up["dL_ddepth"]), t(up["dL_dalpha"]))[4]
    last = step.join()
    one = FusedRasterStep(6000, 256, 256, "cuda", n_streams=1)
    one.view(settings[-1], det["means3D"], det["colors"], det["opacities"], det["scales"], det["rotations"], t(ups[-1]["dL_dcolor"]),
             t(ups[-1]["dL_ddepth"]), t(ups[-1]["dL_dalpha"]))
    ref = one.join()
    assert float((last["dL_dmeans3D"] - ref["dL_dmeans3D"]).abs().max()) <= 2e-5 * float(ref["dL_dmeans3D"].abs().max())
    del g
    # prepared handles (one per camera and slot, reused over iterations): same images and sums as view(); a second iteration with NEW
    # input arrays and new image gradients through the same handles picks the new pointers up
    step = FusedRasterStep(6000, 256, 256, "cuda", n_streams=2)
    ins = [det[k] for k in ("means3D", "colors", "opacities", "scales", "rotations")]
    gs = [[t(up["dL_dcolor"]), t(up["dL_ddepth"]), t(up["dL_dalpha"])] for up in ups]
    handles = [step.prepare(rs, *gs[v], v % 2) for v, rs in enumerate(settings)]
    for it in range(2):
        if it == 1:
            ins = [x.clone() for x in ins]
            gs = [[x.clone() for x in g3] for g3 in gs]
        for v, hd in enumerate(handles):
            color, depth, alpha, radii, _ = step.run(hd, *ins, image_grads=gs[v] if it == 1 else None, accumulate=True,
                                                     inputs_outlive_join=(it == 1))
            torch.cuda.synchronize()
            for a, b, nm in zip((color, depth, alpha, radii), want_img[v], ("color", "depth", "alpha", "radii")):
                assert torch.equal(a, b), f"prepared view {v}, iteration {it}: {nm} differs from the operator path"
        got = step.join()
        torch.cuda.synchronize()
        for k, w in want.items():
            scale = float(w.abs().max())
            assert float((got[k] - w.reshape(got[k].shape)).abs().max()) <= 2e-5 * scale + 1e-9, (k, it)
    with pytest.raises(RuntimeError):
        step.run(handles[0], ins[0][:10], *ins[1:])
    # run() does not wait for a view's instance count: a binning buffer that turns out too small is noticed when the slot is used next
    # (or at join()) and the view is redone then -- images and sums are right after join()
    rz._capacity[key] = 500
    step = FusedRasterStep(6000, 256, 256, "cuda", n_streams=2)
    handles = [step.prepare(rs, *gs[v], v % 2) for v, rs in enumerate(settings)]
    outs = [step.run(hd, *ins, accumulate=True) for hd in handles]
    got = step.join()
    torch.cuda.synchronize()
    assert rz._capacity[key] > 500
    for v, o in enumerate(outs):
        for a, b, nm in zip(o[:4], want_img[v], ("color", "depth", "alpha", "radii")):
            assert torch.equal(a, b), f"deferred redo, view {v}: {nm} differs from the operator path"
    for k, w in want.items():
        scale = float(w.abs().max())
        assert float((got[k] - w.reshape(got[k].shape)).abs().max()) <= 2e-5 * scale + 1e-9, (k, "deferred redo")


def test_optimistic_forward_is_bit_identical_and_survives_overflow():
    """The autograd node enqueues scatter / sort / blend before the host has read the instance count, against a binning buffer
    sized from earlier frames (ag_raster_forward_optimistic).  Same images and gradients, bit for bit in the forward, as the
    two-stage path; a frame that outgrows the planned capacity is redone through the two-stage path and raises the capacity."""
    import torch
    from animatablegaussians_amd import rasterizer as rz
    from animatablegaussians_amd.rasterizer import GaussianRasterizer
    sc = synth.random_gaussians(4000, seed=5, img=256, focal=275.0)
    sc.update(synth.upstream_grads(256, 256, 3))
    cam = h.cam_of(sc)
    rs = h.gpu_settings(sc, cam)
    key = (4000, 256, 256, rs.bg.device.index)

    def run():
        inp = h.gpu_inputs(sc, requires_grad=True)
        m2d = torch.zeros_like(inp["means3D"], requires_grad=True)
        color, radii, depth, alpha = GaussianRasterizer(rs)(means3D=inp["means3D"], means2D=m2d, opacities=inp["opacities"],
                                                            colors_precomp=inp["colors"], scales=inp["scales"], rotations=inp["rotations"])
        g = {k: torch.from_numpy(sc[k]).cuda() for k in ("dL_dcolor", "dL_ddepth", "dL_dalpha")}
        torch.autograd.backward([color, depth, alpha], [g["dL_dcolor"], g["dL_ddepth"], g["dL_dalpha"]])
        grads = [inp[k].grad.clone() for k in ("means3D", "opacities", "colors", "scales", "rotations")] + [m2d.grad.clone()]
        return [color, depth, alpha, radii], grads

    rz._capacity.pop(key, None)
    out0, gr0 = run()                                   # first frame of the configuration: two-stage path, plans a capacity
    cap = rz._capacity[key]
    R = h.gpu_native_forward(sc, cam)["num_rendered"]
    assert cap >= R + R // 4
    out1, gr1 = run()                                   # optimistic, roomy buffer
    rz._capacity[key] = R                               # optimistic, exactly full
    out2, gr2 = run()
    rz._capacity[key] = max(R // 3, 1)                  # too small: device-side guard, redo, capacity raised again
    out3, gr3 = run()
    assert rz._capacity[key] >= R + R // 4
    for outs in (out1, out2, out3):
        for a, b in zip(outs, out0):
            assert torch.equal(a, b)
    for grs in (gr1, gr2, gr3):
        for a, b in zip(grs, gr0):                      # float atomics: order-dependent rounding only
            assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max()))
    # raw ABI: overflow is reported with the true count and leaves the binning buffer untouched
    from animatablegaussians_amd import _lib
    rz._capacity[key] = max(R // 3, 1)
    sentinel_cap = max(R // 3, 1)
    import ctypes
    L = _lib.lib()
    inp = h.gpu_inputs(sc)
    P = 4000
    geom = torch.empty((L.ag_raster_geom_bytes(P),), dtype=torch.uint8, device="cuda")
    img = torch.empty((L.ag_raster_image_bytes(256, 256),), dtype=torch.uint8, device="cuda")
    binning = torch.full((L.ag_raster_binning_bytes(sentinel_cap),), 0xAB, dtype=torch.uint8, device="cuda")
    oc, od, oa = (torch.empty((c, 256, 256), device="cuda") for c in (3, 1, 1))
    radii = torch.empty((P,), dtype=torch.int32, device="cuda")
    a = _lib.AgRasterForwardArgs()
    a.P, a.W, a.H, a.sh_degree, a.sh_coeffs, a.prefiltered = P, 256, 256, 0, 0, 0
    a.tan_fovx, a.tan_fovy, a.scale_modifier = rs.tanfovx, rs.tanfovy, 1.0
    for name, t in (("bg", rs.bg), ("means3D", inp["means3D"]), ("colors_precomp", inp["colors"]), ("opacities", inp["opacities"]),
                    ("scales", inp["scales"]), ("rotations", inp["rotations"]), ("viewmatrix", rs.viewmatrix),
                    ("projmatrix", rs.projmatrix), ("campos", rs.campos), ("out_color", oc), ("out_depth", od), ("out_alpha", oa),
                    ("radii", radii), ("geom_buffer", geom), ("image_buffer", img), ("binning_buffer", binning)):
        setattr(a, name, t.data_ptr())
    a.geom_bytes, a.image_bytes, a.binning_bytes = geom.numel(), img.numel(), binning.numel()
    Rh = ctypes.c_int32(0)
    rc = L.ag_raster_forward_optimistic(ctypes.byref(a), sentinel_cap, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.byref(Rh))
    torch.cuda.synchronize()
    assert rc == _lib.AG_ERR_SCRATCH_TOO_SMALL and Rh.value == R
    assert bool((binning == 0xAB).all())
    rz._capacity.pop(key, None)


def test_optimistic_path_edge_cases_through_autograd():
    """The autograd node's optimistic forward on the reference's edge cases: P == 0 (all-zero colour, no native call), a frame with
    every Gaussian behind the near plane after a populated frame of the same shape (num_rendered 0 against a planned capacity:
    colour == bg), and gradients that are exactly zero for culled Gaussians."""
    import torch
    from animatablegaussians_amd import rasterizer as rz
    from animatablegaussians_amd.rasterizer import GaussianRasterizer
    sc = synth.random_gaussians(300, seed=9, img=96, focal=100.0)
    cam = h.cam_of(sc)
    rs = h.gpu_settings(sc, cam)
    key = (300, 96, 96, rs.bg.device.index)
    rz._capacity.pop(key, None)

    def run(scene):
        inp = h.gpu_inputs(scene, requires_grad=True)
        m2d = torch.zeros_like(inp["means3D"], requires_grad=True)
        out = GaussianRasterizer(rs)(means3D=inp["means3D"], means2D=m2d, opacities=inp["opacities"], colors_precomp=inp["colors"],
                                     scales=inp["scales"], rotations=inp["rotations"])
        (out[0].sum() + out[2].sum() + out[3].sum()).backward()
        return out, inp

    run(sc)                                                     # plans a capacity for (300, 96, 96)
    assert rz._capacity[key] > 0
    behind = dict(sc, means3D=sc["means3D"] + np.array([0, 0, 10.0], np.float32))
    (color, radii, depth, alpha), inp = run(behind)             # optimistic path, zero instances
    assert not radii.any() and not alpha.any() and not depth.any()
    assert torch.equal(color, rs.bg[:, None, None].expand_as(color))
    for k in ("means3D", "opacities", "colors", "scales", "rotations"):
        assert not inp[k].grad.any(), k
    empty = {k: (v[:0] if isinstance(v, np.ndarray) and v.shape[:1] == (300,) else v) for k, v in sc.items()}
    (color, radii, depth, alpha), _ = run(empty)                # P == 0
    assert radii.numel() == 0 and not color.any() and not alpha.any()
    rz._capacity.pop(key, None)
    rz._capacity.pop((0, 96, 96, rs.bg.device.index), None)


def test_the_large_class_sort_is_enqueued_only_once_a_frame_needed_it():
    """Round 5 (include/ag_raster.h ag_raster_large_tile_sort): the enqueue-everything paths skip the tile sort's large-class launch until a frame
    of the process has had a tile of >= 2048 instances.  From a forgotten state: a scene with short lists leaves the flag down; the first
    OPTIMISTIC frame with such a tile is refused inside the library, redone by the caller's overflow path and comes out bit-identical to the
    host-synchronised path (which always launches the large class and is itself bit-exact against the oracle) -- through the operator
    (GaussianRasterizer) and through the library-owned step (enqueue + collect, gradients included); the flag is up afterwards."""
    import torch
    from animatablegaussians_amd import _lib, rasterizer as rz
    from animatablegaussians_amd.rasterizer import FusedRasterStep, GaussianRasterizer
    L = _lib.lib()
    small = synth.random_gaussians(4000, seed=3, img=256, focal=275.0)
    rs = np.random.RandomState(7)
    P = 9000
    big = synth.random_gaussians(P=P)
    big["means3D"] = (rs.normal(0, 0.004, (P, 3))).astype(np.float32)
    big["scales"] = np.full((P, 3), 0.002, np.float32)
    big["opacities"] = np.full((P, 1), 0.02, np.float32)

    def operator(scene, cam):
        inp = h.gpu_inputs(scene)
        with torch.no_grad():
            color, radii, depth, alpha = GaussianRasterizer(h.gpu_settings(scene, cam))(means3D=inp["means3D"], means2D=torch.zeros_like(inp["means3D"]), opacities=inp["opacities"],
                                                                                        colors_precomp=inp["colors"], scales=inp["scales"], rotations=inp["rotations"])
        torch.cuda.synchronize()
        return color.cpu().numpy(), alpha.cpu().numpy(), radii.cpu().numpy()

    def same(got, ref):
        assert np.array_equal(got[0], ref["color"]) and np.array_equal(got[1], ref["alpha"]) and np.array_equal(got[2], ref["radii"])
    try:
        assert L.ag_raster_large_tile_sort(0) == 0
        cam_s, cam_b = h.cam_of(small), h.cam_of(big)
        ref_s, ref_b = h.gpu_native_forward(small, cam_s), h.gpu_native_forward(big, cam_b)          # plan + render: always the full sort
        _bitexact(ref_b, h.oracle_forward(big, cam_b))
        assert (h.oracle_forward(big, cam_b)["ranges"][:, 1] - h.oracle_forward(big, cam_b)["ranges"][:, 0]).max() > 4096
        for _ in range(3):
            same(operator(small, cam_s), ref_s)
        assert L.ag_raster_large_tile_sort(-1) == 0                       # nothing long: the launch is still skipped
        same(operator(big, cam_b), ref_b)                                 # first frame of this configuration: plan + render
        same(operator(big, cam_b), ref_b)                                 # optimistic, launch skipped -> refused -> redone
        assert L.ag_raster_large_tile_sort(-1) == 1
        same(operator(big, cam_b), ref_b)                                 # optimistic with the launch
        # the library-owned step: forward + backward of the view against the operator path, from a forgotten state
        inp = h.gpu_inputs(big, requires_grad=True)
        rset = h.gpu_settings(big, cam_b)
        up = synth.upstream_grads(rset.image_width, rset.image_height, 5)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        m2d = torch.zeros_like(inp["means3D"], requires_grad=True)
        color, radii, depth, alpha = GaussianRasterizer(rset)(means3D=inp["means3D"], means2D=m2d, opacities=inp["opacities"], colors_precomp=inp["colors"],
                                                              scales=inp["scales"], rotations=inp["rotations"])
        torch.autograd.backward([color, depth, alpha], [t(up["dL_dcolor"]), t(up["dL_ddepth"]), t(up["dL_dalpha"])])
        det = {k: v.detach() for k, v in inp.items() if v is not None}
        assert L.ag_raster_large_tile_sort(0) == 0
        step = FusedRasterStep(P, rset.image_height, rset.image_width, "cuda", n_streams=1)
        c2, d2, a2, r2, _ = step.view(rset, det["means3D"], det["colors"], det["opacities"], det["scales"], det["rotations"],
                                      t(up["dL_dcolor"]), t(up["dL_ddepth"]), t(up["dL_dalpha"]), accumulate=True)
        got = step.join()
        torch.cuda.synchronize()
        assert L.ag_raster_large_tile_sort(-1) == 1
        assert torch.equal(c2, color.detach()) and torch.equal(a2, alpha.detach()) and torch.equal(r2, radii)
        for k, w in (("dL_dmeans3D", inp["means3D"].grad), ("dL_dcolors", inp["colors"].grad), ("dL_dopacity", inp["opacities"].grad)):
            scale = float(w.abs().max())
            assert float((got[k] - w.reshape(got[k].shape)).abs().max()) <= 5e-5 * scale + 1e-9, k
    finally:
        L.ag_raster_large_tile_sort(0)
