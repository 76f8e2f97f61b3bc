"""SMPL-X forward -- SURVEY.md §8(f)-2 -- against the reference's own `smplx.SMPLX` class.

Fixture tests/golden/smplx_body.npz: the reference class (imported from /root/reference in the build container) evaluated on
the synthetic model file of `synth.smplx_model_arrays` for the three calls of a data item (dataset_mv_rgb.py:118-143) and the
`cano2live` products (:170-171), two frames, in float32 and float64.  CPU: the torch oracle restatement reproduces it.
GPU: `animatablegaussians_amd.smplx.SMPLX` (ag_smplx_forward, one batch of three) reproduces it to 1e-4 of the value scale
(measured ~3e-7: the reference's own float32 sits 3e-7 from its float64)."""
import math
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "smplx_body.npz")
ITEM_KEYS = ("live_smpl_v", "cano_smpl_v", "live_smpl_v_woRoot", "joints", "cano_jnts", "cano2live_jnt_mats", "cano2live_jnt_mats_woRoot")


def _cano():
    import torch
    cp = np.zeros(75, np.float32)                       # config.py:9-15
    cp[3 + 3 * 1 + 2] = math.radians(25)
    cp[3 + 3 * 2 + 2] = math.radians(-25)
    cp = torch.from_numpy(cp)
    return cp[3:6], cp[:3], cp[6:69]                    # global_orient, transl, body_pose


def test_oracle_reproduces_the_reference_class():
    import torch
    from animatablegaussians_amd import synth
    from oracle import smplx_oracle as so
    gold = np.load(GOLD)
    arrays, params = synth.smplx_model_arrays(), synth.smplx_pose_params(n=2)
    go, tr, bp = _cano()
    for dt, tag, tol in ((torch.float32, "f32", 2e-6), (torch.float64, "f64", 2e-7)):
        m = so.model_tensors(arrays, dt)
        for i in range(2):
            item = so.data_item(m, params, i, go, tr, bp)
            for k in ITEM_KEYS:
                ref, got = gold[f"{tag}_{i}_{k}"], item[k].numpy()
                if ref.shape[0] != got.shape[0]:
                    got = got[::16]                     # float64 truth is stored for every 16th vertex
                assert float(np.abs(got - ref).max()) <= tol, (tag, i, k)


def test_synthetic_model_file_has_the_layout_the_reference_reads():
    from animatablegaussians_amd import synth
    a = synth.smplx_model_arrays(V=10475, shape_dims=20, n_faces=64)
    assert a['v_template'].shape == (10475, 3) and a['posedirs'].shape == (10475, 3, 486) and a['J_regressor'].shape == (55, 10475)
    assert a['weights'].shape == (10475, 55) and np.allclose(a['weights'].sum(1), 1) and np.allclose(a['J_regressor'].sum(1), 1)
    par = a['kintree_table'][0].astype(np.int64)
    assert par[0] == 2 ** 32 - 1 and all(par[j] < j for j in range(1, 55))


def test_constructor_refuses_what_is_not_built():
    from animatablegaussians_amd.smplx import SMPLX
    with pytest.raises(NotImplementedError):
        SMPLX({}, use_pca=True)
    with pytest.raises(FileNotFoundError):
        SMPLX("/nonexistent/dir/SMPLX_NEUTRAL.npz", use_pca=False)


@pytest.fixture(scope="module")
def gpu_model():
    import torch
    from animatablegaussians_amd import synth
    from animatablegaussians_amd.smplx import SMPLX
    return SMPLX(synth.smplx_model_arrays(), gender='neutral', use_pca=False, num_pca_comps=45, flat_hand_mean=True, batch_size=1,
                 device=torch.device("cuda", 0))


@pytest.mark.gpu
def test_data_item_matches_the_reference_class(gpu_model):
    import torch
    from animatablegaussians_amd import synth
    gold = np.load(GOLD)
    params = synth.smplx_pose_params(n=2)
    go, tr, bp = _cano()
    for i in range(2):
        item = gpu_model.data_item(params, i, go, tr, bp)
        for k in ITEM_KEYS:
            ref = gold[f"f32_{i}_{k}"]
            got = item[k].cpu().numpy()
            if k == "joints":
                ref = ref[:22]                          # dataset_mv_rgb.py:155
            scale = max(1.0, float(np.abs(ref).max()))
            assert float(np.abs(got - ref).max()) <= 1e-4 * scale, (i, k, float(np.abs(got - ref).max()))
            assert float(np.abs(got - ref).max()) <= 5e-6 * scale, (i, k, float(np.abs(got - ref).max()))
        assert item['kin_parent'].tolist() == list(synth.SMPLX_PARENTS[:22])
        lv = torch.from_numpy(gold[f"f32_{i}_live_smpl_v"])
        np.testing.assert_allclose(item['live_bounds'].cpu().numpy(), torch.stack([lv.min(0)[0] - 0.15, lv.max(0)[0] + 0.15]).numpy(), atol=1e-5)


@pytest.mark.gpu
def test_single_call_surface_joints_and_batching(gpu_model):
    """forward() with the reference's argument names: 127 joints (55 + 21 vertex picks + 51 landmarks), A, defaults for
    omitted arguments, v_shaped; a batch of 5 (two skinning launches: 4 + 1) equals five single calls bit for bit."""
    import torch
    from animatablegaussians_amd import synth
    from oracle import smplx_oracle as so
    gold = np.load(GOLD)
    p = {k: torch.from_numpy(v) for k, v in synth.smplx_pose_params(n=2).items()}
    out = gpu_model.forward(betas=p['betas'], global_orient=p['global_orient'][:1], transl=p['transl'][:1], body_pose=p['body_pose'][:1],
                            jaw_pose=p['jaw_pose'][:1], expression=p['expression'][:1], left_hand_pose=p['left_hand_pose'][:1],
                            right_hand_pose=p['right_hand_pose'][:1], return_full_pose=True, return_shaped=True)
    assert out.joints.shape == (1, 127, 3) and out.A.shape == (1, 55, 4, 4) and out.full_pose.shape == (1, 165)
    np.testing.assert_allclose(out.joints[0].cpu().numpy(), gold["f32_0_joints"], atol=5e-6)
    np.testing.assert_allclose(out.A[0].cpu().numpy(), gold["f32_0_live_A"], atol=5e-6)
    m = so.model_tensors(synth.smplx_model_arrays(), torch.float32)
    vs = m['v_template'] + torch.einsum('bl,mkl->bmk', p['betas'], m['shapedirs'])[0]
    np.testing.assert_allclose(out.v_shaped[0].cpu().numpy(), vs.numpy(), atol=2e-6)
    # defaults (body_models.py:1185-1199): everything omitted = zero pose, zero shape
    rest = gpu_model.forward()
    ref = so.forward(m, torch.zeros(1, 10))
    np.testing.assert_allclose(rest.vertices[0].cpu().numpy(), ref['vertices'][0].numpy(), atol=5e-6)
    assert float((rest.A[0, :, :3, :3].cpu() - torch.eye(3)).abs().max()) < 1e-6
    # batching
    q = {k: torch.from_numpy(v) for k, v in synth.smplx_pose_params(seed=5, n=5).items()}
    kw = lambda s: dict(betas=q['betas'], global_orient=q['global_orient'][s], transl=q['transl'][s], body_pose=q['body_pose'][s],
                        jaw_pose=q['jaw_pose'][s], expression=q['expression'][s], left_hand_pose=q['left_hand_pose'][s],
                        right_hand_pose=q['right_hand_pose'][s])
    allb = gpu_model.forward(**kw(slice(0, 5)))
    for i in range(5):
        one = gpu_model.forward(**kw(slice(i, i + 1)))
        assert torch.equal(one.vertices[0], allb.vertices[i]) and torch.equal(one.A[0], allb.A[i]) and torch.equal(one.joints[0], allb.joints[i])


@pytest.mark.gpu
def test_mat4_mul_inverse_general_matrices():
    import torch
    from animatablegaussians_amd.smplx import mat4_mul_inverse
    g = torch.Generator().manual_seed(3)
    a = torch.randn(130, 4, 4, generator=g)
    b = torch.randn(65, 4, 4, generator=g) + 2 * torch.eye(4)
    ref = (a.double().reshape(2, 65, 4, 4) @ torch.linalg.inv(b.double())).reshape(130, 4, 4)
    got = mat4_mul_inverse(a.cuda(), b.cuda()).cpu().double()
    cond = torch.linalg.cond(b.double()).repeat(2)
    assert float(((got - ref).abs().amax((1, 2)) / (ref.abs().amax((1, 2)) * cond)).max()) < 1e-6
    with pytest.raises(RuntimeError):
        mat4_mul_inverse(a.cuda(), torch.randn(7, 4, 4).cuda())
