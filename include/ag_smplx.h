/*
 * ag_smplx.h — C ABI of the SMPL-X body-model forward (libag_hip.so), fp32, B poses of one subject per call.
 *
 * the producer of `cano2live_jnt_mats` (what AvatarNet.transform_cano2live skins the Gaussians with)
 * is the dataset-side SMPL-X forward -- three `smpl_model.forward` calls per item (live, canonical, live without root:
 * dataset/dataset_mv_rgb.py:118-143) followed by `live.A @ inv(cano.A)` (:170-171).  The reference runs it on the CPU in
 * the data loader (smplx/body_models.py:1114-1290 -> smplx/lbs.py:152-246); here the B poses go through three launches
 * that read the 61-MB pose-corrective basis ONCE for all of them.
 *
 * Replaces, stage by stage (smplx/lbs.py):
 *   :208  v_shaped = v_template + blend_shapes(betas ++ expression, shapedirs ++ expr_dirs)      (ag_smplx_forward, kernel 1)
 *   :212  J = vertices2joints(J_regressor, v_shaped)           (kernel 2, through the per-model fold of ag_smplx_prepare)
 *   :218  rot_mats = batch_rodrigues(pose)  (:299-330, incl. its `+ 1e-8` inside the norm)        (kernel 2)
 *   :221  pose_feature = (rot_mats[1:] - I).view(-1)                                              (kernel 2)
 *   :235  J_transformed, A = batch_rigid_transform(rot_mats, J, parents)  (:347-405)              (kernel 2)
 *   :223  pose_offsets = pose_feature @ posedirs;  :233 v_posed = pose_offsets + v_shaped         (kernel 3)
 *   :239-248  T = W @ A;  verts = (T @ [v_posed, 1])[:3]                                          (kernel 3)
 *   body_models.py:1272-1275  `+ transl` on vertices, joints and A[:, :3, 3] AFTER the skinning   (kernels 2, 3)
 * Device pointers, contiguous row-major; 0 on success (codes in ag_raster.h).
 */
#ifndef AG_SMPLX_H
#define AG_SMPLX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The constant arrays of one body model, as smplx/body_models.py:237-260,1049-1073 registers them. */
typedef struct AgSmplxModel {
    int32_t V;                 /* vertices (10475 for SMPL-X) */
    int32_t J;                 /* joints of the kinematic tree (55), J <= 64 */
    int32_t NB;                /* shape + expression coefficients in use (10 + 10) */
    int32_t reserved;
    const float* v_template;   /* [V][3] */
    const float* shapedirs;    /* [V][3][NB]  = cat(shapedirs[..., :num_betas], expr_dirs) (body_models.py:1233) */
    const float* posedirs;     /* [9 (J-1)][3 V]  (body_models.py:248-252: reshape(-1, P).T) */
    const float* J_regressor;  /* [J][V] dense */
    const int32_t* parents;    /* [J], parents[0] = -1, parents[j] < j */
    const float* lbs_weights;  /* [V][J] */
    const float* joint_template; /* [J][3]      = J_regressor . v_template   } written once per model by ag_smplx_prepare */
    const float* joint_dirs;     /* [J][3][NB]  = J_regressor . shapedirs    } (device memory owned by the caller)        */
} AgSmplxModel;

/* Folds the joint regressor through the (linear) shape model: lbs.py:208-212 computes J_regressor . (v_template + shapedirs . c)
 * per call, a 10475-long reduction per joint; J_regressor . v_template + (J_regressor . shapedirs) . c is the same sum
 * re-associated (difference ~1e-7 of the joint positions).  joint_template [J][3], joint_dirs [J][3][NB]: device outputs;
 * the model's own joint_template / joint_dirs fields are not read by this call. */
int ag_smplx_prepare(const AgSmplxModel* m, float* joint_template, float* joint_dirs, void* stream);

/* Floats of workspace ag_smplx_forward needs for B poses (v_shaped, un-translated joint matrices, pose features). */
size_t ag_smplx_workspace_floats(const AgSmplxModel* m, int32_t B);

/*
 * B poses of one subject.  shape_components [B][NB]; full_pose [B][J][3] axis-angle (pose mean already added,
 * body_models.py:1203-1213); transl [B][3] or NULL.
 * Outputs: vertices [B][V][3]; joints [B][J][3] (posed joint locations, the first J rows of the reference's `joints`);
 * A [B][J][4][4] (the reference's `A`, relative to the rest pose, translation included).  workspace: device floats.
 */
int ag_smplx_forward(const AgSmplxModel* m, int32_t B, const float* shape_components, const float* full_pose, const float* transl,
                     float* vertices, float* joints, float* A, float* workspace, size_t workspace_floats, void* stream);

/* v_shaped [B][V][3] = v_template + shapedirs . shape_components alone (body_models.py:1277-1279 `return_shaped`, where the
 * reference passes the betas without the expression: the caller zeroes those components). */
int ag_smplx_shape(const AgSmplxModel* m, int32_t B, const float* shape_components, float* v_shaped, void* stream);

/* out[i] = a[i] @ inverse(b[i % b_batch]) for n row-major 4x4 matrices (dataset_mv_rgb.py:170-171: cano2live_jnt_mats =
 * live.A @ inv(cano.A), and the same canonical matrices again for the pose without root: n = 2 J, b_batch = J). */
int ag_mat4_mul_inverse(float* out, const float* a, const float* b, int32_t n, int32_t b_batch, void* stream);

/* Barycentric key points (vertex picks and face landmarks: vertex_joint_selector.py:72-76, lbs.py:108-149):
 * out[b][k] = sum_t w[k][t] * vertices[b][idx[k][t]], t < 3.  idx [K][3] int32, w [K][3]. */
int ag_smplx_keypoints(float* out, const float* vertices, const int32_t* idx, const float* w, int32_t B, int32_t V, int32_t K, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AG_SMPLX_H */
