/*
 * ag_avatar.h — C ABI of the per-Gaussian assembly + linear-blend-skinning kernels (libag_hip.so).
 *
 * These replace the chains of torch ops the reference runs between its StyleUNets and its rasterizer:
 *   AvatarNet.get_positions / get_others / get_colors   network/avatar.py:93-124
 *     (split front/back -> cat along W -> permute -> boolean-mask gather -> 0.05*d + xyz, sigmoid, exp, F.normalize;
 *      activations from gaussians/gaussian_model.py:53-61)
 *   AvatarNet.transform_cano2live                        network/avatar.py:84-91
 *     (einsum('nj,jxy->nxy') blend of the 55 joint matrices, R p + t,
 *      pytorch3d.transforms.quaternion_to_matrix / matrix_to_quaternion)
 * and their autograd backward.  Same conventions as ag_raster.h: device pointers, fp32, contiguous, 0 on success.
 */
#ifndef AG_AVATAR_H
#define AG_AVATAR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Gather + activations.  The three network outputs are NCHW [1, 2C, S, S] with the front map in channels [0, C) and
 * the back map in [C, 2C) (C = 3, 8, 3).  `pix` lists, in ascending order, the set pixels of the reference's
 * [S, 2S] front|back canvas mask as v * 2S + u — exactly the order `canvas[mask]` enumerates them.
 */
typedef struct AgGatherArgs {
    int32_t N;                  /* number of Gaussians = number of set mask pixels */
    int32_t S;                  /* side of one map (1024) */
    const int32_t* pix;         /* [N] */
    const float* position_map;  /* [1, 6, S, S] */
    const float* other_map;     /* [1, 16, S, S]: per side opacity(1), scale(3), rotation(4) */
    const float* color_map;     /* [1, 6, S, S] */
    const float* xyz;           /* [N,3] canonical positions     (cano_gaussian_model.get_xyz) */
    const float* opacity_raw;   /* [N,1]                          (get_opacity_raw) */
    const float* scaling_raw;   /* [N,3]                          (get_scaling_raw) */
    const float* rotation_raw;  /* [N,4]                          (get_rotation_raw) */
    /* forward outputs / backward inputs (upstream gradients when used by the backward) */
    float* positions;           /* [N,3] = 0.05 * d + xyz */
    float* opacity;             /* [N,1] = sigmoid(o + raw) */
    float* scales;              /* [N,3] = exp(s + raw) */
    float* rotations;           /* [N,4] = normalize(r + raw), eps 1e-12 */
    float* colors;              /* [N,3] */
} AgGatherArgs;

/*
 * The three parts can be requested separately, the way the reference trainer's pre-training pass calls them
 * (main_avatar.py:126-160 -> AvatarNet.get_positions / get_others / get_colors one at a time): a part is produced when its
 * output pointer(s) are set -- `positions` | `opacity`+`scales`+`rotations` (all three or none) | `colors` -- and only that
 * part's inputs are read.  A null MAP pointer of a requested part stands for an all-zero network output, which yields the
 * canonical model's own getters (GaussianModel.get_xyz / get_opacity / get_scaling / get_rotation,
 * gaussians/gaussian_model.py:115-147).
 */
int ag_gather_activate_forward(const AgGatherArgs* args, void* stream);

/*
 * Backward: `grads` carries dL/d{positions, opacity, scales, rotations, colors} in the output slots of AgGatherArgs
 * (the map / raw pointers are the forward inputs).  Writes the full gradient maps (zero outside the mask), i.e. what
 * autograd produces for the reference's index/permute/cat/split chain.  A null gradient-map pointer skips that part.
 */
int ag_gather_activate_backward(const AgGatherArgs* fwd_inputs_and_grads, float* dL_dposition_map /*[1,6,S,S]*/,
                                float* dL_dother_map /*[1,16,S,S]*/, float* dL_dcolor_map /*[1,6,S,S]*/, void* stream);

/* Linear-blend skinning of positions and rotations (quaternions as (r, i, j, k)). */
typedef struct AgLbsArgs {
    int32_t N;
    int32_t J;                  /* joints (55 for SMPL-X); 1 <= J <= 256 */
    const float* lbs;           /* [N, J] blend weights */
    const float* jnt_mats;      /* [J, 4, 4] cano2live joint matrices, row-major */
    const float* positions;     /* [N,3] */
    const float* rotations;     /* [N,4] */
    float* out_positions;       /* [N,3]  (backward: dL/d out_positions, input) */
    float* out_rotations;       /* [N,4]  (backward: dL/d out_rotations, input) */
    /* Sparse form of `lbs` (K > 0 selects it; `lbs` may then be NULL).  Blend-weight rows interpolated from the SMPL-X skinning weights
     * have at most ~12 non-zeros out of 55 (gen_data/gen_pos_maps.py:24-39,132): K (joint, weight) pairs per Gaussian, ascending
     * joints, zero-weight padded, stored [K][N] -- 5 K bytes per Gaussian instead of 4 J.  Skipping exact zeros leaves every partial
     * sum of the blend unchanged, so results equal the dense path's. */
    const uint8_t* sp_idx;      /* [K, N] joint indices */
    const float* sp_w;          /* [K, N] weights */
    int32_t K;                  /* 0: dense */
    int32_t reserved;
} AgLbsArgs;

int ag_lbs_forward(const AgLbsArgs* args, void* stream);
int ag_lbs_backward(const AgLbsArgs* fwd_inputs_and_grads, float* dL_dpositions /*[N,3]*/, float* dL_drotations /*[N,4]*/,
                    void* stream);

/*
 * Eval-time hand fusion (network/avatar.py:183-200): inside the bounding boxes of the canonical MANO hands the
 * predicted Gaussians are cross-faded into those of one fixed "mean hands" frame (AvatarNet.generate_mean_hands, :52-77).
 * Per Gaussian, with n_l / n_r = first coordinate of utils/geo_util.py:104-114 normalize_vert_bbox(hand verts, attris = xyz,
 * per_axis = True), i.e. 2 (x - centre_x) / extent_x of the left / right hand box:
 *     wl = sigmoid( 2.5 (n_l + 2)),  wr = sigmoid(-2.5 (n_r - 2)),  both 0 where xyz.y < centre_y;
 *     s = max(wl + wr, 1);  w = (wl + wr) / s;   attr = w * hand_attr + (1 - w) * attr      (positions, opacity, scales, rotations)
 * In place on the four attribute arrays.  left_box / right_box: device pointers to {min_x, max_x} of the hand vertices.
 */
typedef struct AgHandFuseArgs {
    int32_t N;
    int32_t reserved;
    const float* xyz;             /* [N,3] canonical positions (self.init_points) */
    const float* left_box;        /* [2]: min and max x of items['left_cano_mano_v'] */
    const float* right_box;       /* [2]: ... of items['right_cano_mano_v'] */
    const float* centre;          /* [3]: items['cano_smpl_center'] (y is read) */
    const float* hand_positions;  /* [N,3] */
    const float* hand_opacity;    /* [N,1] */
    const float* hand_scales;     /* [N,3] */
    const float* hand_rotations;  /* [N,4] */
    float* positions;             /* [N,3] in/out */
    float* opacity;               /* [N,1] in/out */
    float* scales;                /* [N,3] in/out */
    float* rotations;             /* [N,4] in/out */
} AgHandFuseArgs;

int ag_hand_fuse(const AgHandFuseArgs* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AG_AVATAR_H */
