/*
 * ag_linear.h -- the style path of the StyleUNets: EqualLinear layers on one-row inputs (round 5).
 *
 * Reference: network/styleunet/dual_styleunet.py:131-165 (EqualLinear: F.linear(x, weight * scale, bias * lr_mul), scale = lr_mul / sqrt(in);
 * with an activation: fused_leaky_relu(F.linear(x, weight * scale), bias * lr_mul)), :13-18 (PixelNorm), :594-610 (the mapping network
 * PixelNorm + n_mlp x EqualLinear(lr_mul 0.01, fused leaky ReLU)), :225-300 (ModulatedConv2d.modulation = EqualLinear(style_dim, in_channel,
 * bias_init 1) of every StyledConv / ToRGB: 36 per network).
 *
 * These are matrix-VECTOR products (the batch is one style row): a training step of the three networks ran ~130 torch launches for them
 * (scale the weight, scale the bias, GEMV, the same backwards, two concatenations of 12.6 MB of modulation weights per decoder branch:
 * profiles/r05b_glue_v1.txt, ~0.8 ms of a 33-ms step for a few MFLOP).  Here a CALL evaluates up to AG_LINEAR_MAX_JOBS layers ("jobs") that
 * read one-row inputs -- the 15 modulation layers of a decoder branch's shared stages on the same latent; the mapping-network layer of the
 * three networks, each on its own latent -- as one launch forward and two backward, straight from the parameter tensors (no stacking) and into
 * the per-parameter gradient tensors.
 *
 *   forward   y[b][o] = act( alpha_j * sum_c xn[b][c] * W_j[o][c] + bias_j[o] * bias_mul_j )        one wave per output row
 *             xn = x, or PixelNorm(x) = x * rsqrt(mean_c x^2 + 1e-8) when normalize_input
 *             act = identity, or the reference's fused leaky ReLU: leaky_relu(., 0.2) * sqrt(2)
 *   backward  g'  = g_y * (act: y > 0 ? sqrt 2 : 0.2 sqrt 2)            (the reference's fused_bias_act backward selects on the OUTPUT's sign)
 *             g_W_j[o][c] = alpha_j sum_b g'[b][o] xn[b][c];   g_bias_j[o] = bias_mul_j sum_b g'[b][o]
 *             g_x[b][c]   = sum over the jobs that share the input of alpha_j sum_o g'[b][o] W_j[o][c]: 16-row partial sums per wave, added in a
 *                           fixed order by a second launch (deterministic, no atomics)
 * All tensors dense fp32; in_features a multiple of 4.  Errors: AG_OK or AG_ERR_* (include/ag_raster.h), message in ag_last_error().
 */
#ifndef AG_LINEAR_H
#define AG_LINEAR_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AG_LINEAR_MAX_JOBS 32

typedef struct AgEqualLinearArgs {
    int32_t n_jobs;                               /* 1 .. AG_LINEAR_MAX_JOBS */
    int32_t B;                                    /* rows of every input / output (the style batch; 1 in this product), <= 8 */
    int32_t in_features;                          /* columns of every input and weight, a multiple of 4 */
    int32_t act;                                  /* 0: none; 1: fused leaky ReLU (slope 0.2, gain sqrt 2) */
    int32_t normalize_input;                      /* 1: PixelNorm the input rows first (forward and g_weight; g_x must then be NULL) */
    int32_t reserved;
    const float* x[AG_LINEAR_MAX_JOBS];           /* [B, in]; jobs that read the same input repeat the pointer and are consecutive */
    const float* weight[AG_LINEAR_MAX_JOBS];      /* [out_j, in] */
    const float* bias[AG_LINEAR_MAX_JOBS];        /* [out_j] or NULL */
    int32_t out_features[AG_LINEAR_MAX_JOBS];
    float alpha[AG_LINEAR_MAX_JOBS];              /* EqualLinear.scale */
    float bias_mul[AG_LINEAR_MAX_JOBS];           /* EqualLinear.lr_mul */
    float* y[AG_LINEAR_MAX_JOBS];                 /* [B, out_j] per job (forward: written; backward: read when act) */
    /* backward only */
    const float* g_y[AG_LINEAR_MAX_JOBS];         /* [B, out_j] per job: the jobs' outputs are separate autograd tensors, so their gradients arrive separately */
    float* g_x[AG_LINEAR_MAX_JOBS];               /* [B, in] or NULL; the jobs of one input pass the same pointer: it receives their sum */
    float* g_weight[AG_LINEAR_MAX_JOBS];          /* [out_j, in] or NULL: every element written */
    float* g_bias[AG_LINEAR_MAX_JOBS];            /* [out_j] or NULL */
    float* scratch;                               /* backward with any g_x: ag_equal_linear_scratch_floats() floats */
} AgEqualLinearArgs;

size_t ag_equal_linear_args_bytes(void);
size_t ag_equal_linear_scratch_floats(const AgEqualLinearArgs* a);
int ag_equal_linear_forward(const AgEqualLinearArgs* a, void* stream);
int ag_equal_linear_backward(const AgEqualLinearArgs* a, void* stream);

/*
 * Bilinear resize of [N, H, W] planes to [N, OH, OW], torch.nn.functional.interpolate(mode="bilinear", align_corners=False) semantics
 * (source coordinate (o + 0.5) * in / out - 0.5 clamped at 0, the upper neighbour clamped at in - 1): the view-direction feature of the colour
 * network is resized to the decoder stage's resolution before it is added (dual_styleunet.py:881-883).  torch's kernels take 125 us forward and
 * 88 us backward for the 128 x 128 -> 256 x 256 resize of 128 planes (33 MB written); these are streaming passes (one thread per output
 * element; the backward GATHERS: one thread per input element sums the output gradients that read it, in a fixed order -- no atomics).
 */
int ag_bilinear_resize_forward(float* out, const float* in, int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW, void* stream);
int ag_bilinear_resize_backward(float* g_in, const float* g_out, int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW, void* stream);

/*
 * Input of a view-dependent decoder stage for the stacked members (dual_styleunet.py:881-883 `out = out + F.interpolate(view_feature, size,
 * mode="bilinear")`, for M members that continue rows of the shared state):
 *     x[m] = out[src[m]] (+ resize(vf[m - r0]) for r0 <= m < r1)          x [M, C, H, W], out [*, C, H, W], vf [r1 - r0, C, vh, vw] or NULL
 * one pass: a row selection, an addition and V bilinear resizes (33 MB each at 128 planes 256^2) were three passes and a concatenation before.
 * `src`: M host integers, read at call time.  vh == H and vw == W: plain addition.
 */
int ag_select_add_rows(float* x, const float* out, const int32_t* src, int32_t M, int32_t C, int32_t H, int32_t W, const float* vf, int32_t r0, int32_t r1,
                       int32_t vh, int32_t vw, void* stream);

/*
 * out[p] = sum of the `len` floats of plane p, for `planes` contiguous planes: the bias gradient of a ToRGB head (dual_styleunet.py:607-633: the
 * sum of the output gradient over batch and pixels; torch's reduction ran at ~1 TB/s on these [G, 12 | 32, 512, 512] tensors, 0.2 ms per step).
 * Deterministic: every plane is cut into the same slices on every run, slice sums are added in slice order by a second launch.
 * scratch: ag_plane_sums_scratch_floats(planes, len) floats.
 */
size_t ag_plane_sums_scratch_floats(int32_t planes, int64_t len);
int ag_plane_sums(float* out, const float* in, int32_t planes, int64_t len, float* scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AG_LINEAR_H */
